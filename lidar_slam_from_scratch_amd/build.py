"""In-tree build of libicp_mi355x.so (HIP kernels + C ABI) for gfx950."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libicp_mi355x.so")
_SOURCES = ["capi.hip", "sort.hip", "kernels.h", "device_math.h", "nn_mfma.h", "icp_small.h", "knn_lists.h", "nn_bounded.h", "nn_culled.h", "voxel.h", "scan_context.h", "occupancy.h", "Makefile"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    # a -D variant left in place by a sweep script that did not get to restore the product build
    stamp = os.path.join(CSRC, ".build_flags")
    if os.path.exists(stamp) and open(stamp).read().strip():
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in _SOURCES]
    deps.append(os.path.join(_HERE, "..", "include", "icp_mi355x.h"))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile with hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if force or _stale():
        cmd = ["make", "-C", CSRC, "OUT=" + LIB_PATH, "EXTRA="]
        stamp = os.path.join(CSRC, ".build_flags")
        if force or (os.path.exists(stamp) and open(stamp).read().strip()):
            cmd.insert(1, "-B")
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or out.returncode != 0:
            print(out.stdout)
        if out.returncode != 0:
            raise RuntimeError("hipcc build of libicp_mi355x.so failed")
    return LIB_PATH
