"""Same-box A/B of two builds of the library on what the per-iteration step costs (round 4: the step by a whole wave,
device_math.h ldlt6_solve_wave): the iteration PERIOD at ~8k -> 8k points (small-cloud kernel; two device-resident calls
of 20 and 60 forced iterations, (t60 - t20) / 40), the C3 call (100k -> 100k, 30 iterations, default engine, median of 9)
and a 60-frame file -> pose stream.  Every leg is a child process, legs alternate; the results' bits are printed so that
the two builds can be compared.  usage: python scripts/ab_step.py <other_lib.so> [reps]   ("product" is the in-tree one)"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "scripts"))
import numpy as np, torch
from lidar_slam_from_scratch_amd import capi, synth, odometry
if os.environ.get("ICPMI_AB_LIB"):
    capi._LIB = capi.load_library(os.environ["ICPMI_AB_LIB"])
out = {}
ctx = capi.Context(device=0)
rng = np.random.default_rng(7)
A = synth.lidar_frame(3, voxel=0.4, **synth.DRIVE_200)
B = synth.lidar_frame(4, voxel=0.4, **synth.DRIVE_200)
def sub(P, keep):
    return np.ascontiguousarray(P[np.sort(rng.choice(P.shape[0], min(keep, P.shape[0]), replace=False))])
a, b = sub(A, 8000), sub(B, 8000)
def timed(src, tgt, iters, reps, tol=0.0):
    d_a = torch.as_tensor(tgt, device="cuda"); d_b = torch.as_tensor(src, device="cuda"); torch.cuda.synchronize()
    cfg = capi.Context.make_config(max_iterations=iters, tolerance=tol, min_error=0.0)
    ts, h = [], None
    for _ in range(reps):
        t0 = time.perf_counter(); r, h = ctx.align_device(d_b.data_ptr(), src.shape[0], d_a.data_ptr(), tgt.shape[0], cfg)
        ts.append(time.perf_counter() - t0)
    return ts, h
timed(b, a, 20, 2)
t20 = min(timed(b, a, 20, 12)[0]); ts60, h60 = timed(b, a, 60, 12); t60 = min(ts60)
out["period_8k_us"] = round(1e6 * (t60 - t20) / 40, 3)
out["hist60_tail"] = float.hex(float(h60[-1]))
src, tgt, _ = synth.c3_uniform()
timed(src, tgt, 30, 2)
ts, h = timed(src, tgt, 30, 9)
out["c3_call_ms"] = round(1e3 * float(np.median(ts)), 4)
out["c3_it_per_s"] = round(30 / float(np.median(ts)), 1)
out["c3_hist_tail"] = float.hex(float(h[-1]))
drive = os.environ["ICPMI_AB_DRIVE"]
paths = [p for _, p in capi.discover_frames(drive)]
odometry.run_odometry_stream(paths[:4], ctx)
best = None
for _ in range(3):
    t0 = time.perf_counter(); tr = odometry.run_odometry_stream(paths, ctx); w = time.perf_counter() - t0
    best = w if best is None or w < best else best
out["stream_ms_per_frame"] = round(1e3 * best / (len(paths) - 1), 4)
out["stream_median_ms"] = round(float(np.median(tr.frame_ms)), 4)
out["stream_iterations"] = int(sum(tr.iterations))
print(json.dumps(out))
''' % (ROOT, ROOT)


def main():
    other = os.path.abspath(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import run_sequence
    drive = tempfile.mkdtemp(prefix="icpmi_ab_drive_")
    run_sequence.write_synthetic_drive(drive, 0, 60, workers=min(16, os.cpu_count() or 1))
    legs = []
    for rep in range(reps):
        for name, lib in (("other", other), ("product", "")):
            env = dict(os.environ, ICPMI_AB_DRIVE=drive, ICPMI_AB_LIB=lib)
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            if r.returncode != 0:
                print(r.stderr[-2000:], file=sys.stderr)
                return 1
            q = json.loads(r.stdout.strip().splitlines()[-1])
            q["leg"] = name
            legs.append(q)
            print(json.dumps(q), flush=True)
    keys = ("period_8k_us", "c3_call_ms", "stream_ms_per_frame", "stream_median_ms")
    summ = {n: {k: min(l[k] for l in legs if l["leg"] == n) for k in keys} for n in ("other", "product")}
    summ["bits_equal"] = all(len({l[k] for l in legs}) == 1 for k in ("hist60_tail", "c3_hist_tail", "stream_iterations"))
    print(json.dumps({"summary_min_over_legs": summ, "other": os.path.basename(other)}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
