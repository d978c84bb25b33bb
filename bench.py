#!/usr/bin/env python3
"""bench.py -- ICP iterations/sec on the 100k -> 100k synthetic scan (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one ICP iteration (icp.hpp:181-232: correspondence search, residual + 6x6
normal-equation reduction, solve, transform).  The timed region is ONE icpmi_align_device
call with max_iterations = K, tolerance = min_error = 0 (so exactly K iterations run,
icp.hpp:210,214 cannot fire), inputs already resident in HBM.  It therefore also contains
what the reference pays on every call: normal estimation of the target (icp.hpp:169-171)
and the post-loop evaluation pass (icp.hpp:235-252).  One such call is ~10 ms, too short for
one sample to be a robust headline (a single host stall would move it by half), so the call
is timed `--repeats` times (default 7), each sample bracketed by barrier + synchronize on
both sides, and `value` = K / the MEDIAN call time; min / max are in `call_ms`.
`ms_per_step` x K is still the time of one call.  The loop-only rate is reported beside it
as `steady_state_it_per_s`.

With N > 1 ranks the source cloud is sharded N ways (strong scaling of the same 100k ->
100k job); the 30-double all-reduce per iteration runs over RCCL inside the library.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector = FP32 MFMA (v_mfma_f32_16x16x4_f32)
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (the pipe k_nn_coarse runs on)
FLOP_PER_PAIR = 8.0        # 3 sub + 1 mul + 2 fma


_CPU_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
cpu, points, steps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.sched_setaffinity(0, {cpu})                       # before the oracle is loaded or called: one core, no migration
import numpy as np
from lidar_slam_from_scratch_amd import synth
from oracle import oracle as orc
src, tgt, _ = synth.c3_uniform(points)
t0 = time.perf_counter()
r = orc.icp_point_to_plane(src, tgt, max_iterations=steps, tolerance=0.0, min_error=0.0, faithful=True, nthreads=1)
wall = time.perf_counter() - t0
print(json.dumps({"wall_s": wall, "loops": int(r.loop_iterations), "loop_s": r.loop_seconds, "setup_s": r.setup_seconds,
                  "final_s": r.final_seconds, "affinity": sorted(os.sched_getaffinity(0))}))
"""


def cpu_baseline_pinned(points, steps, samples=3):
    """the 1-thread leg PINNED to one core (os.sched_setaffinity in a child process, before the oracle is
    called), `samples` separate runs, the median reported with every sample beside it."""
    import subprocess
    allowed = sorted(os.sched_getaffinity(0))
    cpu = allowed[len(allowed) // 2]   # (not core 0, where the kernel's housekeeping tends to land)
    runs = []
    for _ in range(samples):
        r = subprocess.run([sys.executable, "-c", _CPU_CHILD, ROOT, str(cpu), str(points), str(steps)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError("pinned CPU baseline child failed: " + r.stderr[-400:])
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    rates = [max(x["loops"], 1) / x["wall_s"] for x in runs]
    med = runs[int(np.argsort(rates)[len(rates) // 2])]
    return {"value": float(np.median(rates)), "samples_it_per_s": rates, "pinned_cpu": cpu, "affinity_seen_by_child": med["affinity"],
            "loops": med["loops"], "steady_state_it_per_s": max(med["loops"], 1) / med["loop_s"],
            "setup_s": med["setup_s"], "loop_s": med["loop_s"], "final_s": med["final_s"]}


def cpu_baseline(src, tgt, steps):
    """The oracle (CPU restatement of the reference: kd-tree, two NN passes per iteration,
    single thread like the reference) on the same workload.  Timed as the checker's
    baseline only -- never part of the measured GPU path.  The 1-thread figure is the median of three
    runs pinned to one core (cpu_baseline_pinned); the run below, in this process, supplies the result
    the GPU's pose is compared with."""
    from oracle import oracle as orc
    pinned = cpu_baseline_pinned(src.shape[0], steps)
    t0 = time.time()
    r = orc.icp_point_to_plane(src, tgt, max_iterations=steps, tolerance=0.0, min_error=0.0,
                               faithful=True, nthreads=1)
    wall = time.time() - t0
    loops = max(r.loop_iterations, 1)
    # the same restatement with the queries partitioned over every host core: NOT the
    # reference (which is single-threaded); reported for fairness
    nth = os.cpu_count() or 1
    t0 = time.time()
    ra = orc.icp_point_to_plane(src, tgt, max_iterations=steps, tolerance=0.0, min_error=0.0,
                                faithful=False, nthreads=nth)
    wall_all = time.time() - t0
    return {
        "all_cores_not_the_reference": {"value": max(ra.loop_iterations, 1) / wall_all, "cores": nth,
                                        "note": "oracle, deduplicated NN pass, std threads over queries"},
        "value": pinned["value"], "unit": "ICP iterations/s", "cores": 1, "kind": "port",
        "pinned_cpu": pinned["pinned_cpu"], "samples_it_per_s": pinned["samples_it_per_s"],
        "sample": "same C3 %d->%d pair, one full call of %d iterations (kd-tree build + 20-NN normals + loop + final "
                  "pass), 1 thread pinned to cpu %d (sched_setaffinity in a child process), median of %d runs"
                  % (src.shape[0], tgt.shape[0], loops, pinned["pinned_cpu"], len(pinned["samples_it_per_s"])),
        "unpinned_in_process_it_per_s": loops / wall,
        "steady_state_it_per_s": pinned["steady_state_it_per_s"],
        "setup_s": pinned["setup_s"], "loop_s": pinned["loop_s"], "final_s": pinned["final_s"],
        "result": r,
    }


def measure_traffic(points, search, timeout_s=75):
    """HBM bytes per launch of the dominant kernel (k_nn_coarse_bounded: the all-pairs pass of every ICP iteration),
    measured in this run: two
    `rocprofv3 --kernel-trace --pmc <counter>` passes in child processes -- FETCH_SIZE and WRITE_SIZE
    do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots) -- over scripts/run_align_once.py
    on the same workload.  Both counters come in KiB.  gfx950 correction from the same guide:
    FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) coalesced reads at 64 bytes, so
    the fetch side is doubled (an upper bound here: the operand stream is 16 B per lane, the fp64
    query loads are not); WRITE_SIZE is taken as it reads.  Returns a dict, or None when rocprofv3
    is not on this machine or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None:
        return None
    kib = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="icpmi_pmc_", dir="/tmp")
        cmd = [rocprof, "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", out_dir, "--",
               sys.executable, os.path.join(ROOT, "scripts", "run_align_once.py"), str(search), str(points), "6", "2"]
        # bounded: the headline must not wait long for the profiler (a pass takes ~15 s, most of it the child's
        # `import torch`); a pass that does not finish leaves traffic = null.  The pass runs in a session of its own
        # and a timeout ends the whole process GROUP: killing the rocprofv3 wrapper alone would leave the profiled
        # child on the GPU behind the headline
        import signal
        try:
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                                    stderr=subprocess.STDOUT, start_new_session=True)
        except Exception:  # noqa: BLE001
            shutil.rmtree(out_dir, ignore_errors=True)
            return None
        try:
            proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.wait()
            shutil.rmtree(out_dir, ignore_errors=True)
            return None
        r = proc
        vals, first = [], []
        for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                # the pass kernel of all but a call's first pass (k_nn_coarse_bounded); a build or setting without it
                # (ICPMI_NN_BOUNDED=0) leaves only k_nn_coarse<0,..>
                if "k_nn_coarse_bounded" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
                elif "k_nn_coarse<0" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    first.append(float(row["Counter_Value"]))
        vals = vals or first
        shutil.rmtree(out_dir, ignore_errors=True)
        if r.returncode != 0 or not vals:
            return None
        kib[counter] = sum(vals) / len(vals)
    return {"hbm_bytes_per_launch": int((2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024.0),
            "FETCH_SIZE_KiB": kib["FETCH_SIZE"], "WRITE_SIZE_KiB": kib["WRITE_SIZE"],
            "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over scripts/run_align_once.py, "
                   "mean over the k_nn_coarse_bounded launches (every pass of a call); fetch doubled (gfx950 "
                   "wide-read correction)"}


def self_launch(n):
    """Run this script under torch.distributed.run with n ranks on 127.0.0.1; returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--search", type=int, default=2,
                    help="the engine `value` is measured with: 2 (default) bf16 MFMA over ALL pairs + exact fp64 resolve -- the "
                         "north_star's brute force, comparable across rounds; 1 exact fp64; 3 the culled engine; 0 whatever "
                         "AUTO picks.  The library's default engine (AUTO) is always measured beside it: `default_engine`")
    ap.add_argument("--no-default-engine", action="store_true",
                    help="skip the second timed series on the library's default engine (AUTO: the culled search at this size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 counter passes behind roofline.traffic")
    ap.add_argument("--repeats", type=int, default=7,
                    help="timed calls of --steps iterations each; value = steps / the median call time")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: take the multi-rank code path (process group, RCCL communicator "
                         "inside the library, sharded kernels) with a world of 1")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--force-fallback-exchange", action="store_true",
                    help="rehearsal: pretend the library's RCCL communicator failed and exchange through torch.distributed")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N-rank run where RCCL cannot form a communicator (several ranks sharing "
                         "one GPU): process group over gloo, the library's two exchanges through host callbacks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process
        # per GPU.  This parent makes no GPU call (it does not even import torch): it only waits
        # for the ranks; rank 0 prints the one JSON line on the stdout it inherits.
        sys.exit(self_launch(args.gpus))

    # stdout carries exactly one line, the JSON: whatever a library prints there (RCCL announces
    # its version on stdout when a communicator is created) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from lidar_slam_from_scratch_amd import capi, dist as icpdist, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d "
                         "(or with no launcher at all: bench.py starts its own ranks)" % (args.gpus, world, args.gpus))
    if args.rehearse_gloo:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    src, tgt, _T = synth.c3_uniform(args.points)
    lo, hi = icpdist.shard_bounds(src.shape[0], world, rank)
    d_src = torch.from_numpy(np.ascontiguousarray(src[lo:hi])).to(dev)
    d_tgt = torch.from_numpy(tgt).to(dev)
    torch.cuda.synchronize()
    value_search = args.search

    state = {"ctx": None, "exchange": None}

    def open_ctx(search, profile):
        """A context on `search`; in a multi-rank run with the library's own communicator (RCCL), or -- every rank agreeing
        through one all-reduce -- the fallback exchange through torch's collectives."""
        if state["ctx"] is not None:
            if dist is not None:
                state["ctx"].comm_finalize()
            state["ctx"].close()
        ctx = capi.Context(device=local_rank, search=search, profile=profile)
        state["ctx"] = ctx
        if dist is None:
            return ctx
        if args.rehearse_gloo:
            icpdist.init_callbacks(ctx, dist)
            return ctx
        ok = 1.0
        try:
            if args.force_fallback_exchange:
                raise capi.IcpError(capi.ERR_RCCL, "--force-fallback-exchange")
            icpdist.init_rccl(ctx, dist, device=dev, allow_single=args.force_dist)
        except capi.IcpError as e:
            sys.stderr.write("rank %d: library communicator failed (%s)\n" % (rank, e))
            ok = 0.0
        agreed = torch.tensor([ok], dtype=torch.float64, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if float(agreed.item()) < 1.0:
            state["exchange"] = ("torch.distributed collectives staged through device tensors (the library's RCCL "
                                 "communicator could not be formed)")
            ctx.comm_finalize()
            icpdist.init_callbacks(ctx, dist, device=dev)
        return ctx

    def call(iters):
        cfg = capi.Context.make_config(max_iterations=iters, tolerance=0.0, min_error=0.0)
        return state["ctx"].align_device(d_src.data_ptr(), hi - lo, d_tgt.data_ptr(), tgt.shape[0], cfg)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_calls(repeats):
        """`repeats` samples of ONE call of args.steps iterations, each bracketed by barrier +
        synchronize on both sides, the MAX over ranks taken per sample -> (last result, times)"""
        times, out = [], None
        for _ in range(max(1, repeats)):
            fence()
            t0 = time.perf_counter()
            out = call(args.steps)
            fence()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64, device="cpu" if args.rehearse_gloo else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            times.append(el)
        return out, times

    def series(search):
        """warm-up call, then `--repeats` timed calls on a fresh context of engine `search` (profile level 1: HIP events
        around the call, the loop, every 4th launch of the dominant kernel and every 4th all-reduce; the fp64 engine has
        no single dominant kernel: level 2 times its whole search pass)"""
        ctx = open_ctx(search, 2 if search == 1 else 1)
        if args.warmup > 0:
            call(args.warmup)
        ctx.reset_profile()
        (res, hist), times = timed_calls(args.repeats)
        return res, hist, times, ctx.get_profile()

    def comm_block(prof):
        """Who the communicator says took part (icpmi_comm_info: ncclCommCount / UserRank / CuDevice, every rank's device
        and PCI bus id gathered through the library's own all-gather) and what one all-reduce of a pass costs."""
        if dist is None:
            return None
        info = state["ctx"].comm_info()   # (a collective: every rank calls it)
        n_ex = int(prof["exchange_launches"])
        return {"kind": info["kind"], "n_ranks": info["n_ranks"], "rank_of_reporter": info["rank"],
                "devices": info["devices"], "pci_bus_ids": info["pci_bus_ids"],
                "distinct_gpus": len(set(info["pci_bus_ids"])),
                "allreduce_ms_per_pass": prof["exchange_ms"] / n_ex if n_ex else None,
                "allreduces_timed": n_ex,
                "how": "HIP events on the library's stream around ncclAllReduce of 30 doubles, every 4th pass of the timed calls"}

    def culled_roofline(prof, n_local, m):
        """The default engine's dominant kernel: the coarse pass over the (64-row group, split) pairs within reach.
        Its work is the EXECUTED pairs (what the box test leaves), priced like the all-pairs kernel's."""
        if not prof["coarse_launches"] or not prof["bounded_launches"]:
            return None
        k_ms = prof["coarse_ms"] / prof["coarse_launches"]
        pairs_run = prof["nn_group_pairs_run"] * 32.0 * 2048.0 / max(1, prof["bounded_launches"])   # per launch (32-row tile x 2048-target split)
        pairs_all = float(n_local) * float(m)
        ach = FLOP_PER_PAIR * pairs_run / (k_ms * 1e-3) / 1e12
        return {"kernel": "k_nn_coarse_groups (bf16 MFMA over the (32-row tile, 2048-target split) pairs whose boxes are "
                          "within the tile's bound of each other)",
                "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                "traffic": None, "avg_launch_ms": k_ms, "launches_timed": int(prof["coarse_launches"]),
                "executed_pairs_per_launch": pairs_run, "all_pairs_per_launch": pairs_all,
                "pairs_culled_frac": 1.0 - prof["nn_group_pairs_run"] / max(1, prof["nn_group_pairs"]),
                "flop_per_launch": FLOP_PER_PAIR * pairs_run,
                "equivalent_all_pairs_tflops": FLOP_PER_PAIR * pairs_all / (k_ms * 1e-3) / 1e12,
                "note": "achieved = 8 flop x EXECUTED pairs / the kernel's average launch; equivalent_all_pairs_tflops prices the "
                        "same launch as if it had evaluated every pair (what it makes unnecessary)"}

    res, hist, call_times, prof = series(value_search)
    elapsed = float(np.median(call_times))
    assert res.loop_iterations == args.steps, (res.loop_iterations, args.steps)
    comm = comm_block(prof)
    # per-stage breakdown from a second, untimed call with every stage bracketed by events.
    # Single-GPU runs only: the multi-rank line carries nothing that needs a second communicator.
    stage = None
    if dist is None:
        ctx = open_ctx(value_search, 2)
        call(args.steps)
        ctx.reset_profile()
        call(args.steps)
        stage = ctx.get_profile()

    # The library's DEFAULT engine (ICPMI_SEARCH_AUTO: the culled search on a target of this size) on the same job, same
    # timing protocol.  Reported beside `value` (the all-pairs engine, comparable with earlier rounds), with its own
    # roofline on the pairs it executes; same correspondences, so history and pose must be the all-pairs run's.
    default_engine = None
    if not args.no_default_engine and value_search != 0:
        try:
            dres, dhist, dtimes, dprof = series(capi.SEARCH_AUTO)
            del_ = float(np.median(dtimes))
            n_local, m = hi - lo, tgt.shape[0]
            culled = dprof["nn_group_pairs"] > 0
            default_engine = {
                "search": "ICPMI_SEARCH_AUTO -> " + ("culled MFMA search (nn_culled.h)" if culled else "the same engine as `value`"),
                "value": args.steps / del_, "unit": "ICP iterations/s", "ms_per_step": 1e3 * del_ / args.steps,
                "call_ms": {"median": 1e3 * del_, "min": 1e3 * min(dtimes), "max": 1e3 * max(dtimes), "samples": len(dtimes)},
                "steady_state_it_per_s": (args.steps + 1) * len(dtimes) / (dprof["loop_ms"] * 1e-3) if dprof["loop_ms"] > 0 else None,
                "vs_value": (args.steps / del_) / (args.steps / elapsed),
                "roofline": culled_roofline(dprof, n_local, m),
                "num_iterations_equal": bool(dres.num_iterations == res.num_iterations),
                "pose_delta_vs_all_pairs": list(synth.pose_delta(np.array(dres.transformation[:]).reshape(4, 4),
                                                                 np.array(res.transformation[:]).reshape(4, 4))),
                "history_max_abs_diff": float(np.abs(np.asarray(dhist) - np.asarray(hist)).max()),
                "history_bit_equal": bool(np.array_equal(np.asarray(dhist), np.asarray(hist))),
                "comm": comm_block(dprof),
                "note": "same correspondences as the all-pairs pass (the reference's own search culls too: kdtree.hpp:139,177); "
                        "(32-row tile, split) pairs culled by a bounding-box test against each row's bound -- the previous "
                        "match's exact distance, in a call's first pass the nearest sorted target around the row's Morton place"}
        except Exception as e:  # noqa: BLE001  (the headline must not depend on the second series)
            default_engine = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        n_local, m = hi - lo, tgt.shape[0]
        mfma = prof["coarse_launches"] > 0
        culled_value = prof["nn_group_pairs"] > 0   # (--search 0 or 3: `value` itself is the culled engine)
        # dominant kernel: the coarse pass (bf16 MFMA engines) or the whole fp64 pass (exact engine)
        k_ms = prof["coarse_ms"] / prof["coarse_launches"] if mfma else prof["nn_ms"] / max(prof["nn_launches"], 1)
        nn_avg_ms = stage["nn_ms"] / max(stage["nn_launches"], 1) if stage else None
        flops = FLOP_PER_PAIR * n_local * m
        achieved = flops / (k_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if mfma else PEAK_FP32_TFLOPS
        # measured now, in child processes (this process keeps its own GPU context; the children are
        # started, not exec'ed): see measure_traffic
        traffic_info = None
        if mfma and not culled_value and dist is None and not args.no_traffic:
            state["ctx"].close()
            state["ctx"] = None
            traffic_info = measure_traffic(args.points, value_search)
        traffic = traffic_info["hbm_bytes_per_launch"] if traffic_info else None
        algo_bytes = 24 * n_local + 24 * m + 4 * n_local
        if culled_value:
            roofline = culled_roofline(prof, n_local, m)
        else:
            roofline = {
                "kernel": "k_nn_coarse_bounded (bf16 MFMA, all %dx%d pairs; every pass of the call)" % (n_local, m)
                if mfma else "k_nn_f64",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_measurement": traffic_info,
                "launches_timed": int(prof["coarse_launches"]) if mfma else int(prof["nn_launches"]),
                "flop_per_launch": flops, "avg_launch_ms": k_ms,
                # what the pipe actually executes: one 32x32x16 MFMA (32,768 flop) per 1024 pairs
                "executed_mfma_tflops": (n_local * m / 1024.0) * 32768.0 / (k_ms * 1e-3) / 1e12 if mfma else None,
                "achieved_vs_fp32_vector_peak": achieved / PEAK_FP32_TFLOPS,
                "nn_pass_ms": nn_avg_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
                # the north_star's "achieved HBM GB/s": from the MEASURED bytes when the counter passes ran,
                # the algorithmic figure (24N + 24M + 4N) beside it under its own name
                "achieved_hbm_GBps": (traffic if traffic else algo_bytes) / (k_ms * 1e-3) / 1e9,
                "achieved_hbm_GBps_basis": "measured traffic" if traffic else "algorithmic bytes (no counter pass)",
                "algorithmic_hbm_GBps": algo_bytes / (k_ms * 1e-3) / 1e9,
                "hbm_peak_GBps": 8000.0,
            }
        out = {
            "metric": "ICP iterations/sec (100k->100k pts)",
            "value": args.steps / elapsed,
            "unit": "ICP iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "call_ms": {"median": 1e3 * elapsed, "min": 1e3 * min(call_times), "max": 1e3 * max(call_times),
                        "samples": len(call_times),
                        "note": "one sample = one icpmi_align_device call of `steps` iterations; value = steps / median"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C3: synthetic %d->%d-pt uniform random scan, %d ICP iterations, "
                                   "1 call incl. 20-NN normals + final pass" % (src.shape[0], m, args.steps),
                       "source_points": int(src.shape[0]), "target_points": int(m),
                       "parallelism": ("source sharded x%d, REHEARSAL (not a scaling number): ranks share GPUs, "
                                       "exchange through gloo host callbacks" % world) if args.rehearse_gloo
                       else "source sharded x%d, 30-double all-reduce/iter: %s" % (world, state["exchange"] or "RCCL inside the library")
                       if dist is not None else "single GPU",
                       "search": ("culled bf16 MFMA search (nn_culled.h) + exact fp64 resolve" if culled_value else
                                  "bf16 MFMA coarse pass over all pairs + exact fp64 resolve of the slots listed under each row's bound "
                                  "(the previous match's distance; in the first pass the nearest sorted target around the row's Morton place)")
                       if mfma else "exact fp64 brute force"},
            "steady_state_it_per_s": (args.steps + 1) * len(call_times) / (prof["loop_ms"] * 1e-3) if prof["loop_ms"] > 0 else None,
            "stage_ms_untimed_call": {k: stage[k] for k in ("nn_ms", "coarse_ms", "reduce_ms", "transform_ms",
                                                            "normals_ms", "setup_ms", "loop_ms", "total_ms")}
            if stage else None,
            "resolve_counters": {k: stage[k] for k in ("nn_recheck_queries", "nn_fallback_queries", "knn_fallback_rows")}
            if stage else None,
            "final_error": res.final_error,
            "rccl": comm,
            "default_engine": default_engine,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and dist is None:
            cb = cpu_baseline(src, tgt, args.cpu_steps or args.steps)
            r = cb.pop("result")
            T = np.array(res.transformation[:]).reshape(4, 4)
            if r.loop_iterations == args.steps:
                dt, dr = synth.pose_delta(T, r.transformation)
                out["parity"] = {"pose_dt_m": dt, "pose_dr_rad": dr,
                                 "num_iterations_equal": bool(r.num_iterations == res.num_iterations),
                                 "final_error_abs_diff": abs(r.final_error - res.final_error)}
                # correspondence mismatches of one search pass (first-iteration geometry),
                # GPU through the C ABI vs the oracle's kd-tree, and the normals of the target
                from oracle import oracle as orc
                nth = os.cpu_count() or 1
                ctx = open_ctx(value_search, 0)   # the engine `value` was measured with
                gi, gd = ctx.nearest_batch(tgt, src)
                oi, od = orc.KDTree(tgt).nearest_batch(src, nthreads=nth)
                out["parity"]["nn_index_mismatches"] = int((gi != oi).sum())
                out["parity"]["nn_sqdist_mismatches"] = int((gd != od).sum())
                out["parity"]["normal_mismatches"] = int(
                    (ctx.estimate_normals(tgt, 20) != orc.estimate_normals(tgt, None, 20, nthreads=nth)).any(axis=1).sum())
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_1thread"] = out["value"] / cb["value"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if state["ctx"] is not None:
        if dist is not None:
            state["ctx"].comm_finalize()
        state["ctx"].close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
