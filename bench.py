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


def cpu_baseline(src, tgt, steps):
    """The oracle (CPU restatement of the reference: kd-tree, two NN passes per iteration,
    single thread like the reference) on the same workload.  Timed as the checker's
    baseline only -- never part of the measured GPU path."""
    from oracle import oracle as orc
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
        "value": loops / wall, "unit": "ICP iterations/s", "cores": 1, "kind": "port",
        "sample": "same C3 100k->100k pair, one full call of %d iterations (kd-tree build + "
                  "20-NN normals + loop + final pass), 1 thread" % loops,
        "steady_state_it_per_s": loops / r.loop_seconds,
        "setup_s": r.setup_seconds, "loop_s": r.loop_seconds, "final_s": r.final_seconds,
        "result": r,
    }


def measure_traffic(points, search, timeout_s=75):
    """HBM bytes per launch of the dominant kernel (k_nn_coarse_bounded: the all-pairs pass of every ICP iteration but
    a call's first, which is k_nn_coarse<0>), measured in this run: two
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
        try:
            # bounded: the headline must not wait long for the profiler (a pass takes ~15 s, most of it
            # the child's `import torch`); a pass that does not finish leaves traffic = null
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, timeout=timeout_s)
        except Exception:  # noqa: BLE001
            shutil.rmtree(out_dir, ignore_errors=True)
            return None
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
                   "mean over the k_nn_coarse_bounded launches (every pass of a call but its first); fetch doubled (gfx950 "
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
    ap.add_argument("--search", type=int, default=0,
                    help="0 auto (= 2 at this size), 1 exact fp64, 2 bf16 MFMA over all pairs + certified resolve, "
                         "3 the same with box culling of (query block, target split) units")
    ap.add_argument("--no-pruned-extra", action="store_true",
                    help="skip the extra, untimed-by-the-contract run of the opt-in pruned engine")
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

    # profile level 1: HIP events around the dominant kernel and the call/loop only
    # (the fp64 engine has no single dominant kernel bracket: time its whole NN pass instead)
    ctx = capi.Context(device=local_rank, search=args.search, profile=2 if args.search == 1 else 1)
    exchange = None
    if dist is not None:
        if args.rehearse_gloo:
            icpdist.init_callbacks(ctx, dist)
        else:
            # the library's own RCCL communicator; if it cannot be formed on some rank, every rank
            # (they agree through one all-reduce) falls back to exchanging through torch's collectives
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
                exchange = "torch.distributed collectives staged through device tensors (the library's RCCL communicator could not be formed)"
                ctx.comm_finalize()
                icpdist.init_callbacks(ctx, dist, device=dev)

    def call(iters):
        cfg = capi.Context.make_config(max_iterations=iters, tolerance=0.0, min_error=0.0)
        return ctx.align_device(d_src.data_ptr(), hi - lo, d_tgt.data_ptr(), tgt.shape[0], cfg)

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

    if args.warmup > 0:
        call(args.warmup)
    ctx.reset_profile()
    (res, hist), call_times = timed_calls(args.repeats)
    elapsed = float(np.median(call_times))
    prof = ctx.get_profile()
    assert res.loop_iterations == args.steps, (res.loop_iterations, args.steps)
    # per-stage breakdown from a second, untimed call with every stage bracketed by events.
    # Single-GPU runs only: the multi-rank line carries nothing that needs a second communicator.
    stage = None
    if dist is None:
        ctx.close()
        ctx = capi.Context(device=local_rank, search=args.search, profile=2)
        call(args.steps)
        ctx.reset_profile()
        call(args.steps)
        stage = ctx.get_profile()

    # extra: the opt-in pruned engine (ICPMI_SEARCH_MFMA_PRUNED) on the same job, same timing
    # protocol.  Reported beside `value`, never as `value`: it is not an all-pairs pass.
    pruned = None
    if dist is None and not args.no_pruned_extra and args.search in (0, 2):
        ctx.close()
        ctx = capi.Context(device=local_rank, search=capi.SEARCH_MFMA_PRUNED, profile=1)
        if args.warmup > 0:
            call(args.warmup)
        ctx.reset_profile()
        (pres, phist), ptimes = timed_calls(args.repeats)
        pel = float(np.median(ptimes))
        pp = ctx.get_profile()
        pruned = {"value": args.steps / pel, "unit": "ICP iterations/s", "ms_per_step": 1e3 * pel / args.steps,
                  "steady_state_it_per_s": (args.steps + 1) * len(ptimes) / (pp["loop_ms"] * 1e-3) if pp["loop_ms"] > 0 else None,
                  "units_culled_frac": pp["nn_pruned_blocks"] / max(1, pp["nn_coarse_blocks"]),
                  "coarse_avg_launch_ms": pp["coarse_ms"] / max(1, pp["coarse_launches"]),
                  "pose_delta_vs_all_pairs": list(synth.pose_delta(np.array(pres.transformation[:]).reshape(4, 4),
                                                                   np.array(res.transformation[:]).reshape(4, 4))),
                  "history_max_abs_diff": float(np.abs(np.asarray(phist) - np.asarray(hist)).max()),
                  "note": "opt-in ICPMI_SEARCH_MFMA_PRUNED: same correspondences, (512-query block, 2048-target "
                          "split) units culled by a bounding-box test against the previous iteration's distances"}

    if rank == 0:
        n_local, m = hi - lo, tgt.shape[0]
        mfma = prof["coarse_launches"] > 0
        # dominant kernel: k_nn_coarse (bf16 MFMA engine) or the whole fp64 pass (exact engine)
        k_ms = prof["coarse_ms"] / prof["coarse_launches"] if mfma else prof["nn_ms"] / max(prof["nn_launches"], 1)
        nn_avg_ms = stage["nn_ms"] / max(stage["nn_launches"], 1) if stage else None
        flops = FLOP_PER_PAIR * n_local * m
        achieved = flops / (k_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if mfma else PEAK_FP32_TFLOPS
        # measured now, in child processes (this process keeps its own GPU context; the children are
        # started, not exec'ed): see measure_traffic
        traffic_info = None
        if mfma and dist is None and not args.no_traffic:
            ctx.close()
            traffic_info = measure_traffic(args.points, args.search)
            ctx = capi.Context(device=local_rank, search=args.search)
        traffic = traffic_info["hbm_bytes_per_launch"] if traffic_info else None
        algo_bytes = 24 * n_local + 24 * m + 4 * n_local
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
                       else "source sharded x%d, 30-double all-reduce/iter: %s" % (world, exchange or "RCCL inside the library")
                       if dist is not None else "single GPU",
                       "search": "bf16 MFMA coarse pass over all pairs + exact fp64 resolve (certified after the pass in a call's "
                                 "first iteration, behind the previous matches' distances from the second on)" if mfma
                       else "exact fp64 brute force"},
            "steady_state_it_per_s": (args.steps + 1) * len(call_times) / (prof["loop_ms"] * 1e-3) if prof["loop_ms"] > 0 else None,
            "stage_ms_untimed_call": {k: stage[k] for k in ("nn_ms", "coarse_ms", "reduce_ms", "transform_ms",
                                                            "normals_ms", "setup_ms", "loop_ms", "total_ms")}
            if stage else None,
            "resolve_counters": {k: stage[k] for k in ("nn_recheck_queries", "nn_fallback_queries", "knn_fallback_rows")}
            if stage else None,
            "final_error": res.final_error,
            "pruned_engine_extra": pruned,
            "roofline": {
                "kernel": "k_nn_coarse_bounded (k_nn_coarse<0> in a call's first pass; bf16 MFMA, all %dx%d pairs)" % (n_local, m)
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
            },
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
                ctx.close()
                ctx = capi.Context(device=local_rank, search=args.search)   # the engine `value` was measured with
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
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
