#!/usr/bin/env python3
"""bench.py -- NMPC solves/sec on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one pass of the hot path over one batch: solve B = 4096 independent N = 20 NMPC problems
(BASELINE.json configs[2]: constant f_ext, 6-face tightened corridor per stage, cold start) with the
HIP solver, inputs already resident in HBM, outputs left in HBM.  Steps are issued round-robin on 2 HIP streams
(own workspace / outputs each) so that the few long solves at the end of one launch overlap the next launch;
config.single_stream_solves_per_s is the same measurement with strictly serial launches.  N GPUs -> each rank solves its own
4096-problem shard (different seed), no data-path collective ("weak").

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic flops of one interior-point iteration of one N=20, m=6 problem
F_STAGE = 31.5e3                 # flop per stage per iteration (textbook Schur-complement IPM definition)
ALG_BYTES_PER_SOLVE = 26456.0    # dense ABI image: params 23600 + output 2720 + info 136
FP64_PEAK_TFLOPS = 78.6          # MI355X FP64 vector == matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md


def usable_cores():
    """Host cores this process may actually use: min(cpu_count, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(w, seconds_target=12.0):
    """The CPU oracle (same algorithm, FP64, OpenMP over problems) timed on a bounded sample of the
    same workload on this box's host cores.  Test infrastructure used only as the reported baseline."""
    import tests.oracle_lib as OL
    cores = usable_cores()
    B = w["xinit"].shape[0]
    OL.solve_batch(w, nthreads=cores)  # warm up threads / page in
    reps, solved, t0 = 0, 0, time.perf_counter()
    conv = 0
    while True:
        z, fl, info = OL.solve_batch(w, nthreads=cores)
        reps += 1; solved += B; conv += int((fl == 1).sum())
        dt = time.perf_counter() - t0
        if dt >= seconds_target or reps >= 2000:
            break
    # the one piece of the reference's own hot path that runs without the ForcesPro licence: its CasADi model callback
    # (oracle/_ref, built from the reference sources in place), timed per stage call next to the port's stage evaluation
    anchor = None
    try:
        import ctypes
        so = os.path.join(OL.ORC_DIR, "_ref", "libref_model_normal.so")
        if os.path.exists(so):
            ref = ctypes.CDLL(so)
            fn = ctypes.cast(ref.FORCESNLPsolver_normal_casadi2forces, ctypes.c_void_p)
            a, b = ctypes.c_double(0), ctypes.c_double(0)
            OL.lib().orc_time_callback(fn, 0, 300000, ctypes.byref(a), ctypes.byref(b))
            anchor = {"reference_callback_ns_per_stage_call": a.value, "port_stage_eval_ns_per_call": b.value,
                      "what": "FORCESNLPsolver_normal_casadi2forces (reference, casadi2forces.c:42-245) vs orc_stage_eval, 1 thread"}
    except Exception as e:  # the anchor is optional evidence, never a reason to lose the bench line
        anchor = {"error": str(e)}
    n1 = min(B, 512)
    sub = {k: (v[:n1] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in w.items()}
    t1 = time.perf_counter(); OL.solve_batch(sub, nthreads=1); dt1 = time.perf_counter() - t1
    return dict(value=solved / dt, unit="solves/s", cores=cores, kind="port",
                sample=f"the same {B}-problem batch solved {reps}x by oracle/liboracle.so (FP64 CPU restatement of the "
                       f"same interior-point method, not ForcesPro: its binary is licence-locked), OpenMP over problems on "
                       f"{cores} threads (os.cpu_count()={os.cpu_count()}, affinity/cgroup-limited to {cores}), {dt:.1f} s; "
                       f"single-thread {n1 / dt1:.0f} solves/s",
                converged_frac=conv / solved, reference_anchor=anchor)


def pmc_traffic(batch):
    """HBM-side bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of this
    same command, calibrated on this solver's access width -- tools/collect_profiles.sh, tools/summarize_profiles.py),
    taken from the newest committed profiles/*_pmc_traffic.json whose batch size matches; None otherwise."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            t = json.load(open(f))
            if int(t["batch"]) == int(batch):
                return float(t["bytes_per_launch"]), os.path.basename(f)
        except Exception:
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued round-robin on (each with its own workspace and outputs); "
                         "1 = strictly back-to-back launches.  Default 2: the tail of one launch (a few long solves) overlaps "
                         "the head of the next; the strictly serial rate is measured as well and reported in config")
    args = ap.parse_args()

    import torch
    from forces_resilient_planner_amd import solver, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in the product)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("FRP_BENCH_FORCE_DIST"):  # (the env knob lets a 1-GPU box exercise the RCCL path)
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    B = args.batch
    w = workloads.config2(B, seed=workloads.SEED0 + 3 + 1000 * rank)
    ds = solver.DeviceSolver(B, w["N"], w["M"], 6, w["model"], f"cuda:{local_rank}")
    ds.upload(w)
    stream = torch.cuda.current_stream(dev)
    lanes = [(ds, stream)]
    for _ in range(1, max(1, args.streams)):  # further streams: own solver state / outputs, same resident inputs
        d2 = solver.DeviceSolver(B, w["N"], w["M"], 6, w["model"], f"cuda:{local_rank}")
        d2.xinit, d2.x0, d2.params, d2.nfaces = ds.xinit, ds.x0, ds.params, ds.nfaces
        lanes.append((d2, torch.cuda.Stream(dev)))
    torch.cuda.synchronize(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        lanes[i % len(lanes)][0].solve(lanes[i % len(lanes)][1])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        lanes[i % len(lanes)][0].solve(lanes[i % len(lanes)][1])
    barrier()
    elapsed = time.perf_counter() - t0
    # the same K steps strictly back to back on ONE stream (no overlap between launches), for reference
    t1 = time.perf_counter()
    for _ in range(args.steps):
        ds.solve(stream)
    barrier()
    elapsed_serial = time.perf_counter() - t1
    if dist is not None:
        t = torch.tensor([elapsed, elapsed_serial], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_serial = float(t[0].item()), float(t[1].item())

    fl = ds.exitflag.cpu().numpy(); it = ds.iters.cpu().numpy()
    stats = torch.tensor([float((fl == 1).sum()), float(it.sum()), float(B)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)  # 3 scalars of summary statistics, not on the data path
    conv_frac = float(stats[0] / stats[2]); mean_it = float(stats[1] / stats[2])

    # dominant kernel: average duration over the same launches, HIP events on the launch stream
    kernel_ms = ds.time_solve(max(1, args.steps), stream)
    torch.cuda.synchronize(dev)

    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        f_solve = mean_it * w["N"] * F_STAGE
        achieved_tf = B * f_solve / (kernel_ms * 1e-3) / 1e12
        achieved_gbs = B * ALG_BYTES_PER_SOLVE / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(B)
        out = {
            "metric": "NMPC solves/sec, batch=4096 horizons N=20", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: batch=4096 per GPU, N=20, constant f_ext~U[-3,3]^3, "
                                   "6-face tightened corridor per stage, cold start, reference 30-row parameter layout",
                       "batch_per_gpu": B, "horizon": int(w["N"]), "converged_frac": conv_frac,
                       "mean_ipm_iterations": mean_it, "streams": len(lanes),
                       "single_stream_solves_per_s": world * B * args.steps / elapsed_serial,
                       "single_stream_ms_per_step": elapsed_serial / args.steps * 1e3, "p95_ipm_iterations": float(np.percentile(it, 95)),
                       "max_ipm_iterations": int(it.max()), "tolerances": 1e-4},
            "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "nmpc_ipm_kernel", "kernel_ms": kernel_ms,
                         "flops_per_launch": B * f_solve,
                         "note": "FP64 (vector == matrix peak 78.6 TFLOP/s); flops = SURVEY 8d definition "
                                 "mean_it * N * 31.5 kflop per solve",
                         "hbm_algorithmic": {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": achieved_gbs / HBM_PEAK_GBS}},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        elif not args.no_cpu:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
