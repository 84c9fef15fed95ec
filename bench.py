#!/usr/bin/env python3
"""bench.py -- NMPC solves/sec on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one pass of the hot path over one batch: solve B = 4096 independent N = 20 NMPC problems
(BASELINE.json configs[2]: constant f_ext, 6-face tightened corridor per stage, cold start) with the
HIP solver, inputs already resident in HBM, outputs left in HBM, strictly serial launches on one stream; the K-step
region is timed 5 times and the median reported (config.pipelined_solves_per_s: the same steps round-robin on 2 streams,
informational).  N GPUs -> each rank solves its own 4096-problem shard (different seed), no data-path collective ("weak").
--scaling strong / --config 3 / --config 4: the other BASELINE configs and the scatter -> solve -> gather form of SURVEY 8e.

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


def native_oracle():
    """The CPU baseline runs the oracle compiled FOR THIS HOST (-O3 -march=native, OpenMP), as SURVEY 8d asks: the in-tree
    oracle/liboracle.so is built in the build container for a generic x86-64-v3 target.  Built into a temp dir at bench
    time (gcc is in the image); falls back to the in-tree library if that fails."""
    import ctypes, subprocess, tempfile
    import tests.oracle_lib as OL
    try:
        d = tempfile.mkdtemp(prefix="frp_oracle_native_")
        so = os.path.join(d, "liboracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-shared", os.path.join(OL.ORC_DIR, "nmpc_model.c"),
                               os.path.join(OL.ORC_DIR, "nmpc_ipm.c"), "-o", so, "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.orc_solve.restype = ctypes.c_int
        OL._lib = lib
        return "-O3 -march=native (built on this host)"
    except Exception:
        return "-O3 -march=x86-64-v3 (in-tree build)"


def cpu_baseline(w, seconds_target=12.0, diverge_mu=1e3, mu0=1.0):
    """The CPU oracle (same algorithm, FP64, OpenMP over problems) timed on a bounded sample of the
    same workload on this box's host cores.  Test infrastructure used only as the reported baseline."""
    import tests.oracle_lib as OL
    flags = native_oracle()
    cores = usable_cores()
    B = w["xinit"].shape[0]
    oopt = OL.default_options(diverge_mu=diverge_mu, mu0=mu0)
    OL.solve_batch(w, oopt, nthreads=cores)  # warm up threads / page in
    reps, solved, t0 = 0, 0, time.perf_counter()
    conv = 0
    while True:
        z, fl, info = OL.solve_batch(w, oopt, nthreads=cores)
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
    t1 = time.perf_counter(); OL.solve_batch(sub, oopt, nthreads=1); dt1 = time.perf_counter() - t1
    cpu_model = ""
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return dict(value=solved / dt, unit="solves/s", cores=cores, kind="port",
                sample=f"the same {B}-problem batch solved {reps}x by the oracle (FP64 CPU restatement of the "
                       f"same interior-point method, not ForcesPro: its binary is licence-locked), compiled {flags}, OpenMP over problems on "
                       f"{cores} threads (os.cpu_count()={os.cpu_count()}, affinity/cgroup-limited to {cores}; {cpu_model}), {dt:.1f} s; "
                       f"single-thread {n1 / dt1:.0f} solves/s",
                single_thread_solves_per_s=n1 / dt1, converged_frac=conv / solved, reference_anchor=anchor)


def pmc_traffic(batch, kernel):
    """HBM-side bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of this
    same command, calibrated on this solver's access width -- tools/collect_r02.sh, tools/summarize_profiles.py),
    taken from the newest committed profiles/*_pmc_traffic.json of the same kernel and batch size; None otherwise."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            t = json.load(open(f))
            if int(t["batch"]) == int(batch) and (kernel + "<") in t.get("kernel", ""):
                return float(t["bytes_per_launch"]), os.path.basename(f)
        except Exception:
            pass
    return None, None


def self_launch_command(n, argv, port=None):
    """The command line `python bench.py --gpus N ...` re-executes itself with when no launcher set WORLD_SIZE."""
    port = port or int(os.environ.get("FRP_BENCH_PORT", "0")) or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


XGMI_LINK_GBS = 153.0  # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)


def device_cus(default=256):
    """CUs of the current device from the runtime's device properties (hipDeviceProp_t::multiProcessorCount through torch), not a constant;
    `default` only where no device is visible (--dry-run on a CPU box), and the plan says which it was."""
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count), "hipDeviceProp"
    except Exception:  # noqa: BLE001
        pass
    return default, "assumed (no device visible)"


def auto_chunks(world, B, N, MF, cus=None):
    """--chunks 0 of the strong-scaling step: pieces per shard and the reason (logged).  A piece below two rounds of resident workgroups
    costs more than its transfer can hide (profiles/r04_strong_chunks.txt: 4 pieces +46 % at configs[2] on one GPU), and on one rank
    nothing is transferred."""
    if cus is None:
        cus = device_cus()[0]
    # resident workgroups per CU of the variant the launch takes (round 6: horizons 20 < N <= 30 with <= 16 rows run three per CU from 7 x CUs problems on)
    per_cu = 4 if (N <= 20 and MF <= 6) else (3 if N <= 20 else ((3 if (N <= 30 and MF <= 16 and B > 7 * cus) else 2) if N <= 32 else 1))
    slots = cus * per_cu
    if world <= 1:
        return 1, "one rank: nothing to overlap"
    if B >= 2 * slots:
        return 2, f"shard of {B} >= two rounds of the {slots} resident workgroups ({cus} CUs x {per_cu}): transfers of piece c +- 1 under the solve of piece c"
    return 1, f"shard of {B} < two rounds of the {slots} resident workgroups ({cus} CUs x {per_cu}): a smaller piece costs more than its transfer hides"


def transfer_bytes_per_problem(N, M):
    """What the strong-scaling step ships per problem: xinit (9) + x0 (17 N) + parameters ((10 + 4 M) N) doubles + face counts (N ints) to the
    rank, the plan (17 N doubles) + exit flag + iteration count back (bench.py: full_in / full_out)."""
    return (9 + 17 * N + (10 + 4 * M) * N) * 8 + 4 * N, 17 * N * 8 + 8


def launch_plan(args):
    """What `bench.py --gpus N ...` would do, without a device: shards per rank, pieces per shard, the policy behind them."""
    from forces_resilient_planner_amd import distributed as D
    world = max(1, args.gpus)
    cfg = args.config
    base_B = {2: 4096, 3: 16384, 4: 65536}[cfg]
    N, MF = {2: (20, 6), 3: (30, 15), 4: (20, 6)}[cfg]
    strong = args.scaling == "strong"
    if strong or cfg == 4:
        B_total = args.batch or base_B
        if cfg == 4 and not strong:
            B_total = (args.batch or base_B // world) * world
        shards = [D.shard_range(B_total, r, world) for r in range(world)]
    else:
        B = args.batch or (base_B if cfg == 2 else base_B // world)
        B_total = B * world
        shards = [(r * B, (r + 1) * B) for r in range(world)]
    cus, cus_src = device_cus()
    M = {2: 30, 3: 15, 4: 30}[cfg]
    b_in, b_out = transfer_bytes_per_problem(N, M)
    plan = {"world_size": world, "config": cfg, "scaling": args.scaling, "batch_total": B_total, "cus_per_device": cus, "cus_source": cus_src, "ranks": []}
    for r, (lo, hi) in enumerate(shards):
        ch, why = (auto_chunks(world, hi - lo, N, MF, cus) if args.chunks <= 0 else (args.chunks, "--chunks given")) if (strong and cfg != 4) else (1, "weak scaling: no transfer in the step")
        cb = D.chunk_bounds(hi - lo, ch)  # (the same boundaries strong_step_overlapped walks)
        pieces = [[lo + int(a), lo + int(b)] for a, b in zip(cb[:-1], cb[1:])]
        rec = {"rank": r, "problems": [lo, hi], "pieces": ch, "piece_ranges": pieces, "why": why}
        if strong and cfg != 4 and r != 0:
            # what the first real 8-GPU line can be compared with: every peer's pieces travel over ITS OWN xGMI link to / from rank 0 (the
            # scatter of a piece is one grouped P2P exchange: the root's seven links carry seven shards in parallel), so the time of a
            # piece is its bytes over one link's rate, not over the sum of the links
            rec["bytes_to_rank_per_piece"] = [int((b - a) * b_in) for a, b in pieces]
            rec["bytes_from_rank_per_piece"] = [int((b - a) * b_out) for a, b in pieces]
            rec["expected_scatter_ms_per_piece_at_153GBs"] = [round((b - a) * b_in / (XGMI_LINK_GBS * 1e9) * 1e3, 4) for a, b in pieces]
            rec["expected_gather_ms_per_piece_at_153GBs"] = [round((b - a) * b_out / (XGMI_LINK_GBS * 1e9) * 1e3, 4) for a, b in pieces]
        plan["ranks"].append(rec)
    if strong and cfg != 4 and world > 1:
        per = shards[1][1] - shards[1][0]
        plan["transfer_model"] = {"bytes_in_per_problem": b_in, "bytes_out_per_problem": b_out, "link_GBs": XGMI_LINK_GBS,
                                  "shard_scatter_ms": round(per * b_in / (XGMI_LINK_GBS * 1e9) * 1e3, 4), "shard_gather_ms": round(per * b_out / (XGMI_LINK_GBS * 1e9) * 1e3, 4),
                                  "note": "per peer, one link each way; the pieces of a shard travel under the solves of its other pieces (strong_scaling_phases.exposed_transfer_ms of the "
                                          "real line is what was NOT hidden); an exposed time well above shard_scatter_ms / pieces means the links are shared or the group is serialised"}
    plan["collectives"] = ("none on the data path (weak scaling: every rank solves its own problems); summary statistics by all_reduce / all_gather" if not strong
                           else "grouped P2P scatter of the inputs from rank 0 / gather of the plans to rank 0 per piece (RCCL), under the solves")
    return plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU (weak) / in total (strong); default: the BASELINE size of the config")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4],
                    help="BASELINE.json configs[i]: 2 = headline (N=20, 6 faces), 3 = N=30 / <=15 faces / time-varying f_ext, "
                         "4 = Monte-Carlo receding horizon (a step is one tick of every planner)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank solves its own batch (no data-path communication).  strong: rank 0 owns the whole batch in "
                         "HBM; every step scatters the shards (grouped RCCL send/recv), solves, gathers the plans back (SURVEY 8e); "
                         "configs[4]: one nominal problem is broadcast, the samples are drawn on every rank's device")
    ap.add_argument("--repeats", type=int, default=5, help="the K-step region is timed this many times; the MEDIAN is reported")
    ap.add_argument("--no-order-hint", action="store_true", help="configs[4]: queue the problems by the cost of the initial guess instead of by the previous tick's iteration counts")
    ap.add_argument("--mu0", type=float, default=None, help="frp_nmpc_options.mu0, the barrier parameter the interior-point iteration starts from (default: the library's 1.0); "
                    "a caller of warm-started problems (configs[4], the device tick) lowers it; set for the GPU and for the CPU baseline alike, named in config")
    ap.add_argument("--twist", type=int, default=0, help="frp_nmpc_options.twist: stages the model wave eliminates forward while the Riccati wave runs the rest backward "
                    "(0 = the plain solve, -1 = 9 N / 20 up to 1024 problems and 3 N / 10 beyond; N <= 20 only; DESIGN 9.1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / end-to-end / drop-in latency legs")
    ap.add_argument("--chunks", type=int, default=0, help="--scaling strong: pieces every shard is cut into so that transfers overlap the solve (1 = serial scatter -> "
                    "solve -> gather; 0 = auto: 2 when there is more than one rank and a shard holds at least two rounds of resident workgroups, else 1 -- measured on "
                    "one GPU, where nothing is transferred: a piece smaller than a round of resident workgroups costs more than it hides, profiles/r04_strong_chunks.txt)")
    ap.add_argument("--dry-run", action="store_true", help="print the planned shard / piece table of this launch (one JSON line: which rank solves which "
                    "problems, in how many pieces and why) and exit: no device, no process group -- what an N-GPU run WOULD do")
    ap.add_argument("--streams", type=int, default=1,
                    help="informational: `value` is always the strictly serial single-stream rate; the rate with the steps issued "
                         "round-robin on 2 streams (the tail of one launch overlapping the next) is reported in config.pipelined_*")
    args = ap.parse_args()
    if args.dry_run:
        print(json.dumps(launch_plan(args)))
        return

    # `python bench.py --gpus N` by itself must be an N-rank run: without a launcher around it (no WORLD_SIZE) re-execute under
    # torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = self_launch_command(args.gpus, sys.argv[1:])
        os.execv(cmd[0], cmd)

    import torch
    from forces_resilient_planner_amd import distributed as D
    from forces_resilient_planner_amd import layout as L
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
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    strong = args.scaling == "strong"
    if strong and dist is None:
        class _Solo:  # one rank: scatter / gather degenerate to device copies, same code path
            P2POp = None
            @staticmethod
            def get_world_size(): return 1
            @staticmethod
            def get_rank(): return 0
            @staticmethod
            def batch_isend_irecv(ops): return []
            @staticmethod
            def broadcast(t, src): return None
        sdist = _Solo
    else:
        sdist = dist

    cfg = args.config
    base_B = {2: 4096, 3: 16384, 4: 65536}[cfg]
    f64 = dict(dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    phases = None
    fleet = None
    if cfg in (2, 3):
        gen = workloads.config2 if cfg == 2 else workloads.config3
        MF = 6 if cfg == 2 else 15
        if strong:
            B_total = args.batch or base_B
            lo, hi = D.shard_range(B_total, rank, world)
            B = hi - lo
            w = gen(B_total, seed=workloads.SEED0 + 3) if rank == 0 else None  # the same problems as rank 0 of the weak-scaling run
            wcpu = w
            N, M = (w["N"], w["M"]) if rank == 0 else (20 if cfg == 2 else 30, 30 if cfg == 2 else 15)
            model = 0
            ds = solver.DeviceSolver(max(B, 1), N, M, MF, model, f"cuda:{local_rank}")
            if rank == 0:  # the whole batch lives in rank 0's HBM
                full_in = [torch.from_numpy(np.ascontiguousarray(w[k])).to(dev) for k in ("xinit", "x0", "params")] + \
                          [torch.from_numpy(np.ascontiguousarray(w["nfaces"], dtype=np.int32)).to(dev)]
                full_out = [torch.zeros((B_total, N, L.NZ), **f64), torch.zeros((B_total,), dtype=torch.int32, device=dev),
                            torch.zeros((B_total,), dtype=torch.int32, device=dev)]
            else:
                full_in, full_out = [None] * 4, [None] * 3
            shard_in = [ds.xinit[:B], ds.x0[:B], ds.params[:B], ds.nfaces[:B]]
            shard_out = [ds.z[:B], ds.exitflag[:B], ds.iters[:B]]

            # the step streams the transfers UNDER the solve: every shard in pieces, piece c + 1 arriving and the plans of piece
            # c - 1 leaving while piece c is solved (distributed.strong_step_overlapped; --chunks 1 = the serial scatter -> solve -> gather)
            comm_s, comp_s = torch.cuda.Stream(dev), [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            if args.chunks <= 0:
                args.chunks, why = auto_chunks(world, B, N, MF)
                if rank == 0:
                    print(f"[bench] --chunks 0: {args.chunks} piece(s) per shard ({why})", file=sys.stderr, flush=True)

            def step():
                # (pieces alternate between two compute streams: the long solves that end one piece's launch overlap the next piece)
                D.strong_step_overlapped(full_in, shard_in, shard_out, full_out, B_total, sdist,
                                         lambda a_, b_, c_: ds.solve_range(a_, b_, torch.cuda.current_stream(dev), piece=c_ % 2), chunks=args.chunks,
                                         comm_stream=comm_s, compute_stream=comp_s)
            phases = (lambda: D.scatter_batch(full_in, shard_in, B_total, sdist), lambda: ds.solve(stream) if B > 0 else None,
                      lambda: D.gather_batch(shard_out, full_out, B_total, sdist))
        else:
            B = args.batch or (base_B if cfg == 2 else base_B // max(1, args.gpus))
            B_total = B * world
            w = gen(B, seed=workloads.SEED0 + 3 + 1000 * rank)
            wcpu = w
            N, M, model = w["N"], w["M"], w["model"]
            ds = solver.DeviceSolver(B, N, M, MF, model, f"cuda:{local_rank}")
            ds.upload(w)

            def step():
                ds.solve(stream)
        if cfg == 3:
            # configs[3]'s generator makes (locally) infeasible instances; the benchmark opts into the early infeasibility
            # exit (frp_nmpc_options.diverge_mu, default 1e3) -- the CPU baseline below runs with the same setting
            ds.opt.diverge_mu = 10.0
        ds.opt.twist = args.twist
        if args.mu0 is not None:
            ds.opt.mu0 = args.mu0
    else:
        # configs[4]: Monte-Carlo f_ext around ONE nominal problem, warm-started receding horizon; a step = one tick
        from forces_resilient_planner_amd.workloads import _bbox_faces, _weights
        B_total = args.batch or base_B
        if not strong:
            B_total = (args.batch or base_B // max(1, args.gpus)) * world
        lo, hi = D.shard_range(B_total, rank, world)
        B = hi - lo
        ticks = args.warmup + args.steps * (args.repeats + 2) + 4
        w1 = workloads.config4_nominal(1, ticks=ticks)
        N, M, model = w1["N"], w1["M"], w1["model"]
        wcpu = workloads.config4_nominal(min(B_total, 4096), ticks=1)
        # one nominal problem from rank 0 to everybody (a few KB), the samples drawn where they are used
        nominal = [torch.from_numpy(np.ascontiguousarray(w1[k])).to(dev) for k in ("mpc_output", "E")]
        fbar = torch.from_numpy(np.ascontiguousarray(w1["f_ext"].mean(0))).to(dev)
        if sdist is not None:
            D.broadcast_nominal(nominal + [fbar], sdist)
        fleet = solver.DeviceFleet(max(B, 1), N, M, 6, model, _weights(model), f"cuda:{local_rank}")
        ds = fleet.solver
        if args.mu0 is not None:
            ds.opt.mu0 = args.mu0
        ds.order_by_last_iters = not args.no_order_hint  # queue order = the previous tick's iteration counts
        fleet.mpc_output.copy_(nominal[0].expand(max(B, 1), N + 1, L.NZ))
        fleet.ellipsoid.copy_(nominal[1].expand(max(B, 1), N, 3, 3))
        fleet.poly_nfaces.fill_(6)
        f_ext = D.monte_carlo_fext(fbar.cpu().numpy(), 0.5, lo, max(hi, lo + 1), workloads.SEED0 + 5, dev)
        yaw = float(w1["heading"][0])
        ref_yaw = torch.full((max(B, 1), N), yaw, **f64)
        refs, As, bs = [], [], []
        for t in range(ticks):
            r1 = w1["ref_long"][:, t:t + N]
            A1, b1 = _bbox_faces(r1, np.full((1, N), yaw))
            refs.append(torch.from_numpy(r1).to(dev)); As.append(torch.from_numpy(A1).to(dev)); bs.append(torch.from_numpy(b1).to(dev))
        tick = [0]
        # every tick of a receding horizon has its own iteration counts (5.0 on the first ticks, 3.0 forty ticks on): the rate and the
        # kernel time are averages over the timed ticks, so the iteration count they are priced with is too -- summed on the device,
        # one small reduction per tick inside the timed step (~10 us of a ~10 ms tick)
        it_acc = torch.zeros(2, dtype=torch.int64, device=dev)

        def step():
            t = tick[0]; tick[0] += 1
            fleet.poly_A.copy_(As[t].expand(max(B, 1), N, 6, 3)); fleet.poly_b.copy_(bs[t].expand(max(B, 1), N, 6))
            if t > 0:
                fleet.coldstart(thrust=7.3)
            fleet.tick(f_ext, refs[t].expand(max(B, 1), N, 3).contiguous(), ref_yaw)
            it_acc[0] += ds.iters[:max(B, 1)].sum(); it_acc[1] += 1
    torch.cuda.synchronize(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(fn, k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        barrier()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        step()
    # the dominant kernel's duration comes from THE SAME launches as ms_per_step: a hipEvent pair around the solver kernel of
    # every 4th launch of the timed regions, on the launch stream (frp_nmpc_kernel_timing_*), read after the last region
    # (a pair costs its launch ~8 us of queue bubbles: observing every launch would slow the region it measures by 0.8 %)
    nrep = max(1, args.repeats)
    ev_stride = 4 if nrep * args.steps >= 16 and not strong else 1   # (strong scaling: a step is several piece launches, all of them timed)
    solver.kernel_timing_begin((nrep * args.steps + 8) * (max(1, args.chunks) if strong else 1), ev_stride)
    if cfg == 4:
        it_acc.zero_()
    reps = [timed(step, args.steps) for _ in range(nrep)]
    timed_ticks_it = [int(x) for x in it_acc.cpu()] if cfg == 4 else None  # [sum of iterations over the timed ticks, ticks]
    kernel_ms, kernel_launches = solver.kernel_timing_end()
    if strong and kernel_launches:  # per step: the sum over the pieces of this rank's shard
        kernel_ms = kernel_ms * kernel_launches / (nrep * args.steps)
    per_rank_s = [float(np.median(reps))]  # this rank's own median region time (before the max over ranks)
    ranks_seen = 1
    if dist is not None:
        ranks_seen = int(dist.get_world_size())  # what RCCL itself reports after init: the driver can see that N ranks took part
        pr = [torch.zeros(1, **f64) for _ in range(ranks_seen)]
        dist.all_gather(pr, torch.tensor(per_rank_s, **f64))  # (summary statistics, not on the data path)
        per_rank_s = [float(x.item()) for x in pr]
        t = torch.tensor(reps, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # every repeat: the slowest rank
        reps = [float(x) for x in t.cpu()]
    elapsed = float(np.median(reps))

    # informational: the same steps issued round-robin on two streams (own solver state / outputs each)
    pipelined = None
    if cfg in (2, 3) and not strong:
        d2 = solver.DeviceSolver(B, N, M, ds.MF, model, f"cuda:{local_rank}")
        d2.xinit, d2.x0, d2.params, d2.nfaces = ds.xinit, ds.x0, ds.params, ds.nfaces
        d2.opt = ds.opt
        s2 = torch.cuda.Stream(dev)
        lanes = [(ds, stream), (d2, s2)]
        cnt = [0]

        def step2():
            l = lanes[cnt[0] % 2]; cnt[0] += 1
            l[0].solve(l[1])
        timed(step2, 2)
        t2 = float(np.median([timed(step2, args.steps) for _ in range(3)]))
        if dist is not None:
            tt = torch.tensor([t2], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); t2 = float(tt.item())
        pipelined = B_total * args.steps / t2

    # per-phase times of a strong-scaling step (not inside the timed region: each phase is bracketed by a barrier here)
    phase_ms = None
    if phases is not None:
        acc = [0.0, 0.0, 0.0]
        for _ in range(5):
            for i, ph in enumerate(phases):
                acc[i] += timed(ph, 1)
        if dist is not None:
            tt = torch.tensor(acc, dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); acc = [float(x) for x in tt.cpu()]
        phase_ms = {"scatter_ms": acc[0] / 5 * 1e3, "solve_ms": acc[1] / 5 * 1e3, "gather_ms": acc[2] / 5 * 1e3, "chunks": args.chunks,
                    # what the overlapped step still pays for the transfers: its time minus the solve of the whole shard in one launch
                    "exposed_transfer_ms": elapsed / args.steps * 1e3 - acc[1] / 5 * 1e3,
                    # the prediction `--dry-run` prints for this launch: one peer's shard over ONE xGMI link each way (world 1: nothing crosses a link)
                    "predicted_shard_scatter_ms_at_153GBs": (0.0 if world <= 1 else (B_total + world - 1) // world * transfer_bytes_per_problem(int(N), int(M))[0] / (XGMI_LINK_GBS * 1e9) * 1e3),
                    "predicted_shard_gather_ms_at_153GBs": (0.0 if world <= 1 else (B_total + world - 1) // world * transfer_bytes_per_problem(int(N), int(M))[1] / (XGMI_LINK_GBS * 1e9) * 1e3),
                    "note": "scatter / solve / gather timed one after the other, each between barriers (what the serial step would cost); the timed "
                            "step itself cuts every shard into `chunks` pieces and streams the transfers under the solves"}

    fl = ds.exitflag[:max(B, 1)].cpu().numpy(); it = ds.iters[:max(B, 1)].cpu().numpy()
    if B == 0:
        fl = fl[:0]; it = it[:0]
    redo = ds.info[:max(B, 1), 7].cpu().numpy() if B > 0 else np.zeros(0)  # iterations redone with the Gauss-Newton Hessian (uncounted in `iters`)
    stats = torch.tensor([float((fl == 1).sum()), float(it.sum()), float(len(fl)), float(it.max() if len(it) else 0), float(redo.sum())], **f64)
    if cfg == 4 and B > 0 and timed_ticks_it[1] > 0:
        stats[1] = timed_ticks_it[0] / timed_ticks_it[1]  # mean over the timed ticks (the other statistics: the last tick)
    if dist is not None:
        mx = stats[3:4].clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)  # summary statistics, not on the data path
        stats[3] = mx[0]
    conv_frac = float(stats[0] / stats[2]); mean_it = float(stats[1] / stats[2])
    ranks_solving = torch.tensor([1.0 if B > 0 else 0.0], **f64)  # n_gpus = the ranks that actually solved a shard
    if dist is not None:
        dist.all_reduce(ranks_solving, op=dist.ReduceOp.SUM)
    ranks_solving = int(ranks_solving.item())

    torch.cuda.synchronize(dev)

    if rank == 0:
        total = B_total * args.steps
        value = total / elapsed
        MFc = {2: 6, 3: 15, 4: 6}[cfg]
        f_stage = F_STAGE + 18.0 * (MFc - 6)  # the 18 m term of F_stage
        f_solve = mean_it * N * f_stage
        achieved_tf = B * f_solve / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        alg_bytes = 8.0 * (9 + 17 * N + N * (10 + 4 * M) + 1) + 8.0 * 17 * N + 136.0  # dense ABI image of one solve: params + output + info (26 456 B at N = 20, M = 30)
        achieved_gbs = B * alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        kname = "nmpc_ipm_lds_kernel"
        traffic, traffic_src = pmc_traffic(B, kname) if cfg == 2 else (None, None)
        names = {2: "BASELINE.json configs[2]: N=20, constant f_ext~U[-3,3]^3, 6-face tightened corridor per stage, cold start, reference 30-row parameter layout",
                 3: "BASELINE.json configs[3]: N=30, time-varying f_ext, per-stage polytopes with <=15 faces, cold start",
                 4: "BASELINE.json configs[4]: Monte-Carlo f_ext~N(fbar,0.5^2 I) around one nominal problem, N=20, warm-started receding horizon, one step = one tick (pack + solve + update on the device)"}
        out = {
            "metric": "NMPC solves/sec, batch=4096 horizons N=20", "value": value, "unit": "solves/s",
            "n_gpus": int(ranks_solving), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": names[cfg] + f"; batch {B} per GPU" + (f" of {B_total} in total, scattered from / gathered to rank 0 every step" if strong and cfg != 4 else ""),
                       "baseline_config": cfg, "world_size": world, "ranks_seen": ranks_seen,
                       "per_rank_ms_per_step": [x / args.steps * 1e3 for x in per_rank_s],
                       "per_rank_solves_per_s": ([B_total / max(1, ranks_seen) * args.steps / x for x in per_rank_s] if not strong else None), "collectives": ("RCCL (torch.distributed nccl backend)" if dist is not None else "none (single process)"),
                       "batch_per_gpu": B, "batch_total": B_total, "horizon": int(N), "converged_frac": conv_frac,
                       "mean_ipm_iterations": mean_it, **({"mean_ipm_iterations_note": "mean over the timed ticks (what the rate and the kernel time average over); p95 / max / redos: the last tick"} if cfg == 4 else {}), "mean_gauss_newton_redos": float(stats[4] / stats[2]), "max_ipm_iterations": int(stats[3]), "p95_ipm_iterations": float(np.percentile(it, 95)) if len(it) else 0.0,
                       "timing": f"median of {len(reps)} repeats of the {args.steps}-step region, strictly serial launches on one stream; max over ranks per repeat",
                       "repeat_ms_per_step": [r / args.steps * 1e3 for r in reps],
                       "pipelined_solves_per_s": pipelined,
                       "pipelined_note": "the same steps issued round-robin on 2 HIP streams (the few long solves at the end of a launch overlap the head of the next); informational, never `value`",
                       "tolerances": 1e-4, "twist": int(args.twist), "mu0": float(ds.opt.mu0)},
            "roofline": {"bound": "fp64-issue",
                         "bound_detail": "FP64 VALU + FP64 MFMA issue slots (they share the SIMD's FP64 datapath) of in-order wavefronts on the serial stage chain (DESIGN 5); priced against the FP64 matrix == vector peak",
                         "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms": kernel_ms, "kernel_launches_timed": kernel_launches,
                         "kernel_ms_source": f"hipEvent pairs around the solver kernel of every {ev_stride}{'th' if ev_stride > 1 else 'st'} launch of the timed regions themselves (all repeats), on the launch stream",
                         "flops_per_launch": B * f_solve,
                         "note": "FP64 (vector == matrix peak 78.6 TFLOP/s); flops = SURVEY 8d definition "
                                 "mean_it * N * F_stage per solve (F_stage = 31.5 kflop at 6 faces)",
                         "hbm_algorithmic": {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": achieved_gbs / HBM_PEAK_GBS, "bytes_per_solve": alg_bytes}},
        }
        if phase_ms is not None:
            out["config"]["strong_scaling_phases"] = phase_ms
        if not args.no_cpu and world == 1 and not strong and cfg == 2:
            # secondary timings of SURVEY 8d (never `value`): host buffers in and out, and the single-problem drop-in call
            outs = solver.solve_batch_host(wcpu)
            e2e = []
            for _ in range(7):
                t1 = time.perf_counter(); solver.solve_batch_host(wcpu, out=outs); e2e.append(time.perf_counter() - t1)
            pcs = wcpu["params"]
            wcomp = dict(wcpu); wcomp["M"] = 6
            wcomp["params"] = np.ascontiguousarray(np.concatenate([pcs[:, :, :10], pcs[:, :, 10:28], pcs[:, :, 100:106]], axis=2))
            outc = solver.solve_batch_host(wcomp)
            e2c = []
            for _ in range(7):
                t1 = time.perf_counter(); solver.solve_batch_host(wcomp, out=outc); e2c.append(time.perf_counter() - t1)
            # the same call with the caller's arrays registered once (frp_nmpc_host_register): nothing is staged, a gather kernel reads the
            # live part of the inputs from the caller's memory, the solver writes the plans in place.  (Guarded like the other secondary
            # legs: a hipHostRegister refusal -- a container's memlock limit at ~30 MB of pinned memory -- must not cost the bench line.)
            as_c = lambda d: {k: (np.ascontiguousarray(v, dtype=(np.int32 if k in ("nfaces", "models") else np.float64)) if isinstance(v, np.ndarray) else v) for k, v in d.items()}

            def registered_leg(wl, pipelined):
                wr = as_c(wl)
                outr = tuple(np.zeros_like(a) for a in outs)
                reg = [wr[k] for k in ("xinit", "x0", "params", "nfaces") if isinstance(wr.get(k), np.ndarray)] + list(outr)
                res = {}
                solver.host_register(*reg)
                try:
                    solver.solve_batch_host(wr, out=outr)
                    tt = []
                    for _ in range(7):
                        t1 = time.perf_counter(); solver.solve_batch_host(wr, out=outr); tt.append(time.perf_counter() - t1)
                    res["s"] = float(np.median(tt)); res["out"] = outr
                    if pipelined:
                        # two batches in flight (frp_nmpc_solve_batch_host_begin / _wait): a second set of registered buffers with the same problems;
                        # while one batch solves the other's inputs are gathered over the host link.  Steady state: time per batch over 16 batches.
                        wr2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in wr.items()}
                        out2 = tuple(np.zeros_like(a) for a in outs)
                        reg2 = [wr2[k] for k in ("xinit", "x0", "params", "nfaces") if isinstance(wr2.get(k), np.ndarray)] + list(out2)
                        solver.host_register(*reg2)
                        try:
                            sets = [(wr, outr), (wr2, out2)]
                            tk = [solver.solve_batch_host_begin(*sets[0]), solver.solve_batch_host_begin(*sets[1])]
                            for q in (0, 1):
                                solver.solve_batch_host_wait(tk[q]); tk[q] = solver.solve_batch_host_begin(*sets[q])
                            nb = 16
                            t1 = time.perf_counter()
                            for i in range(nb):
                                q = i & 1
                                solver.solve_batch_host_wait(tk[q]); tk[q] = solver.solve_batch_host_begin(*sets[q])
                            for q in (0, 1):
                                solver.solve_batch_host_wait(tk[q])
                            res["pipelined_s"] = (time.perf_counter() - t1) / (nb + 2)
                            res["pipelined_same"] = bool(np.array_equal(out2[0], outs[0]) and np.array_equal(outr[0], outs[0]) and np.array_equal(out2[1], outs[1]))
                        finally:
                            solver.host_unregister(*reg2)
                finally:
                    solver.host_unregister(*reg)
                return res

            e2 = {"solves_per_s": B / float(np.median(e2e)), "ms_per_batch": float(np.median(e2e)) * 1e3,
                  "buffers": "solves_per_s / ms_per_batch: the caller's PAGEABLE numpy arrays (persistent device buffers, pinned staging filled by a few copy threads, chunks "
                             "of B/16, B/4 and the rest whose copies and solves overlap) -- the key rounds 1-4 reported; registered_*: the same arrays registered once "
                             "(frp_nmpc_host_register): pinned in place, read by a gather kernel and written by the solver over PCIe, no staging (round 5 reported THIS "
                             "under solves_per_s); pipelined_*: registered buffers, two batches in flight (frp_nmpc_solve_batch_host_begin / _wait)",
                  "compact_layout_solves_per_s": B / float(np.median(e2c)), "compact_layout_ms_per_batch": float(np.median(e2c)) * 1e3,
                  "same_plans": bool(np.array_equal(outs[0], outc[0]))}
            try:
                r = registered_leg(wcpu, True)
                e2.update({"registered_solves_per_s": B / r["s"], "registered_ms_per_batch": r["s"] * 1e3,
                           "registered_equals_pageable": bool(np.array_equal(outs[0], r["out"][0]) and np.array_equal(outs[1], r["out"][1])),
                           "pipelined_solves_per_s": B / r["pipelined_s"], "pipelined_ms_per_batch": r["pipelined_s"] * 1e3, "pipelined_same_plans": r["pipelined_same"]})
            except Exception as ex:  # noqa: BLE001
                e2["registered_error"] = repr(ex)[:300]
            try:  # ... and the 6-row layout in registered buffers: every input contiguous, the gather at the link's rate (tools/ubench/zc_read)
                r = registered_leg(wcomp, True)
                e2.update({"compact_layout_registered_solves_per_s": B / r["s"], "compact_layout_registered_ms_per_batch": r["s"] * 1e3,
                           "compact_layout_pipelined_solves_per_s": B / r["pipelined_s"], "compact_layout_pipelined_ms_per_batch": r["pipelined_s"] * 1e3,
                           "same_plans": bool(e2["same_plans"] and np.array_equal(outs[0], r["out"][0]) and r["pipelined_same"])})
            except Exception as ex:  # noqa: BLE001
                e2["compact_layout_registered_error"] = repr(ex)[:300]
            e2["what"] = ("frp_nmpc_solve_batch_host, host buffers in and out (PCIe-inclusive, median of 7; pipelined: 18 batches back to back).  dense = the reference's "
                          "30-row parameter layout in the caller's buffers (face counts given: the staging copy / the gather packs the 6 live rows, "
                          "8.9 of 26.4 KB per problem cross PCIe), compact = the same problems handed over with M = 6 rows")
            out["end_to_end"] = e2
            import ctypes
            w0 = workloads.config0()
            p = solver.ForcesParams(); o = solver.ForcesOutput(); info = solver.ForcesInfo()
            p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel(); p.num_of_threads = 1
            lat = []
            for i in range(25):
                t1 = time.perf_counter()
                flag = solver.lib().FORCESNLPsolver_normal_solve(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), None, None)
                lat.append(time.perf_counter() - t1)
            out["dropin_latency_ms"] = {"value": float(np.median(lat[5:])) * 1e3, "exitflag": int(flag), "iterations": int(info.it),
                                        "what": "BASELINE configs[0] through FORCESNLPsolver_normal_solve (params staged in pinned mapped memory, read and written in place by "
                                                "the one-problem solve; one launch, the completion word the kernel stores behind its outputs spun on), warm, median of 20"}
            try:  # the same call with the latency option; the environment variable is read once per process: a child measures it
                import subprocess, re
                e = dict(os.environ); e["FRP_NMPC_TWIST"] = "-1"
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "twist_latency.py"), "dropin"], env=e, capture_output=True, text=True, timeout=120)
                mm = re.search(r"median ([0-9.]+) us", r.stdout)
                if mm:
                    out["dropin_latency_ms"]["with_FRP_NMPC_TWIST=-1"] = float(mm.group(1)) * 1e-3
            except Exception as ex:  # noqa: BLE001 -- informational leg
                out["dropin_latency_ms"]["with_FRP_NMPC_TWIST=-1"] = None
            # the whole planner tick on the device (SURVEY 8f rows f-1..f-4a around the solve): ms per stage of the tick, never `value`
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("full_tick_bench", os.path.join(ROOT, "tools", "full_tick_bench.py"))
                ftb = importlib.util.module_from_spec(spec); spec.loader.exec_module(ftb)
                ft = ftb.run(B=4096, TICKS=10, P=20000, GRID=0.5, SPLIT=0)
                out["full_tick"] = {"ms_per_tick": ft["ms_per_tick"], "planner_ticks_per_s": ft["planner_ticks_per_s"], "ms_per_step": ft["ms_per_step"],
                                    "converged_frac": ft["converged_frac"], "polytopes_per_planner": ft["polytopes_per_planner"],
                                    "what": "DeviceFleet.full_tick for " + ft["workload"] + ": stage references -> tube -> corridor (cloud grid) -> pack -> solve "
                                            "(corridors of up to 30 rows: the (20, 10) kernel variant) -> update, HIP events per step on the launch stream, mean of 10 ticks; "
                                            "per-kernel rooflines: profiles/r05_tick_rooflines.json (tools/tick_rooflines.py)"}
                # the same ticks with the barrier parameter a warm-started solve begins with as an OPTION of the caller (frp_nmpc_options.mu0 =
                # 0.2 instead of 1: tools/full_tick_bench.py, profiles/r05_tick_mu0.txt) -- informational, beside the default above
                fw = ftb.run(B=4096, TICKS=10, P=20000, GRID=0.5, SPLIT=0, mu0=0.2)
                out["full_tick"]["with_options_mu0_0.2"] = {"ms_per_tick": fw["ms_per_tick"], "ms_per_step": fw["ms_per_step"], "converged_frac": fw["converged_frac"],
                                                            "mean_iters": fw["mean_iters"], "mean_iters_default": ft["mean_iters"],
                                                            "plans_vs_default_mu0": fw["plans_vs_default_mu0"]}
            except Exception as e:  # secondary evidence, never a reason to lose the bench line
                out["full_tick"] = {"error": repr(e)}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(wcpu, diverge_mu=float(ds.opt.diverge_mu), mu0=float(ds.opt.mu0))
        elif not args.no_cpu:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:  # RCCL prints its version banner through C stdio: flush it first so that the JSON line is the LAST line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
