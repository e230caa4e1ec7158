#!/usr/bin/env python3
"""bench.py — cost+gradient evaluations per second of the batched GTOP callback.

One "step" = one pass of the hot path (GradTrajOptimizer::getCostAndGradient,
src/grad_traj_optimizer.cpp:281-448 of the reference) over one resident batch
of synthetic trajectories.  Default workload = BASELINE.json configs[1]:
1 024 trajectories per GPU, 20 control points (m = 6 segments, 45 free
variables), shared 200^3 distance field, fp64.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N
ranks itself (fresh child processes, before this process touches a GPU) and
relays rank 0's line.

Multi-GPU: the batch shards across ranks (weak scaling: per-GPU batch fixed),
the distance field is replicated, and the only collective is a bucketed RCCL
all-gather of the per-trajectory costs (one collective per bucket of steps,
overlapped with the next bucket's kernels); --gather-grads adds the optional
all-gather of each bucket's last gradient.

Rank 0 prints ONE JSON line (contract in the task statement), extended with
`roofline` (dominant kernel: gtop_eval_wave_kernel, HBM-bound, algorithmic bytes
of SURVEY.md §8d) and `cpu_baseline` (the oracle's C restatement timed on this
box's host cores; N=1 only).

Environment knobs for rehearsals on a one-GPU box (never set by the driver):
  GTOP_BENCH_FORCE_DIST=1    initialise the process group (and run the
                             collective) even at world size 1
  GTOP_BENCH_BACKEND=gloo    collectives over gloo (staged through the host)
  GTOP_BENCH_SHARE_DEVICE=1  every rank uses device 0 (RCCL refuses two ranks
                             on one device, so only together with gloo)
The JSON line then carries "rehearsal": true.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC (RCCL's transports, and the buffers the ranks map from each other for the push gather) — the pool's driver
# supports no other kind; set before anything initialises the GPU runtime
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")
TOL = {"f64": 1e-5,     # BASELINE.json north_star
       "f32": 2e-4}     # the bound tests/test_gpu_api.py holds the fp32 path to


def measured_traffic(key):
    """HBM-side bytes per launch of the evaluation kernel for this workload, from the
    rocprofv3 --pmc passes committed under profiles/ (tools/pmc_collect.sh +
    tools/pmc_summary.py; counters cannot be read from inside this process).
    {} when this workload has not been profiled."""
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f).get(key) or {}
    except (OSError, ValueError):
        return {}


# Instruction-issue model: a SIMD issues one wave64 instruction of this path per CYCLES_PER_INST cycles
# whatever its kind and however many wavefronts it holds (tools/ubench/pk_rate: 4.4-5.4 measured for fp64, scalar fp32
# and packed fp32 at 1, 2 and 4 wavefronts per SIMD; 4 is the hardware's figure for one wavefront's stream,
# MI355X_MICROARCH.md "vector-instruction ISSUE cost"), at the 2.4 GHz maximum clock.
CYCLES_PER_INST = 4.5
CLOCK_GHZ = 2.4
N_SIMD = 1024           # 256 CUs x 4


def issue_roofline(wkey, launch_us):
    """`roofline_issue`: time the SIMDs need merely to ISSUE the kernel's instructions — issued instructions per
    wavefront (SQ_INSTS_* of the committed PMC summary) x wavefronts per SIMD x cycles per instruction / clock —
    over the measured launch time.  None when the workload has no PMC summary under profiles/."""
    t = measured_traffic(wkey)
    ipw, waves = t.get("issued_per_wave"), t.get("waves_per_launch")
    if not ipw or not waves:
        return None
    waves_per_simd = waves / N_SIMD
    t_issue_us = ipw["total"] * max(1.0, waves_per_simd) * CYCLES_PER_INST / (CLOCK_GHZ * 1e3)
    return {"bound": "instruction issue", "issued_per_wavefront": ipw, "wavefronts_per_launch": waves,
            "wavefronts_per_simd": waves_per_simd, "cycles_per_instruction": CYCLES_PER_INST, "clock_ghz": CLOCK_GHZ,
            "issue_time_us": t_issue_us, "launch_us": launch_us, "frac": t_issue_us / launch_us,
            "source": t.get("issue_source")}


def algorithmic_bytes(m, elem):
    """per evaluation, e*[(9(m-1)+18+m) + (1+9(m-1))] + e*8*30*m."""
    n = 9 * (m - 1)
    return elem * ((n + 18 + m) + (1 + n)) + elem * 8 * 30 * m


def dominant_kernel(m, B, dtype, pinned):
    """Name of the kernel that serves this workload (the launch rule, csrc/gtop_kernels.hip gtop_eval_plan /
    pick_geometry) — what the rocprofv3 CSVs under profiles/ list it as.  Template arguments: arithmetic type, 64-bit
    field indices, samples per lane, trajectories per wavefront, collision term, register budget (wavefronts per SIMD),
    optimizer state, DYN, more than 12 segments."""
    R = "double" if dtype == "f64" else "float"
    if pinned:
        return "gtop_eval_wave_kernel (pinned samples per lane)"
    tail = "(anonymous namespace)::GtopNoMma, false, "
    if 12 < m <= 64 and ((64 // m) * ((m + 11) // 12) >= 5 or (m + 11) // 12 >= 4) and B >= 1024 * (64 // m):
        return f"gtop_eval_wave_kernel<{R}, false, 30, 1, true, 3, {tail}false>"    # one lane per segment past 12 segments
    if dtype == "f32" and m <= 12 and B * m >= 65536 * 6:
        return f"gtop_eval_wave_kernel<{R}, false, 30, 1, true, 3, {tail}false>"    # one lane per segment, 64 / m trajectories per wavefront
    if m <= 10 and m != 6 and B >= (8192 if m <= 5 else 4096):
        return f"gtop_eval_wave_kernel<{R}, false, 10, 1, true, 3, {tail}false>"    # three lanes per segment, 21 / m trajectories per wavefront
    if m <= 6:
        if B >= (2048 if dtype == "f32" else 4096):
            return f"gtop_eval_wave_kernel<{R}, false, 6, 2, true, 3, {tail}false>"   # two trajectories per wavefront (fp32: packed pairs)
        return f"gtop_eval_wave_kernel<{R}, false, 3, 1, true, {3 if B >= 3072 else 2}, {tail}false>"
    if m <= 12 and B <= (1024 if dtype == "f64" else 512):
        return f"gtop_eval_wave_kernel<{R}, false, 3, 1, true, 2, {tail}false, 2>"      # two wavefronts per trajectory
    return f"gtop_eval_wave_kernel<{R}, false, 6, 1, true, 3, {tail}{'true' if m > 12 else 'false'}>"


def host_threads():
    """Threads the CPU legs may use: the affinity mask, capped by the cgroup CPU
    quota and by 16 (a 1-GPU box's CPU share) — os.cpu_count() reports the
    whole host and would oversubscribe the quota with spinning OpenMP threads."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories PER GPU")
    ap.add_argument("--segments", type=int, default=6, help="m (6 -> '20 control points')")
    ap.add_argument("--grid", type=int, default=200, help="distance field is grid^3")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--density", type=float, default=0.02)
    ap.add_argument("--bucket", type=int, default=50, help="steps per graph / per cost all-gather")
    ap.add_argument("--gather-grads", action="store_true",
                    help="also all-gather the gradient of each bucket's last step (a global optimizer step's input)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra workloads reported under 'extras'")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-sort", action="store_true",
                    help="keep the generator's random batch order (default: Morton-ordered for L2 locality)")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts per workgroup (0 = auto, 1: there is one)")
    ap.add_argument("--spl", type=int, default=0, help="samples per lane (0 = auto, 3 or 6)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as fresh
    child processes (this parent has not imported torch, let alone touched a
    GPU) and relay their exit code; rank 0's JSON line goes straight to the
    inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def parity_check(ctx, oracle, osdf, x, Df, T, host_rows, dtype, rows=256):
    """HIP path vs the oracle on the first `rows` trajectories.  The WHOLE batch
    is launched (so the kernel variant checked is the one that gets timed);
    its first rows are compared.  host_rows = (T, Df, x) numpy, same order."""
    import numpy as np
    import torch
    n = min(rows, x.shape[0])
    c_dev, g_dev = ctx.eval_device(x, Df, T)
    torch.cuda.synchronize()
    Th, Dfh, xh = host_rows
    c_ref, g_ref, _ = oracle.eval_batch(Th[:n], Dfh[:n], xh[:n], osdf, oracle.make_params(), nthreads=host_threads())
    c = c_dev[:n].double().cpu().numpy()
    g = g_dev[:n].double().cpu().numpy()
    rc = float(np.max(np.abs(c - c_ref) / np.abs(c_ref)))
    rg = float(np.max(np.max(np.abs(g - g_ref), axis=1) / np.max(np.abs(g_ref), axis=1)))
    tol = TOL[dtype]
    return {"n": n, "max_rel_cost": rc, "max_rel_grad": rg, "tol": tol, "ok": bool(rc <= tol and rg <= tol)}


def oracle_field(oracle, mp, ctx):
    osdf = oracle.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    osdf.dist[:] = ctx.get_sdf().reshape(-1)
    return osdf


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np  # noqa: F401  (used by helpers)
    import torch
    import torch.distributed as dist
    import grad_traj_optimization_amd as gtop
    from grad_traj_optimization_amd import problem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GTOP_BENCH_BACKEND", "nccl")     # nccl == RCCL on ROCm
    share_dev = os.environ.get("GTOP_BENCH_SHARE_DEVICE") == "1"
    force_dist = os.environ.get("GTOP_BENCH_FORCE_DIST") == "1"
    rehearsal = share_dev or backend != "nccl" or (force_dist and world == 1)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    dev_index = 0 if share_dev else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    collective = world > 1 or force_dist
    if collective:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        kw = dict(device_id=dev) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)

    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    elem = 8 if args.dtype == "f64" else 4
    m, Bl = args.segments, args.batch
    B_total = Bl * world

    # ---- synthetic inputs (same seeds on every rank; each rank keeps its shard) ----
    mp = problem.make_map(args.grid, density=args.density, seed=0)
    ctx = gtop.GtopContext(device=dev_index)
    if args.waves or args.spl:
        ctx.set_launch_geometry(args.waves, args.spl)
    t0 = time.time()
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())       # ESDF built on the GPU, stays resident
    esdf_s = time.time() - t0
    if rank == 0:
        log(f"map {args.grid}^3 built on GPU in {esdf_s:.3f} s")
    batch = problem.make_trajectories(B_total, m, mp, seed=1)
    if not args.no_sort:   # one-time setup, like setPath: trajectories that run together read the same map region
        batch = problem.permute(batch, problem.spatial_order(batch.waypoints, mp.origin, mp.map_size))
    lo, hi = problem.shard_range(B_total, rank, world)
    x = torch.tensor(batch.x[lo:hi], dtype=tdtype, device=dev)
    Df = torch.tensor(batch.Df[lo:hi].reshape(-1, 18), dtype=tdtype, device=dev)
    T = torch.tensor(batch.T[lo:hi], dtype=tdtype, device=dev)
    n = x.shape[1]

    # Steps per graph.  Single GPU, no collective (the driver's N = 1 run): ALL steps of the timed region in one
    # hipGraph — no gaps between graphs inside the region (round 3: forty 50-kernel graphs put 3-5 % of launch gaps
    # into a 2 000-step region).  With a collective: buckets of --bucket steps, one all-gather each.  Where the gathers
    # are host-side asynchronous calls (gloo, or GTOP_BENCH_CAPTURE_GATHER=0) a gather runs beside the next bucket's
    # kernels, so a short run is split in (at least) two buckets to have something to hide behind; where they are
    # captured into the bucket graphs (RCCL, the default) they run in line behind their bucket's kernels — exposed by
    # construction, see the measurements at the capture below — and a short run is ONE bucket: one gather, not two.
    MAX_GRAPH_STEPS = 4096
    # GTOP_BENCH_GATHER: "push" (default) = the all-gather as point-to-point stores (gtop_push_rows: one kernel behind the
    # bucket's last evaluation writes this rank's rows into every rank's buffer, the peers' buffers mapped over CUDA-IPC),
    # falling back to "library" = the process group's all_gather_into_tensor (RCCL) when the buffers cannot be mapped
    want_push = collective and os.environ.get("GTOP_BENCH_GATHER", "push") == "push"
    gathers_in_line = collective and os.environ.get("GTOP_BENCH_CAPTURE_GATHER", "1") == "1" and (want_push or backend == "nccl")
    if not collective:
        G = max(1, min(args.steps, MAX_GRAPH_STEPS))
    elif gathers_in_line:
        G = max(1, min(args.bucket, args.steps))
    else:
        G = max(1, min(args.bucket, args.steps // 2 if args.steps >= 2 else 1))
    while args.steps % G:
        G -= 1
    nbuckets = args.steps // G

    # ---- parity gate (rank 0): HIP vs oracle on a subsample of this rank's shard ----
    parity = None
    osdf = None
    if rank == 0:
        from oracle import oracle
        osdf = oracle_field(oracle, mp, ctx)
        parity = parity_check(ctx, oracle, osdf, x, Df, T, (batch.T[lo:hi], batch.Df[lo:hi], batch.x[lo:hi]),
                              args.dtype)
        log(f"parity {parity}")
        if not parity["ok"]:
            print(f"bench.py: PARITY FAILED {parity}", file=sys.stderr)
            sys.exit(3)

    # ---- launch plan: one hipGraph per result ring buffer, G steps each, a device-clock stamp at either end ----
    from grad_traj_optimization_amd.distributed import ResultGatherPipeline
    stream = torch.cuda.current_stream(dev)
    graphs = None
    STAMP_INIT = torch.tensor([2 ** 63 - 1, 0], dtype=torch.int64, device=dev)
    stamps = STAMP_INIT.clone()          # [earliest, latest] device-clock stamp since the last reset
    clock_hz = ctx.clock_hz()

    def run_bucket_eager(j, steps=None, stamp=True, tail_stamp=True):
        if stamp:
            ctx.clock_stamp(stamps)
        for s in range(G if steps is None else steps):
            ctx.eval_device(x, Df, T, pipe.cost_ring[j][s], pipe.grad_ring[j])
        if stamp and tail_stamp:
            ctx.clock_stamp(stamps)     # behind the last kernel, in front of the bucket's all-gather

    def run_bucket_graph(j):
        graphs[j % len(graphs)].replay()   # (a single-bucket run has one graph: every replay writes ring 0)

    pipe = ResultGatherPipeline(world, rank, G, hi - lo, tdtype, dev, run_bucket_eager,
                                n_free=n, gather_grads=args.gather_grads, collective=collective)
    launch_mode = "eager"
    gather_mode = "none" if not collective else "host call per bucket, overlapped with the next bucket"
    push_mode, push_note = False, None
    if want_push:
        push_mode, push_note = pipe.enable_push(ctx)
        if rank == 0:
            log(f"gather by point-to-point stores: {'on' if push_mode else 'OFF, library all-gather instead — ' + push_note}")
    nrings = 2 if collective else min(2, nbuckets)
    kernel_graphs = None                 # the same buckets without their all-gather (collective runs: what the gather costs)
    if not args.no_graph:
        def capture(with_gather):
            gs = []
            for j in range(nrings):
                gph = torch.cuda.CUDAGraph()
                # thread_local: a HIP call from another thread (RCCL's watchdog) must not break the capture
                with torch.cuda.graph(gph, capture_error_mode="thread_local"):
                    if with_gather and push_mode:
                        # the gather kernel takes the bucket's closing clock stamp itself as it starts: a node saved
                        run_bucket_eager(j, tail_stamp=False)
                        pipe.gather_now(j, clock_minmax=stamps)
                    else:
                        run_bucket_eager(j)
                        if with_gather:
                            pipe.gather_now(j)
                gs.append(gph)
            return gs
        # with a collective (RCCL only): first try to capture the bucket's all-gather INTO its graph, behind the last
        # kernel — one graph launch per bucket and no collective call from the host (which costs ~50 us exposed per
        # short timed region); if RCCL cannot be captured here, fall back to graphs of kernels + host-side gathers
        attempts = ([True] if (gathers_in_line and (push_mode or backend == "nccl")) else []) + [False]
        for with_gather in attempts:
            try:
                if with_gather:     # the communicator must exist (and have run once) before it is captured
                    for j in range(nrings):
                        pipe.gather_now(j)
                    torch.cuda.synchronize()
                graphs = capture(with_gather)
                pipe.run_bucket_fn = run_bucket_graph
                pipe.gather_in_bucket_fn = with_gather
                launch_mode = "hipgraph"
                if with_gather:
                    gather_mode = ("point-to-point stores (gtop_push_rows), captured in each bucket's hipGraph" if push_mode
                                   else "captured in each bucket's hipGraph")
                    kernel_graphs = capture(False)
                else:
                    kernel_graphs = graphs
                break
            except Exception as e:   # capture unsupported: fall back, say so
                print(f"bench.py: graph capture {'with the all-gather ' if with_gather else ''}failed ({e})", file=sys.stderr)
                graphs = None
                torch.cuda.synchronize()
        if graphs is None:
            print("bench.py: using eager launches", file=sys.stderr)
        # The captured all-gather of a bucket runs BEHIND the bucket's kernels, on the same stream: exposed by
        # construction.  The two ways to run it beside the next bucket's kernels were built and measured at RCCL world
        # size 1 (tools/calls/call_r4_20.sh, 20 steps in 2 buckets / 200 in 4, host us): one graph for the whole region
        # with the gathers as forked branches 176 / 1 200; kernel graphs on the main stream and gather graphs on a side
        # stream behind events 212 / 1 365; host-side async all-gather calls 195 / 1 213 — against 126 / 834 for this
        # form (kernels alone: 122 / 841).  A cross-stream dependency costs this runtime tens of microseconds, more than
        # a small all-gather can take; so the gathers stay in line, and a run has as few of them as its --bucket allows.
    if rank == 0:
        log(f"launch mode {launch_mode}, {G} steps per bucket x {nbuckets}, backend {backend if collective else 'none'}")
    run_bucket, drain = pipe.run_bucket, pipe.drain
    if graphs is not None and pipe.gather_in_bucket_fn:
        def run_bucket(b):                # (nothing pending on the host in this form: straight to the graph)
            graphs[(b & 1) % len(graphs)].replay()

    def barrier():
        if collective:
            dist.barrier()

    # ---- graph upload: the first replay of an instantiated graph pays a one-time cost (about 25 us here).  It is
    #      part of building the launch plan, like the capture itself, and is reported as such; it is not a step.
    #      It goes through the pipeline, so the first all-gather of each ring buffer (RCCL sets its channels up
    #      lazily, hundreds of ms) is paid here too, not inside the timed region of a short run.
    upload_replays = 0
    if graphs is not None or collective:
        for j in range(nrings):
            run_bucket(j)
            upload_replays += 1
        drain()
        if kernel_graphs is not None and kernel_graphs is not graphs:
            for gph in kernel_graphs:
                gph.replay()
        torch.cuda.synchronize()

    # ---- warmup: exactly W steps (whole buckets through the timed path, the rest as single launches) ----
    for b in range(args.warmup // G):
        run_bucket(b)
    drain()
    run_bucket_eager(0, args.warmup % G, stamp=False)
    torch.cuda.synchronize()

    # ---- clock warm-up: CLOCK_WARMUP_MS of the timed region's own bucket replays, declared in `config`.  After the
    #      parity gate's seconds of host work the card needs 10-20 ms of load to reach its sustained clocks
    #      (tools/clock_ramp.py: 4.07 us per launch in the first 8 ms, 3.85 from 24 ms on, flat to 185 ms).  Round 3
    #      warmed up only the probe behind the region, so `value` and `roofline.frac` described two clock states; now
    #      the region, the device-clock stamps around it and the probe all run at the sustained clocks.  With a
    #      collective every rank must replay the same number of buckets, so the count comes from the arguments alone.
    CLOCK_WARMUP_MS = 40.0
    clock_warmup_buckets = 0
    if not collective:
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < CLOCK_WARMUP_MS * 1e-3:
            for _ in range(max(1, 200 // G)):
                run_bucket(clock_warmup_buckets)
                clock_warmup_buckets += 1
            torch.cuda.synchronize()
    else:
        est_bucket_s = G * 5e-6 * max(1.0, Bl * m / (1024.0 * 6))
        for _ in range(max(2, int(math.ceil(CLOCK_WARMUP_MS * 1e-3 / est_bucket_s)))):
            run_bucket(clock_warmup_buckets)
            clock_warmup_buckets += 1
        drain()
        torch.cuda.synchronize()

    # ---- the collectives the timed region's bracket itself uses (the start-time broadcast, the barrier), once, untimed:
    #      RCCL sets an operation's channels up lazily at its first use, and part of that work ran INTO a short region
    #      (the first 20-step region of a process read 170-280 us, every later one 114: tools/calls/call_r4_26.sh)
    if collective and os.environ.get("GTOP_BENCH_WARM_BRACKET", "1") == "1":
        for _ in range(2):
            tw = torch.tensor([0.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.broadcast(tw, 0)
            tw.item()
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly K = nbuckets*G steps ----
    def timed_region(run_b):
        """K steps through run_b(bucket), bracketed by barrier + synchronize on both sides.  Returns this rank's host
        seconds and the device clock's seconds between the region's first and last stamp (first kernel's dispatch to the
        end of the last kernel, collectives of the buckets in between included)."""
        stamps.copy_(STAMP_INIT)
        start_at = None
        if collective:
            # Ranks leave a barrier tens of microseconds apart, and the region's collective charges the last one's
            # lateness to everybody (a fifth of a 0.1 ms region).  So the ranks of this node agree — before the barrier
            # — on an instant shortly after it (CLOCK_MONOTONIC is one clock for all processes of a node) and start
            # there, together.
            tt0 = torch.tensor([time.perf_counter() + 0.003], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.broadcast(tt0, 0)
            start_at = float(tt0.item())
        barrier()
        torch.cuda.synchronize()
        if start_at is not None and 0.0 < start_at - time.perf_counter() < 0.01:   # (another node's clock, a slow barrier: start now)
            # the wait for the common instant, with the GPU kept at work: three idle milliseconds let the card's clocks sag
            # (the region's kernels then read 4.0-4.1 us where the single path, which has no such gap, reads 3.75)
            if kernel_graphs is not None and os.environ.get("GTOP_BENCH_BUSY_WAIT", "1") == "1":
                while start_at - time.perf_counter() > 400e-6:
                    kernel_graphs[0].replay()
                    torch.cuda.synchronize()
                stamps.copy_(STAMP_INIT)
                torch.cuda.synchronize()
            while time.perf_counter() < start_at:
                pass
        t0 = time.perf_counter()
        for b in range(nbuckets):
            run_b(b)
        drain()                    # the stream now also waits for the last bucket's all-gather (no host wait yet)
        ta = time.perf_counter()
        while not stream.query():  # poll: a blocking wait sleeps on an interrupt, tens of us after the last kernel ends
            pass
        tb = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()   # this rank's K steps and their collectives are done; the job's time is the MAX over ranks
        barrier()
        st = stamps.tolist()
        if rank == 0 and os.environ.get("GTOP_BENCH_REGION_SPLIT"):     # diagnostic: where the host's time around the region goes
            log(f"region split: launch calls {(ta - t0) * 1e6:.1f} us, poll {(tb - ta) * 1e6:.1f}, synchronize {(t1 - tb) * 1e6:.1f}, "
                f"total {(t1 - t0) * 1e6:.1f}; device span {(st[1] - st[0]) / clock_hz * 1e6:.1f}")
        return t1 - t0, (st[1] - st[0]) / clock_hz

    # One untimed rehearsal of the whole bracket first (declared in `config`, like the clock warm-up).  The first region
    # of a process is not like the others: with RCCL's captured all-gather in the buckets it read 170-290 us where every
    # later one reads 114 (world size 1, tools/calls/call_r4_26.sh), its 20 kernels themselves spread over 140-250 us;
    # on the single path its launch call takes 28 us where every later one takes 14 (109 against 98 us for the region,
    # call_r4_31) — one-time costs of the bracket's first use, like the graphs' upload, not a step's.
    region_rehearsals = 0
    if os.environ.get("GTOP_BENCH_REHEARSE_REGION", "1") == "1":
        timed_region(run_bucket)
        region_rehearsals = 1
    elapsed, gpu_elapsed = timed_region(run_bucket)
    if os.environ.get("GTOP_BENCH_REGION_REPEATS"):      # diagnostic: the same region again, to see its run-to-run spread
        for k in range(int(os.environ["GTOP_BENCH_REGION_REPEATS"])):
            e2, g2 = timed_region(run_bucket)
            if rank == 0:
                log(f"region repeat {k}: host {e2 * 1e6:.1f} us, device span {g2 * 1e6:.1f} us (first: {elapsed * 1e6:.1f} / {gpu_elapsed * 1e6:.1f})")
    kern_ms_host = elapsed * 1e3 / args.steps
    kern_ms_gpu = gpu_elapsed * 1e3 / args.steps
    timed_src = (f"device wall clock ({clock_hz / 1e6:.0f} MHz) stamped by a one-lane kernel in front of the first and behind "
                 f"the last of the region's {args.steps} launches ({nbuckets} graph launch(es) of {G})")
    host_src = f"host clock around the timed region (launch call(s) and completion poll included)"

    # what the collective costs: the same K steps through the same buckets WITHOUT their all-gather, timed the same way
    kernels_only = None
    if collective and kernel_graphs is not None:
        def run_kernels_only(b):
            kernel_graphs[b % len(kernel_graphs)].replay()
        kernels_only = timed_region(run_kernels_only)

    # Cross-check of the kernel's launch time: a probe of 1 000 launches of the same kernel, graph-replayed 50 at a
    # time, HIP events on the same stream, right after the timed region (no collective).  The roofline's `frac` is
    # computed from the timed region's device-clock figure; the probe, the host clock's figure and the rocprofv3
    # average of the same command committed under profiles/ go into the line beside it.
    GR, reps = 50, 20
    scratch = torch.zeros(GR, hi - lo, dtype=tdtype, device=dev)

    def run_probe():
        for s_ in range(GR):
            ctx.eval_device(x, Df, T, scratch[s_], pipe.grad_ring[0])
    probe = run_probe
    probe_mode = "eager"
    if graphs is not None:
        try:
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, capture_error_mode="thread_local"):
                run_probe()
            gph.replay()
            probe = gph.replay
            probe_mode = "hipgraph"
        except Exception:
            torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.010:      # (the clocks are up already; this covers the probe graph's upload)
        for r in range(reps):
            probe()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for r in range(reps):
        probe()
    e1.record(stream)
    torch.cuda.synchronize()
    probe_ms = e0.elapsed_time(e1) / (reps * GR)
    probe_src = (f"HIP events around {reps * GR} launches of the same kernel ({probe_mode}, {GR} per graph) right after "
                 f"the timed region")

    by_rank = None
    if collective:
        cdev = dev if backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed, gpu_elapsed] + (list(kernels_only) if kernels_only else [0.0, 0.0]),
                            dtype=torch.float64, device=cdev)
        allr = torch.zeros(world * 4, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 4).cpu()
        by_rank = {"host_s": allr[:, 0].tolist(), "gpu_s": allr[:, 1].tolist(),
                   "kernels_only_host_s": allr[:, 2].tolist(), "kernels_only_gpu_s": allr[:, 3].tolist()}
        elapsed = float(allr[:, 0].max())          # the job's time: the slowest rank's
        # every rank must now hold every rank's costs (and gradients) of the last bucket
        last = pipe.all_costs(nbuckets - 1)
        assert torch.equal(last[rank], pipe.cost_ring[(nbuckets - 1) & 1])
        if args.gather_grads:
            assert torch.equal(pipe.all_grads(nbuckets - 1)[rank], pipe.grad_ring[(nbuckets - 1) & 1])
        if push_mode:
            # the rows the peers stored here against the process group's own all-gather of the same ring
            jl = (nbuckets - 1) & 1
            assert torch.equal(pipe.gathered[jl], pipe.library_all_gather_of_ring(jl)), \
                "rows pushed by the peers differ from the library all-gather"
        if rank == 0 and parity is not None:
            # rank r's rows of the gathered costs are that rank's shard: check rank 0's against the parity launch
            c_chk, _ = ctx.eval_device(x, Df, T)
            assert torch.equal(last[0][-1], c_chk), "gathered costs differ from a direct evaluation"

    if rank == 0:
        log(f"timed region done: {elapsed:.4f} s for {args.steps} steps")
        evals = B_total * args.steps
        value = evals / elapsed
        wkey = f"B{Bl}_m{m}_g{args.grid}_{args.dtype}"
        bpe = algorithmic_bytes(m, elem)
        launch_bytes = (hi - lo) * bpe

        def gbs(ms):
            return launch_bytes / (ms * 1e-3) / 1e9 if ms else None
        achieved = gbs(kern_ms_gpu)                            # GB/s per launch on this rank, from the timed region's device clock
        meas = measured_traffic(wkey)
        rocprof_us = meas.get("rocprof_avg_us")
        is_cfg1 = (Bl, m, args.grid, args.dtype) == (1024, 6, 200, "f64")
        par = "single GPU"
        if world > 1:
            par = (f"batch-sharded x{world}, SDF replicated, bucketed {'RCCL' if backend == 'nccl' else backend} "
                   f"all-gather of costs" + (" + last gradient of each bucket" if args.gather_grads else ""))
        out = {
            "metric": "cost+grad evals/sec (batched trajectories)",
            "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # the same region on the device's own clock (rank 0): first launch's dispatch to the last kernel's end
            "ms_per_step_gpu": kern_ms_gpu,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": f"B={Bl}/GPU x {m} segments ({3 * m + 3} ctrl pts, n={n}), {args.grid}^3 SDF, {args.dtype}"
                            + (" [BASELINE.json configs[1]]" if is_cfg1 and world == 1 else "")
                            + (" [BASELINE.json configs[3] at 8 ranks]" if Bl * world == 131072 and world == 8 else ""),
                "batch_per_gpu": Bl, "global_batch": B_total, "segments": m, "free_vars": n,
                "sdf_grid": [args.grid] * 3, "sdf_occupied_frac": float(mp.occupancy.mean()),
                "params": "opti_node.launch (ws=1, wc=5, alpha=10, d0=0.8, r=0.5), step=2",
                "parallelism": par,
                # "single": no process group, no collective, all steps in one graph — what `--gpus 1` runs whether
                # started directly or under torchrun; "collective": the bucketed path of N > 1
                "path": "collective" if collective else "single",
                "launch": launch_mode, "steps_per_bucket": G, "buckets": nbuckets, "graph_upload_replays": upload_replays,

                "clock_warmup_ms": CLOCK_WARMUP_MS, "clock_warmup_steps": clock_warmup_buckets * G,
                "untimed_region_rehearsals": region_rehearsals,
                "gather": gather_mode,
                "collective_bytes_per_bucket": (world * G * (hi - lo) * elem
                                                + (world * (hi - lo) * n * elem if args.gather_grads else 0))
                                               if collective else 0,
            },
            "roofline": {
                "bound": "hbm", "kernel": dominant_kernel(args.segments, hi - lo, args.dtype, bool(args.waves or args.spl)),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_source": "timed_region_gpu",
                "traffic": meas.get("traffic_bytes"),
                "traffic_source": meas.get("source"),
                "algorithmic_bytes_per_eval": bpe, "evals_per_launch": hi - lo,
                # the launch time by every clock that saw it (us), and the fraction each gives.  timed_region_gpu and
                # probe measure the same thing two ways and must agree; timed_region_host adds the host's launch call
                # and completion poll (a fixed ~25 us per region: 30 % of the driver's 20-step region, 0.3 % of a
                # 2 000-step one); rocprofv3's figure is its instrumented dispatch interval, which for a 4 us kernel
                # is not the kernel (an EMPTY 1 024-workgroup kernel reads 4.4 us there: DESIGN.md 5.1).
                "avg_launch_us": kern_ms_gpu * 1e3,
                "launch_us": {"timed_region_gpu": kern_ms_gpu * 1e3, "probe": probe_ms * 1e3,
                              "timed_region_host": kern_ms_host * 1e3, "rocprof": rocprof_us},
                "frac_by_source": {"timed_region_gpu": gbs(kern_ms_gpu) / HBM_PEAK_GBS,
                                   "probe": gbs(probe_ms) / HBM_PEAK_GBS,
                                   "timed_region_host": gbs(kern_ms_host) / HBM_PEAK_GBS,
                                   "rocprof": (gbs(rocprof_us * 1e-3) / HBM_PEAK_GBS) if rocprof_us else None},
                "launch_us_sources": {"timed_region_gpu": timed_src, "probe": probe_src, "timed_region_host": host_src,
                                      "rocprof": meas.get("rocprof_source")},
                "gpu_vs_probe": kern_ms_gpu / probe_ms,
                "launch_time_source": timed_src,
            },
            # what really bounds the kernel: instruction issue (the 200^3 field is cache resident)
            "roofline_issue": issue_roofline(wkey, kern_ms_gpu * 1e3),
            "parity": parity,
            "esdf_build_s": esdf_s,
        }
        if collective:
            # No scaling figure is computed here (the driver does that from the per-N lines); what the line adds is
            # what the first real N > 1 run needs to be read: who the ranks were, how far apart they finished, and
            # what the all-gather cost on top of the same buckets without it.
            ko = max(by_rank["kernels_only_host_s"]) if kernels_only else None
            out["collective"] = {
                "backend": dist.get_backend(), "ranks": dist.get_world_size(),
                # how the costs travel: "push" = gtop_push_rows (stores into the peers' mapped buffers, checked above
                # against the library all-gather), "library" = the process group's all_gather_into_tensor
                "gather_impl": "push" if push_mode else "library", "push_note": push_note,
                "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0,
                "elapsed_s_by_rank": by_rank["host_s"], "elapsed_s_min": min(by_rank["host_s"]),
                "elapsed_s_max": max(by_rank["host_s"]),
                "gpu_elapsed_s_by_rank": by_rank["gpu_s"],
                "kernels_only_elapsed_s_max": ko,
                "kernels_only_gpu_elapsed_s_by_rank": by_rank["kernels_only_gpu_s"] if kernels_only else None,
                "collective_exposed_us": (elapsed - ko) * 1e6 if ko else None,
                "collective_exposed_us_how": "max-over-ranks host time of the timed region minus the same K steps through "
                                             "the same buckets without their all-gather, timed the same way right after",
            }
        if rehearsal:
            out["rehearsal"] = True
        if world == 1 and not args.no_extras:
            from oracle import oracle
            try:   # untimed additions must never cost the main line
                out["extras"] = extras(args, ctx, batch, mp, osdf, oracle, x, Df, T, tdtype, dev)
            except Exception as e:
                out["extras"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, batch, osdf)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    if collective:
        pipe.close_push()
        dist.destroy_process_group()


def _time_evals(ctx, x, Df, T, reps, per_graph=20):
    """us per launch of gtop_eval_device (HIP events on torch's current stream), measured the way the timed region
    is: replays of a hipGraph of `per_graph` launches after a warm-up of the same graph (eager launches put the
    inter-launch gap of the host's queue into every sample, and five warm-up launches left the clocks low: the
    same kernel read 41 us here and 36 us in its own bench run).  Eager launches if the capture fails."""
    import torch
    cost, grad = ctx.eval_device(x, Df, T)
    for _ in range(5):
        ctx.eval_device(x, Df, T, cost, grad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, capture_error_mode="thread_local"):
            for _ in range(per_graph):
                ctx.eval_device(x, Df, T, cost, grad)
        nrep = max(2, reps // per_graph)
        t_w = time.perf_counter()      # warm-up: 40 ms of the same replays (sustained clocks, see the headline's probe)
        while time.perf_counter() - t_w < 0.040:
            for _ in range(nrep):
                gph.replay()
            torch.cuda.synchronize()
        e0.record()
        for _ in range(nrep):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (nrep * per_graph) * 1e3
    except Exception as e:
        print(f"bench.py: extras: graph capture failed ({e}); eager launches", file=sys.stderr)
        torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ctx.eval_device(x, Df, T, cost, grad)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def _extra_workload(ctx, oracle, osdf, b, m, grid, name, dev, label, reps=200):
    """One more BASELINE.json workload: parity gate on 256 rows, then timed graph replays (_time_evals)."""
    import torch
    dt = torch.float64 if name == "f64" else torch.float32
    xb = torch.tensor(b.x, dtype=dt, device=dev)
    Dfb = torch.tensor(b.Df.reshape(-1, 18), dtype=dt, device=dev)
    Tb = torch.tensor(b.T, dtype=dt, device=dev)
    B = xb.shape[0]
    par = parity_check(ctx, oracle, osdf, xb, Dfb, Tb, (b.T, b.Df, b.x), name)
    entry = {"workload": label, "parity": par}
    if not par["ok"]:
        entry["error"] = "parity gate failed; not timed"
        return entry
    us = _time_evals(ctx, xb, Dfb, Tb, reps)
    bpe = algorithmic_bytes(m, 4 if name == "f32" else 8)
    tr = measured_traffic(f"B{B}_m{m}_g{grid}_{name}")
    entry.update({
        "us_per_launch": us, "evals_per_s": B / (us * 1e-6),
        "roofline": {"bound": "hbm", "kernel": dominant_kernel(m, B, name, False),
                     "achieved": B * bpe / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": B * bpe / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "traffic": tr.get("traffic_bytes"), "algorithmic_bytes_per_eval": bpe,
                     "launch_us": {"probe": us, "rocprof": tr.get("rocprof_avg_us")}},
        "roofline_issue": issue_roofline(f"B{B}_m{m}_g{grid}_{name}", us)})
    return entry


def extras(args, ctx, batch, mp, osdf, oracle, x, Df, T, tdtype, dev):
    """Not part of `value`: (1) the other single-GPU BASELINE.json workloads
    (each behind its own 256-row parity gate; hipGraph replays, HIP-event time
    per launch), (2) the batched optimizer driver on the bench
    batch."""
    import numpy as np
    import torch
    from grad_traj_optimization_amd import problem
    import grad_traj_optimization_amd as gtop
    out = {"workloads": []}
    if args.grid == 200 and args.segments == 6:
        big = problem.make_trajectories(16384, 6, mp, seed=7)
        big = problem.permute(big, problem.spatial_order(big.waypoints, mp.origin, mp.map_size))
        for name in ("f32", "f64"):
            out["workloads"].append(_extra_workload(
                ctx, oracle, osdf, big, 6, 200, name, dev,
                f"B=16384 x 6 segments, 200^3 SDF, {name}" + (" [BASELINE.json configs[2]]" if name == "f32" else "")))
        # the reference's own scene length (opti_node.cpp:61-99: 11 waypoints, 10 segments) as a batch: three lanes per
        # segment, two whole trajectories per wavefront (60 busy lanes; five lanes per segment: 50)
        ten = problem.make_trajectories(8192, 10, mp, seed=9, step_len=(0.5, 1.2))
        ten = problem.permute(ten, problem.spatial_order(ten.waypoints, mp.origin, mp.map_size))
        out["workloads"].append(_extra_workload(ctx, oracle, osdf, ten, 10, 200, "f64", dev,
                                                "B=8192 x 10 segments (the reference scene's length), 200^3 SDF, f64"))
        del ten
        # configs[4]: 8 192 trajectories x 40 control points (m = 12), 400^3 field of mixed obstacle density
        t0 = time.time()
        mp4 = problem.make_map(400, density=0.04, seed=2)
        ctx4 = gtop.GtopContext(device=ctx.device)
        ctx4.init_sdf_map(mp4.map_size, mp4.origin, mp4.resolution)
        ctx4.update_sdf_map(mp4.obstacle_points())
        osdf4 = oracle_field(oracle, mp4, ctx4)
        b4 = problem.make_trajectories(8192, 12, mp4, seed=3)
        b4 = problem.permute(b4, problem.spatial_order(b4.waypoints, mp4.origin, mp4.map_size))
        e4 = _extra_workload(ctx4, oracle, osdf4, b4, 12, 400, "f64", dev,
                             "B=8192 x 12 segments (39 ctrl pts), 400^3 SDF, f64 [BASELINE.json configs[4]]")
        e4["setup_s"] = time.time() - t0
        out["workloads"].append(e4)
        del osdf4
        ctx4.close()
    if tdtype == torch.float64:
        lb, ub = gtop.GtopContext.default_bounds(batch.waypoints[:x.shape[0]])
        lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
        evals = 50
        def run_optimizer(x=x, Df=Df, T=T, lbt=lbt, ubt=ubt, evals=evals):
            """Host clock around one whole optimisation of the batch (launches + completion), the median of five runs
            after 40 ms of the same runs: after the seconds of host work before it the card needs 10-20 ms of load
            to reach its sustained clocks (tools/clock_ramp.py), and a single cold run read 5-30 % long."""
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.040:
                ctx.optimize_device(x.clone(), Df, T, lbt, ubt, evals)
                torch.cuda.synchronize()
            secs, res = [], None
            for _ in range(5):
                xo = x.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = ctx.optimize_device(xo, Df, T, lbt, ubt, evals)
                torch.cuda.synchronize()
                secs.append(time.perf_counter() - t0)
            return sorted(secs)[2], res[1]

        dt_s, cmin = run_optimizer()
        c0, _ = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
        other = {}
        for mode, key in ((1, "seconds_one_launch_per_iteration"), (0, "seconds_with_separate_update_launch")):
            ctx.set_optimizer_fusion(mode)
            other[key], _ = run_optimizer()
        ctx.set_optimizer_fusion(2)
        ctx.set_optimizer_precision("f32")      # the same loop with its evaluations in fp32 (state, update, results fp64)
        other["seconds_with_fp32_evaluations"], cmin32 = run_optimizer()
        ctx.set_optimizer_precision("f64")
        other["median_cost_ratio_fp32_evaluations_vs_fp64"] = float(torch.median(cmin32 / cmin).item())
        # what one pass of the loop costs (evaluation + CCSA-MMA update, nothing but the field's records read from HBM, no
        # launch between passes): the slope between 50 and 100 evaluations per trajectory
        dt_100, _ = run_optimizer(evals=2 * evals)
        per_pass_s = (dt_100 - dt_s) / evals
        bpe_ = algorithmic_bytes(args.segments, 8)
        other["per_pass_us"] = per_pass_s * 1e6
        other["evals_per_s_inside_the_loop"] = x.shape[0] / per_pass_s
        other["inside_the_loop_note"] = (
            "cost+gradient evaluations per second INSIDE the one-launch optimizer loop (each followed by its update; cost "
            "and gradient stay on the chip) — not the bench's metric, which delivers every evaluation's outputs to HBM "
            f"through a launch of its own; on the same algorithmic bytes it would read {x.shape[0] * bpe_ / per_pass_s / 1e9 / HBM_PEAK_GBS:.3f} "
            "of the roofline: what the per-launch floor costs the headline")
        out["optimizer"] = {
            "what": "batched CCSA-MMA on the device, whole loop in one launch (replaces per-problem NLopt LD_MMA)",
            "batch": int(x.shape[0]), "evals_per_trajectory": evals, "seconds": dt_s,
            "trajectories_optimized_per_s": x.shape[0] / dt_s, **other,
            "median_cost_ratio_after_vs_before": float(torch.median(cmin / c0).item())}
        if args.grid == 200 and args.segments == 6:
            # the same driver on the large batch (the launch rule's other regime: for fp32 evaluations two trajectories
            # per wavefront)
            lbb, ubb = gtop.GtopContext.default_bounds(big.waypoints)
            big_in = dict(x=torch.tensor(big.x, device=dev), Df=torch.tensor(big.Df.reshape(-1, 18), device=dev),
                          T=torch.tensor(big.T, device=dev), lbt=torch.tensor(lbb, device=dev),
                          ubt=torch.tensor(ubb, device=dev))
            s64, cb64 = run_optimizer(**big_in)
            ctx.set_optimizer_precision("f32")
            s32, cb32 = run_optimizer(**big_in)
            ctx.set_optimizer_precision("f64")
            out["optimizer"]["large_batch"] = {
                "batch": int(big.x.shape[0]), "evals_per_trajectory": evals, "seconds": s64,
                "trajectories_optimized_per_s": big.x.shape[0] / s64, "seconds_with_fp32_evaluations": s32,
                "trajectories_optimized_per_s_with_fp32_evaluations": big.x.shape[0] / s32,
                "median_cost_ratio_fp32_evaluations_vs_fp64": float(torch.median(cb32 / cb64).item())}
            del big_in
        if not args.no_cpu_baseline:
            # The same job on the host cores: per trajectory a serial CCSA-MMA around the callback, as the reference
            # runs NLopt's LD_MMA around costFunc (grad_traj_optimizer.cpp:137-195) — csrc/mma.hpp standing in for the
            # absent NLopt, the oracle's C restatement as the callback; same start points, bounds and 50 evaluations.
            ns = min(128, x.shape[0])
            Th, Dfh, xh = batch.T[:ns], batch.Df[:ns], batch.x[:ns]
            prm = oracle.make_params()
            xc, cc, nev, sec1 = oracle.optimize_batch(Th, Dfh, xh, lb[:ns], ub[:ns], osdf, prm, evals, nthreads=1)
            ncore = host_threads()
            reps_all = max(1, min(ncore, x.shape[0] // ns))          # more rows for the all-cores run, same work per row
            na = ns * reps_all
            _, _, _, secn = oracle.optimize_batch(batch.T[:na], batch.Df[:na], batch.x[:na], lb[:na], ub[:na], osdf, prm,
                                                  evals, nthreads=ncore)
            dev_min = cmin[:ns].cpu().numpy()
            out["optimizer"]["cpu"] = {
                "what": "serial CCSA-MMA (csrc/mma.hpp; NLopt is absent) around the oracle callback, one trajectory at "
                        "a time — the reference's optimizeTrajectory loop; L/R setup untimed",
                "kind": "port", "evals_per_trajectory": int(nev.max()),
                "trajectories_optimized_per_s": ns / sec1, "cores": 1,
                "sample": f"the first {ns} trajectories of the same batch, {sec1:.2f} s",
                "all_cores": {"trajectories_optimized_per_s": na / secn, "cores": ncore, "trajectories": na,
                              "seconds": secn},
                "max_rel_diff_of_minimum_vs_device": float(np.max(np.abs(cc - dev_min) / np.abs(cc))),
            }
    return out


def cpu_baseline(args, batch, osdf):
    """The oracle's C restatement (kind = "port": the reference itself cannot be
    built here) on a bounded sample of the SAME workload: single thread, as the
    reference's NLopt callback runs; an all-cores figure is added beside it."""
    from oracle import oracle
    prm = oracle.make_params()
    ns = min(256, batch.x.shape[0])
    Ts, Dfs, xs = batch.T[:ns], batch.Df[:ns], batch.x[:ns]
    _, _, sec = oracle.eval_batch(Ts, Dfs, xs, osdf, prm, reps=1, nthreads=1)      # calibrate
    reps = max(1, int(args.cpu_seconds * 0.6 / max(sec, 1e-6)))
    _, _, sec1 = oracle.eval_batch(Ts, Dfs, xs, osdf, prm, reps=reps, nthreads=1)
    ncore = host_threads()
    reps_all = max(1, int(args.cpu_seconds * 0.3 * ncore / max(sec, 1e-6)))
    _, _, secn = oracle.eval_batch(Ts, Dfs, xs, osdf, prm, reps=reps_all, nthreads=ncore)
    return {
        "value": ns * reps / sec1, "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": f"{ns} trajectories of the same batch x {reps} passes, callback only (L/R setup untimed), "
                  f"{sec1:.1f} s",
        "all_cores": {"value": ns * reps_all / secn, "cores": ncore, "seconds": secn},
    }


if __name__ == "__main__":
    main()
