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

Multi-GPU: the batch shards across ranks (weak scaling: per-GPU batch fixed),
the distance field is replicated, and the only collective is a bucketed RCCL
all-gather of the per-trajectory costs (one collective per bucket of steps,
overlapped with the next bucket's kernels).

Rank 0 prints ONE JSON line (contract in the task statement), extended with
`roofline` (dominant kernel: gtop_eval_kernel, HBM-bound, algorithmic bytes
of SURVEY.md §8d) and `cpu_baseline` (the oracle's C restatement timed on this
box's host cores; N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def measured_traffic(key):
    """HBM-side bytes per launch of gtop_eval_kernel for this workload, from the
    rocprofv3 --pmc passes committed under profiles/ (tools/pmc_collect.sh +
    tools/pmc_summary.py; counters cannot be read from inside this process).
    None when this workload has not been profiled."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        return t.get(key)
    except (OSError, ValueError):
        return None



def algorithmic_bytes(m, elem):
    """per evaluation, e*[(9(m-1)+18+m) + (1+9(m-1))] + e*8*30*m."""
    n = 9 * (m - 1)
    return elem * ((n + 18 + m) + (1 + n)) + elem * 8 * 30 * m


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
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra workloads reported under 'extras'")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-sort", action="store_true",
                    help="keep the generator's random batch order (default: Morton-ordered for L2 locality)")
    ap.add_argument("--waves", type=int, default=0, help="waves per trajectory block (0 = auto)")
    ap.add_argument("--spl", type=int, default=0, help="samples per lane (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import grad_traj_optimization_amd as gtop
    from grad_traj_optimization_amd import problem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)   # nccl == RCCL on ROCm

    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    elem = 8 if args.dtype == "f64" else 4
    m, Bl = args.segments, args.batch
    B_total = Bl * world

    # ---- synthetic inputs (same seeds on every rank; each rank keeps its shard) ----
    mp = problem.make_map(args.grid, density=args.density, seed=0)
    ctx = gtop.GtopContext(device=local_rank)
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

    G = max(1, min(args.bucket, args.steps))
    while args.steps % G:
        G -= 1
    nbuckets = args.steps // G
    grad = torch.zeros(hi - lo, n, dtype=tdtype, device=dev)

    # ---- parity gate (rank 0): HIP vs oracle on a subsample of this rank's shard ----
    parity = None
    if rank == 0:
        from oracle import oracle
        nchk = min(256, hi - lo)
        # the whole shard, i.e. the launch geometry (kernel variant) that is timed below; its first rows are checked
        c_dev, g_dev = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
        c_dev, g_dev = c_dev[:nchk], g_dev[:nchk]
        dist_host = ctx.get_sdf()
        osdf = oracle.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
        osdf.dist[:] = dist_host.reshape(-1)
        c_ref, g_ref, _ = oracle.eval_batch(batch.T[lo:lo + nchk], batch.Df[lo:lo + nchk], batch.x[lo:lo + nchk],
                                            osdf, oracle.make_params(), nthreads=host_threads())
        c = c_dev.double().cpu().numpy()
        gg = g_dev.double().cpu().numpy()
        rc = float(np.max(np.abs(c - c_ref) / np.abs(c_ref)))
        rg = float(np.max(np.max(np.abs(gg - g_ref), axis=1) / np.max(np.abs(g_ref), axis=1)))
        tol = 1e-5 if args.dtype == "f64" else 5e-2
        parity = {"n": nchk, "max_rel_cost": rc, "max_rel_grad": rg, "tol": tol, "ok": bool(rc <= tol and rg <= tol)}
        log(f"parity {parity}")
        if not parity["ok"]:
            print(f"bench.py: PARITY FAILED {parity}", file=sys.stderr)
            sys.exit(3)

    # ---- launch plan: one hipGraph per cost ring buffer, G steps each ----
    from grad_traj_optimization_amd.distributed import CostGatherPipeline
    stream = torch.cuda.current_stream(dev)
    graphs = None

    def run_bucket_eager(j):
        for s in range(G):
            ctx.eval_device(x, Df, T, pipe.cost_ring[j][s], grad)

    def run_bucket_graph(j):
        graphs[j].replay()

    pipe = CostGatherPipeline(world, rank, G, hi - lo, tdtype, dev, run_bucket_eager)
    launch_mode = "eager"
    if not args.no_graph:
        try:
            graphs = []
            for j in range(2):
                gph = torch.cuda.CUDAGraph()
                # thread_local: a HIP call from another thread (RCCL's watchdog) must not break the capture
                with torch.cuda.graph(gph, capture_error_mode="thread_local"):
                    run_bucket_eager(j)
                graphs.append(gph)
            pipe.run_bucket_fn = run_bucket_graph
            launch_mode = "hipgraph"
        except Exception as e:   # capture unsupported: fall back to eager launches, say so
            print(f"bench.py: graph capture failed ({e}); using eager launches", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()
    if rank == 0:
        log(f"launch mode {launch_mode}, {G} steps per bucket")
    run_bucket, drain = pipe.run_bucket, pipe.drain

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warmup ----
    wb = max(1, math.ceil(args.warmup / G)) if args.warmup > 0 else 0
    for b in range(wb):
        run_bucket(b)
    drain()
    torch.cuda.synchronize()

    # ---- timed region: exactly K = nbuckets*G steps ----
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for b in range(nbuckets):
        run_bucket(b)
    ev1.record(stream)
    drain()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps     # avg per launch on the launch stream (HIP events)

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # every rank must now hold every rank's costs of the last bucket
        last = pipe.all_costs(nbuckets - 1)
        assert torch.equal(last[rank], pipe.cost_ring[(nbuckets - 1) & 1])

    if rank == 0:
        log(f"timed region done: {elapsed:.4f} s for {args.steps} steps")
        evals = B_total * args.steps
        value = evals / elapsed
        wkey = f"B{Bl}_m{m}_g{args.grid}_{args.dtype}"
        bpe = algorithmic_bytes(m, elem)
        achieved = (hi - lo) * bpe / (kern_ms * 1e-3) / 1e9    # GB/s, per launch on this rank
        out = {
            "metric": "cost+grad evals/sec (batched trajectories)",
            "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": wb * G,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": f"B={Bl}/GPU x {m} segments ({3 * m + 3} ctrl pts, n={n}), {args.grid}^3 SDF, {args.dtype}"
                            + (" [BASELINE.json configs[1]]" if (Bl, m, args.grid, args.dtype) == (1024, 6, 200, "f64") else ""),
                "batch_per_gpu": Bl, "global_batch": B_total, "segments": m, "free_vars": n,
                "sdf_grid": [args.grid] * 3, "sdf_occupied_frac": float(mp.occupancy.mean()),
                "params": "opti_node.launch (ws=1, wc=5, alpha=10, d0=0.8, r=0.5), step=2",
                "parallelism": f"batch-sharded x{world}, SDF replicated, bucketed RCCL all-gather of costs"
                               if world > 1 else "single GPU",
                "launch": launch_mode, "steps_per_bucket": G,
            },
            "roofline": {
                "bound": "hbm", "kernel": "gtop_eval_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (measured_traffic(wkey) or {}).get("traffic_bytes"),
                "traffic_source": (measured_traffic(wkey) or {}).get("source"),
                "algorithmic_bytes_per_eval": bpe, "evals_per_launch": hi - lo,
                "avg_launch_us": kern_ms * 1e3,
            },
            "parity": parity,
            "esdf_build_s": esdf_s,
        }
        if world == 1 and not args.no_extras:
            try:   # untimed additions must never cost the main line
                out["extras"] = extras(args, ctx, batch, mp, x, Df, T, tdtype, dev)
            except Exception as e:
                out["extras"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, batch, mp, ctx)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


def _time_evals(ctx, x, Df, T, reps):
    """us per launch of gtop_eval_device on torch's current stream (HIP events)."""
    import torch
    cost, grad = ctx.eval_device(x, Df, T)
    for _ in range(5):
        ctx.eval_device(x, Df, T, cost, grad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ctx.eval_device(x, Df, T, cost, grad)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def extras(args, ctx, batch, mp, x, Df, T, tdtype, dev):
    """Not part of `value`: (1) the other single-GPU BASELINE.json workloads on the
    same map (eager launches, HIP-event time per launch), (2) the batched
    optimizer driver on the bench batch."""
    import torch
    from grad_traj_optimization_amd import problem
    import grad_traj_optimization_amd as gtop
    out = {"workloads": []}
    if args.grid == 200 and args.segments == 6:
        big = problem.make_trajectories(16384, 6, mp, seed=7)
        big = problem.permute(big, problem.spatial_order(big.waypoints, mp.origin, mp.map_size))
        for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
            xb = torch.tensor(big.x, dtype=dt, device=dev)
            Dfb = torch.tensor(big.Df.reshape(-1, 18), dtype=dt, device=dev)
            Tb = torch.tensor(big.T, dtype=dt, device=dev)
            us = _time_evals(ctx, xb, Dfb, Tb, 200)
            bpe = algorithmic_bytes(6, 4 if name == "f32" else 8)
            out["workloads"].append({
                "workload": f"B=16384 x 6 segments, 200^3 SDF, {name}" + (" [BASELINE.json configs[2]]" if name == "f32" else ""),
                "us_per_launch": us, "evals_per_s": 16384 / (us * 1e-6),
                "roofline_frac": 16384 * bpe / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
    if tdtype == torch.float64:
        lb, ub = gtop.GtopContext.default_bounds(batch.waypoints[:x.shape[0]])
        lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
        evals = 50
        xo = x.clone()
        ctx.optimize_device(xo, Df, T, lbt, ubt, evals)       # warm-up
        torch.cuda.synchronize()
        xo = x.clone()
        t0 = time.perf_counter()
        _, cmin = ctx.optimize_device(xo, Df, T, lbt, ubt, evals)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        c0, _ = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
        other = {}
        for mode, key in ((1, "seconds_one_launch_per_iteration"), (0, "seconds_with_separate_update_launch")):
            ctx.set_optimizer_fusion(mode)
            xo2 = x.clone()
            ctx.optimize_device(xo2, Df, T, lbt, ubt, evals)
            torch.cuda.synchronize()
            xo2 = x.clone()
            t0 = time.perf_counter()
            ctx.optimize_device(xo2, Df, T, lbt, ubt, evals)
            torch.cuda.synchronize()
            other[key] = time.perf_counter() - t0
        ctx.set_optimizer_fusion(2)
        out["optimizer"] = {
            "what": "batched CCSA-MMA on the device, whole loop in one launch (replaces per-problem NLopt LD_MMA)",
            "batch": int(x.shape[0]), "evals_per_trajectory": evals, "seconds": dt_s,
            "trajectories_optimized_per_s": x.shape[0] / dt_s, **other,
            "median_cost_ratio_after_vs_before": float(torch.median(cmin / c0).item())}
    return out


def cpu_baseline(args, batch, mp, ctx):
    """The oracle's C restatement (kind = "port": the reference itself cannot be
    built here) on a bounded sample of the SAME workload: single thread, as the
    reference's NLopt callback runs; an all-cores figure is added beside it."""
    from oracle import oracle
    osdf = oracle.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    osdf.dist[:] = ctx.get_sdf().reshape(-1)
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
