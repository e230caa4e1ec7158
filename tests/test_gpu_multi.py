"""The N-rank bench path on ONE card: (1) RCCL initialised at world size 1 so
that the process group, all_gather_into_tensor and hipGraph replay meet on
hardware; (2) `bench.py --gpus 2` started directly (no torchrun): the
self-launcher, two processes running the HIP kernels, collectives over gloo
because RCCL refuses two ranks on one device; (3) one batch sharded over two
gtop_ctx on device 0, bit-identical to the unsharded run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from grad_traj_optimization_amd import problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # exactly ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_rank_path_with_rccl_at_world_size_one():
    r = _bench(["--gpus", "1", "--steps", "40", "--warmup", "7", "--bucket", "10", "--gather-grads",
                "--no-extras", "--no-cpu-baseline"], {"GTOP_BENCH_FORCE_DIST": "1"})
    assert r["rehearsal"] is True and r["n_gpus"] == 1
    assert r["steps"] == 40 and r["warmup"] == 7           # --warmup honoured exactly
    assert r["config"]["launch"] == "hipgraph" and r["config"]["steps_per_bucket"] == 10
    assert r["config"]["collective_bytes_per_bucket"] == 10 * 1024 * 8 + 1024 * 45 * 8
    assert r["parity"]["ok"] and r["value"] > 1e4
    assert r["config"]["gather"] == "point-to-point stores (gtop_push_rows), captured in each bucket's hipGraph"
    assert r["collective"]["gather_impl"] == "push"
    assert r["config"]["untimed_region_rehearsals"] == 1          # (declared)
    assert r["config"]["path"] == "collective" and r["config"]["buckets"] == 4
    col = r["collective"]
    assert col["backend"] == "nccl" and col["ranks"] == col["rccl_ranks"] == 1
    assert len(col["elapsed_s_by_rank"]) == 1 and col["elapsed_s_min"] == col["elapsed_s_max"] > 0
    assert col["collective_exposed_us"] is not None and col["kernels_only_elapsed_s_max"] > 0
    assert 0 < r["ms_per_step_gpu"] <= r["ms_per_step"] * 1.02      # the device's clock sees no more than the host's


@pytest.mark.timeout(900)
def test_bench_rank_path_with_host_side_gathers():
    # the fallback when RCCL cannot be captured: graphs of kernels, one async all-gather call per bucket
    r = _bench(["--gpus", "1", "--steps", "40", "--warmup", "7", "--bucket", "10", "--gather-grads",
                "--no-extras", "--no-cpu-baseline"],
               {"GTOP_BENCH_FORCE_DIST": "1", "GTOP_BENCH_CAPTURE_GATHER": "0", "GTOP_BENCH_GATHER": "library"})
    assert r["config"]["launch"] == "hipgraph" and r["config"]["gather"].startswith("host call per bucket")
    assert r["parity"]["ok"] and r["value"] > 1e4


@pytest.mark.timeout(900)
def test_bench_short_collective_run_and_its_buckets():
    """The driver's N > 1 command is --steps 20.  With the all-gathers captured into the bucket graphs (RCCL) a gather
    runs in line behind its bucket's kernels, so the run is ONE bucket — one exposed gather, not two (bench.py has the
    measurements of the overlapping alternatives); with host-side asynchronous gathers, which do run beside the next
    bucket's kernels, it is split in two."""
    args = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"]
    r = _bench(args, {"GTOP_BENCH_FORCE_DIST": "1", "GTOP_BENCH_GATHER": "library"})      # RCCL's all-gather, captured
    assert r["config"]["steps_per_bucket"] == 20 and r["config"]["buckets"] == 1 and r["steps"] == 20
    assert r["config"]["gather"] == "captured in each bucket's hipGraph" and r["collective"]["collective_exposed_us"] is not None
    assert r["collective"]["gather_impl"] == "library"
    r = _bench(args, {"GTOP_BENCH_FORCE_DIST": "1"})                                      # the default: stores
    assert r["config"]["buckets"] == 1 and r["collective"]["gather_impl"] == "push"
    r = _bench(args, {"GTOP_BENCH_FORCE_DIST": "1", "GTOP_BENCH_CAPTURE_GATHER": "0", "GTOP_BENCH_GATHER": "library"})
    assert r["config"]["steps_per_bucket"] == 10 and r["config"]["buckets"] == 2 and r["steps"] == 20


@pytest.mark.timeout(900)
def test_gpus_1_is_the_single_path_however_it_is_started():
    """`--gpus 1` must be byte for byte the path BENCH measures — no process group, no collective, every step in one
    graph, the device-clock stamps — whether the driver starts bench.py directly or under torchrun."""
    args = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"]
    direct = _bench(args, {})
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", BENCH] + args,
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    spawned = json.loads(lines[0])
    for r in (direct, spawned):
        assert r["config"]["path"] == "single" and "collective" not in r and "rehearsal" not in r
        assert r["config"]["steps_per_bucket"] == 20 and r["config"]["buckets"] == 1 and r["config"]["gather"] == "none"
        assert r["roofline"]["frac_source"] == "timed_region_gpu"
        assert r["config"]["clock_warmup_ms"] == 40.0 and r["config"]["clock_warmup_steps"] >= 20
        assert r["config"]["untimed_region_rehearsals"] == 1
        f = r["roofline"]["frac_by_source"]
        assert abs(f["timed_region_gpu"] - f["probe"]) <= 0.08 * f["probe"], f      # two clocks, one kernel
        assert abs(r["roofline"]["frac"] - f["timed_region_gpu"]) < 1e-12
    assert direct["config"] == {**spawned["config"], "clock_warmup_steps": direct["config"]["clock_warmup_steps"]}


@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks_on_one_card():
    r = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--batch", "512", "--gather-grads"],
               {"GTOP_BENCH_BACKEND": "gloo", "GTOP_BENCH_SHARE_DEVICE": "1"})
    assert r["rehearsal"] is True and r["n_gpus"] == 2 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 1024 and r["config"]["batch_per_gpu"] == 512
    assert r["parity"]["ok"] and r["value"] > 1e4
    assert "extras" not in r and "cpu_baseline" not in r   # N = 1 only
    col = r["collective"]
    assert col["backend"] == "gloo" and col["ranks"] == 2 and col["rccl_ranks"] == 0
    assert len(col["elapsed_s_by_rank"]) == 2 and col["elapsed_s_max"] >= col["elapsed_s_min"] > 0
    # TWO PROCESSES storing into each other's buffers (mapped over CUDA-IPC; here both on one card): the push path the
    # ranks of a real node take, checked inside bench.py against the process group's own all-gather
    assert col["gather_impl"] == "push", col["push_note"]
    assert r["config"]["gather"].startswith("point-to-point stores") and r["config"]["launch"] == "hipgraph"
    assert r["config"]["buckets"] == 1 and r["config"]["path"] == "collective"


@pytest.mark.timeout(900)
def test_a_rank_that_cannot_map_its_peers_sends_every_rank_to_the_library_gather():
    """The safety net of the push path: rank 1 fails to map rank 0's buffers (forced); BOTH ranks must then drop the
    mappings they hold and gather through the process group — the run completes, says which path it took and why."""
    r = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--batch", "512"],
               {"GTOP_BENCH_BACKEND": "gloo", "GTOP_BENCH_SHARE_DEVICE": "1", "GTOP_PUSH_FORCE_FAIL": "1"})
    col = r["collective"]
    assert col["gather_impl"] == "library" and "rank 1" in col["push_note"] and "GTOP_PUSH_FORCE_FAIL" in col["push_note"]
    assert r["parity"]["ok"] and r["value"] > 1e4 and r["n_gpus"] == 2


def test_two_contexts_on_one_device_equal_the_unsharded_batch(gtop):
    """configs[3] in miniature: contiguous shards on separate contexts (each with its
    own replicated field, built independently) and separate streams, no reduction
    => bit-identical to one context evaluating the whole batch."""
    import torch
    mp = problem.make_map((80, 80, 40), density=0.03, seed=31)
    b = problem.make_trajectories(2048, 6, mp, seed=32)
    dev = torch.device("cuda:0")
    ctxs = []
    for _ in range(3):
        c = gtop.GtopContext(device=0)
        c.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
        c.update_sdf_map(mp.obstacle_points())
        c.set_launch_geometry(1, 3)       # pin the kernel variant: shard size must not pick another summation order
        ctxs.append(c)
    assert np.array_equal(ctxs[0].get_sdf(), ctxs[1].get_sdf())
    for td in (torch.float64, torch.float32):
        x = torch.tensor(b.x, dtype=td, device=dev)
        Df = torch.tensor(b.Df.reshape(-1, 18), dtype=td, device=dev)
        T = torch.tensor(b.T, dtype=td, device=dev)
        c_all, g_all = ctxs[2].eval_device(x, Df, T)
        parts = []
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        for r in range(2):
            lo, hi = problem.shard_range(2048, r, 2)
            with torch.cuda.stream(streams[r]):
                parts.append(ctxs[r].eval_device(x[lo:hi].contiguous(), Df[lo:hi].contiguous(),
                                                 T[lo:hi].contiguous()))
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([p[0] for p in parts]), c_all)
        assert torch.equal(torch.cat([p[1] for p in parts]), g_all)


@pytest.mark.timeout(900)
def test_configs3_batch_in_eight_shards_equals_the_whole(gtop):
    """BASELINE.json configs[3] at full size on one card: 131 072 trajectories over the 200^3 field, evaluated whole
    and as the eight contiguous 16 384-row shards the eight ranks of `bench.py --gpus 8 --batch 16384` would own.
    Rows are independent and the auto rule picks the same body for a shard as for the whole batch (both past the
    4 096-row switch to two trajectories per wavefront), so the concatenation is bit-identical; plus cost >= 1e-3 (:417-418) and finite gradients."""
    import torch
    mp = problem.make_map(200, density=0.02, seed=0)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    B, W = 131072, 8
    b = problem.make_trajectories(B, 6, mp, seed=4)
    dev = torch.device("cuda:0")
    x = torch.tensor(b.x, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    c_all, g_all = ctx.eval_device(x, Df, T)
    torch.cuda.synchronize()
    assert torch.isfinite(c_all).all() and torch.isfinite(g_all).all() and (c_all >= 1e-3).all()
    covered = 0
    for r in range(W):
        lo, hi = problem.shard_range(B, r, W)
        assert lo == covered and hi - lo == B // W
        covered = hi
        c, g = ctx.eval_device(x[lo:hi].contiguous(), Df[lo:hi].contiguous(), T[lo:hi].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(c, c_all[lo:hi]) and torch.equal(g, g_all[lo:hi])
    assert covered == B
