"""Multi-GPU harness: one process per GPU, batch sharded, costs all-gathered.

The path is embarrassingly parallel: trajectories are
independent, the distance field is replicated, so the data path has no
collective.  The only exchange is collecting results: every rank ends up with
every rank's per-trajectory costs.  At one evaluation per ~10 us a collective
per step would be pure latency, so steps are grouped in buckets: each bucket's
costs land in one ring buffer and ONE all-gather (RCCL over xGMI when the
backend is "nccl") ships the whole bucket while the next bucket's kernels run.

`CostGatherPipeline` is backend-agnostic (nccl on GPUs, gloo on CPU for the
tests); the evaluation itself is injected as `run_bucket_fn(ring_index)`, which
must enqueue `steps_per_bucket` evaluations writing costs into
`cost_ring[ring_index][step]`.
"""
import torch
import torch.distributed as dist

from .problem import shard_range  # re-exported: the batch partition rule

__all__ = ["shard_range", "CostGatherPipeline"]


class CostGatherPipeline:
    def __init__(self, world_size, rank, steps_per_bucket, local_batch, dtype, device, run_bucket_fn):
        self.world, self.rank = world_size, rank
        self.G = steps_per_bucket
        self.run_bucket_fn = run_bucket_fn
        # two rings: bucket b computes into ring b&1 while ring (b-1)&1 is in flight
        self.cost_ring = [torch.zeros(self.G, local_batch, dtype=dtype, device=device) for _ in range(2)]
        # output of all_gather_into_tensor = rank-major concatenation along dim 0
        self.gathered = ([torch.zeros(world_size * self.G, local_batch, dtype=dtype, device=device)
                          for _ in range(2)] if world_size > 1 else None)
        self._pending = [None, None]

    def run_bucket(self, b):
        j = b & 1
        if self._pending[j] is not None:   # ring j is about to be overwritten
            self._pending[j].wait()
            self._pending[j] = None
        self.run_bucket_fn(j)
        if self.world > 1:
            self._pending[j] = dist.all_gather_into_tensor(self.gathered[j], self.cost_ring[j], async_op=True)

    def drain(self):
        for j in range(2):
            if self._pending[j] is not None:
                self._pending[j].wait()
                self._pending[j] = None

    def all_costs(self, b):
        """(world, G, local_batch) costs of bucket b on every rank (after drain)."""
        if self.world == 1:
            return self.cost_ring[b & 1].unsqueeze(0)
        return self.gathered[b & 1].view(self.world, self.G, -1)
