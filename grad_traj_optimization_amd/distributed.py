"""Multi-GPU harness: one process per GPU, batch sharded, results all-gathered.

The path is embarrassingly parallel: trajectories are
independent, the distance field is replicated, so the data path has no
collective.  The only exchange is collecting results: every rank ends up with
every rank's per-trajectory costs.  At one evaluation per ~5 us a collective
per step would be pure latency, so steps are grouped in buckets: each bucket's
costs land in one ring buffer and ONE all-gather (RCCL over xGMI when the
backend is "nccl") ships the whole bucket while the next bucket's kernels run.

Optionally (`gather_grads`) the gradient of the bucket's last step is gathered
as well — what a global optimizer step that runs once per bucket would consume
(B x n x e bytes, 45 MiB at B = 131 072 fp64).  xGMI is point to
point, so that gather is per-link bound; it rides behind the next bucket's
kernels like the cost gather does.

`ResultGatherPipeline` is backend-agnostic (nccl on GPUs; gloo on CPU for the
tests, and gloo with device buffers staged through the host for rehearsals on a
one-GPU box); the evaluation itself is injected as `run_bucket_fn(ring_index)`,
which must enqueue `steps_per_bucket` evaluations writing costs into
`cost_ring[ring_index][step]` and gradients into `grad_ring[ring_index]`.
"""
import torch
import torch.distributed as dist

from .problem import shard_range  # re-exported: the batch partition rule

__all__ = ["shard_range", "ResultGatherPipeline", "CostGatherPipeline"]


class ResultGatherPipeline:
    def __init__(self, world_size, rank, steps_per_bucket, local_batch, dtype, device, run_bucket_fn,
                 n_free=0, gather_grads=False, collective=None):
        self.world, self.rank = world_size, rank
        self.G = steps_per_bucket
        self.run_bucket_fn = run_bucket_fn
        self.collective = (world_size > 1) if collective is None else bool(collective)
        self.gather_grads = bool(gather_grads) and self.collective
        device = torch.device(device)
        # gloo moves host memory only: device buffers are staged (rehearsal path, synchronous)
        self._staged = self.collective and device.type == "cuda" and dist.get_backend() == "gloo"
        # two rings: bucket b computes into ring b&1 while ring (b-1)&1 is in flight
        self.cost_ring = [torch.zeros(self.G, local_batch, dtype=dtype, device=device) for _ in range(2)]
        # output of all_gather_into_tensor = rank-major concatenation along dim 0
        self.gathered = ([torch.zeros(world_size * self.G, local_batch, dtype=dtype, device=device)
                          for _ in range(2)] if self.collective else None)
        if n_free:
            g0 = torch.zeros(local_batch, n_free, dtype=dtype, device=device)
            # one gradient buffer is enough unless a gather may still be reading the other bucket's
            self.grad_ring = [g0, torch.zeros_like(g0) if self.gather_grads else g0]
        else:
            self.grad_ring = [None, None]
            self.gather_grads = False
        self.grad_gathered = ([torch.zeros(world_size * local_batch, n_free, dtype=dtype, device=device)
                               for _ in range(2)] if self.gather_grads else None)
        self._pending = [[], []]
        # True when run_bucket_fn enqueues the all-gathers itself (captured into the bucket's hipGraph behind its last
        # kernel: one graph launch per bucket, no collective call from the host)
        self.gather_in_bucket_fn = False

    def _gather(self, out, src):
        if self._staged:
            host_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host_out, src.cpu())
            out.copy_(host_out)
            return None
        return dist.all_gather_into_tensor(out, src, async_op=True)

    def _wait(self, j):
        for w in self._pending[j]:
            if w is not None:
                w.wait()
        self._pending[j] = []

    def run_bucket(self, b):
        j = b & 1
        self._wait(j)                       # ring j is about to be overwritten
        self.run_bucket_fn(j)
        if self.collective and not self.gather_in_bucket_fn:
            self._pending[j].append(self._gather(self.gathered[j], self.cost_ring[j]))
            if self.gather_grads:
                self._pending[j].append(self._gather(self.grad_gathered[j], self.grad_ring[j]))

    def gather_now(self, j):
        """The bucket's all-gathers, synchronously on the current stream (for capture into the bucket's graph)."""
        dist.all_gather_into_tensor(self.gathered[j], self.cost_ring[j])
        if self.gather_grads:
            dist.all_gather_into_tensor(self.grad_gathered[j], self.grad_ring[j])

    def drain(self):
        for j in range(2):
            self._wait(j)

    def all_costs(self, b):
        """(world, G, local_batch) costs of bucket b on every rank (after drain)."""
        if not self.collective:
            return self.cost_ring[b & 1].unsqueeze(0)
        return self.gathered[b & 1].view(self.world, self.G, -1)

    def all_grads(self, b):
        """(world, local_batch, n) gradients of bucket b's last step on every rank (after drain)."""
        if not self.gather_grads:
            return self.grad_ring[b & 1].unsqueeze(0)
        return self.grad_gathered[b & 1].view(self.world, -1, self.grad_ring[0].shape[1])


CostGatherPipeline = ResultGatherPipeline   # round-1 name
