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

`enable_push(ctx)` replaces the library all-gather by point-to-point stores (csrc/gtop_push.hip, gtop_push_rows): every
rank maps every peer's gathered buffers into its own process (CUDA-IPC handles exchanged once through the process
group) and ONE kernel behind a bucket's last evaluation stores the rank's rows into its slot of every buffer — no
collective launch, no ring protocol; the owners read after the closing barrier.  Verified against the process group's own
all-gather before it is trusted; any failure leaves the library path in place.

`ResultGatherPipeline` is backend-agnostic (nccl on GPUs; gloo on CPU for the
tests, and gloo with device buffers staged through the host for rehearsals on a
one-GPU box); the evaluation itself is injected as `run_bucket_fn(ring_index)`,
which must enqueue `steps_per_bucket` evaluations writing costs into
`cost_ring[ring_index][step]` and gradients into `grad_ring[ring_index]`.
"""
import os

import torch
import torch.distributed as dist

from .problem import shard_range  # re-exported: the batch partition rule

__all__ = ["shard_range", "ResultGatherPipeline", "CostGatherPipeline"]


def _host_id():
    """Something two processes share exactly when they run on one machine (boot id; the host name as a fallback)."""
    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            return f.read().strip()
    except OSError:
        import socket
        return socket.gethostname()


class _DeviceBytes:
    """A raw device allocation as something torch.as_tensor can wrap without copying (__cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


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
        self._push = None                   # enable_push: (ctx, per ring: destination addresses of the cost rows / of the gradient)
        self._shared = None                 # (ctx, the peers' mapped buffers, this rank's own shareable allocations)

    def enable_push(self, ctx):
        """Switch the gathers to gtop_push_rows.  Every rank must call this together, before the first bucket.
        The gathered buffers move into allocations of their own that a peer process can map (gtop_shared_alloc: an IPC
        handle names a whole allocation, not a slice of torch's caching allocator); the 64-byte handles travel through
        the process group; every rank maps every peer's buffers FOR ITS OWN DEVICE (gtop_shared_open: lazy peer access
        — what RCCL's own point-to-point transport does).  Returns (ok, why): ok only if EVERY rank mapped every
        peer's buffers and a probe push arrived intact everywhere — decided together, so that all ranks take the same
        path; otherwise nothing has changed and the library all-gather stays in place."""
        if not self.collective:
            return False, "no collective"
        why = "ok"
        dsts = None
        opened, owned = [], []
        old = [self.gathered[0], self.gathered[1]] + (list(self.grad_gathered) if self.gather_grads else [])
        payload, views = None, []
        try:
            dev = old[0].device
            mine = []
            for t in old:
                nbytes = t.numel() * t.element_size()
                ptr, handle = ctx.shared_alloc(nbytes)
                owned.append(ptr)
                mine.append(handle)
                views.append(torch.as_tensor(_DeviceBytes(ptr, nbytes), device=dev).view(t.dtype).view(t.shape))
            payload = (int(ctx.device), mine, _host_id())
        except Exception as e:
            why = f"allocating shareable buffers failed: {e!r}"
        allh = [None] * self.world
        dist.all_gather_object(allh, payload)   # (every rank takes part, whatever happened above)
        if any(h is None for h in allh):
            if payload is not None:
                why = "another rank could not allocate its buffers"
        elif any(h[2] != allh[self.rank][2] for h in allh):          # (one node: every process counts the devices alike)
            why = "the ranks run on several hosts: buffers can only be mapped within one"
        else:
            try:
                if os.environ.get("GTOP_PUSH_FORCE_FAIL") == str(self.rank):     # (tests: the fallback must hold on every rank)
                    raise RuntimeError("forced by GTOP_PUSH_FORCE_FAIL")
                base = []                       # [rank][buffer] -> device address of that rank's buffer in THIS process
                for r in range(self.world):
                    if r == self.rank:
                        base.append(list(owned))
                        continue
                    row = []
                    owner_dev, handles, _ = allh[r]
                    for h in handles:
                        p = ctx.shared_open(h, owner_dev)
                        opened.append(p)
                        row.append(p)
                    base.append(row)
                es = old[0].element_size()
                cost_bytes = self.G * self.cost_ring[0].shape[1] * es
                dsts = {"cost": [[base[r][j] + self.rank * cost_bytes for r in range(self.world)] for j in range(2)],
                        "cost_bytes": cost_bytes}
                if self.gather_grads:
                    gb = self.grad_ring[0].numel() * es
                    dsts["grad"] = [[base[r][2 + j] + self.rank * gb for r in range(self.world)] for j in range(2)]
                    dsts["grad_bytes"] = gb
            except Exception as e:              # (a handle that cannot be opened, no peer access, ...)
                why = f"mapping the peers' buffers failed: {e!r}"
                dsts = None
        # probe: every rank pushes a rank-coded pattern through the real path; every rank checks every slot
        ok_local = dsts is not None
        saved = (self.gathered, self.grad_gathered)
        if ok_local:
            try:
                self.gathered = [views[0], views[1]]
                if self.gather_grads:
                    self.grad_gathered = [views[2], views[3]]
                self._push = (ctx, dsts)
                for j in range(2):
                    self.cost_ring[j].fill_(float(self.rank + 1) + 0.25 * j)
                    self.gathered[j].fill_(-7.0)
                torch.cuda.synchronize()
            except Exception as e:
                ok_local, why = False, f"probe set-up failed: {e!r}"
        dist.barrier()                          # nobody pushes into a buffer its owner is still presetting
        if ok_local:
            try:
                for j in range(2):
                    ctx.push_rows(self.cost_ring[j], dsts["cost"][j], dsts["cost_bytes"])
                torch.cuda.synchronize()
            except Exception as e:
                ok_local, why = False, f"probe push failed: {e!r}"
        dist.barrier()
        if ok_local:
            for j in range(2):
                g = self.gathered[j].view(self.world, -1)
                want = torch.arange(1, self.world + 1, dtype=g.dtype, device=g.device) + 0.25 * j
                if not bool((g == want[:, None]).all()):
                    ok_local, why = False, f"probe rows of ring {j} did not all arrive"
        flags = [None] * self.world
        dist.all_gather_object(flags, (bool(ok_local), why))
        ok = all(f[0] for f in flags)
        for j in range(2):
            self.cost_ring[j].zero_()
        torch.cuda.synchronize()
        if ok:
            for j in range(2):
                self.gathered[j].zero_()
            torch.cuda.synchronize()
            self._shared = (ctx, opened, owned)
        else:
            self._push = None
            self.gathered, self.grad_gathered = saved
            why = "; ".join(f"rank {r}: {f[1]}" for r, f in enumerate(flags) if not f[0])
            views = None
            dist.barrier()                      # every rank has stopped using the mappings
            for p in opened:
                try:
                    ctx.shared_close(p)
                except Exception:
                    pass
            dist.barrier()                      # ... and closed them, before their owners free them
            for p in owned:
                try:
                    ctx.shared_free(p)
                except Exception:
                    pass
        dist.barrier()
        return ok, why

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
        if self.collective and not self.gather_in_bucket_fn and self._push is not None:
            self.gather_now(j)              # stores enqueued behind the bucket's kernels: nothing to wait for on the host
        elif self.collective and not self.gather_in_bucket_fn:
            self._pending[j].append(self._gather(self.gathered[j], self.cost_ring[j]))
            if self.gather_grads:
                self._pending[j].append(self._gather(self.grad_gathered[j], self.grad_ring[j]))

    def gather_now(self, j, clock_minmax=None):
        """The bucket's all-gathers, synchronously on the current stream (for capture into the bucket's graph).
        clock_minmax (push path only): the gather kernel also takes the device-clock stamp as it starts."""
        if self._push is not None:
            ctx, d = self._push
            ctx.push_rows(self.cost_ring[j], d["cost"][j], d["cost_bytes"], clock_minmax=clock_minmax)
            if self.gather_grads:
                ctx.push_rows(self.grad_ring[j], d["grad"][j], d["grad_bytes"])
            return
        dist.all_gather_into_tensor(self.gathered[j], self.cost_ring[j])
        if self.gather_grads:
            dist.all_gather_into_tensor(self.grad_gathered[j], self.grad_ring[j])

    def close_push(self):
        """Unmap the peers' buffers and release this rank's own (the owners must outlive the mappings: barriers)."""
        if self._shared is None:
            return
        ctx, opened, owned = self._shared
        torch.cuda.synchronize()
        dist.barrier()
        self._push = None
        self._shared = None
        self.gathered = [g.clone() for g in self.gathered]          # (results stay readable: ordinary tensors again)
        if self.gather_grads:
            self.grad_gathered = [g.clone() for g in self.grad_gathered]
        torch.cuda.synchronize()
        for p in opened:
            ctx.shared_close(p)
        dist.barrier()
        for p in owned:
            ctx.shared_free(p)

    def library_all_gather_of_ring(self, j):
        """Ring j's costs of every rank through the process group's own all-gather (a check for the push path)."""
        out = torch.zeros_like(self.gathered[j])
        if self._staged:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, self.cost_ring[j].cpu())
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, self.cost_ring[j])
        return out

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
