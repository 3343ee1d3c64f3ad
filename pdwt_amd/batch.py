"""Batch split of the hot path over the GPUs of one node (BASELINE.json north_star).

The reference is single-GPU.  Rows of a batched-1D array are independent signals
(reference src/separable.cu:213) and whole 2D images are independent, so the batch shards with NO
data-path collective: every rank owns a contiguous block of rows (or whole images) and a private
``Wavelets`` instance on its own GPU.  The only exchange steps are
  * ``norm1()``: one all-reduce(SUM) of a single float64 per call (RCCL over xGMI on GPUs);
  * optional gather of results to rank 0 for inspection (outside any timed region).
One process per GPU: ``torch.distributed`` with backend "nccl" (= RCCL) in production, "gloo" in the
CPU tests (tests/test_batch_shard.py), where the per-shard engine is injected.
"""
import numpy as np


def shard_rows(n_rows, world, rank):
    """Contiguous row block [start, start+count) of rank `rank`; the first n_rows % world ranks get one more."""
    base, rem = divmod(int(n_rows), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class ShardedBatch:
    """A batch of independent 1D signals (rows) or 2D images, split across the ranks of a process group.

    engine_factory(local_array) -> object with forward(), inverse(), soft_threshold(beta, app, norm),
    norm1_f64(), get_image(), coeffs.  Default: pdwt_amd.Wavelets on this rank's GPU.
    """

    def __init__(self, local_rows, wname, levels, ndim=1, do_swt=0, group=None, engine_factory=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # with a process group the collectives always run, also for world_size 1 (one code path; a 1-rank "nccl" group is how
        # the RCCL branch is exercised on a 1-GPU box: tests/test_batch_gpu.py)
        self.collective = dist.is_initialized()
        if engine_factory is None:
            import pdwt_amd

            def engine_factory(a):
                return pdwt_amd.Wavelets(a, wname, levels, do_swt=do_swt, ndim=ndim)
        # device tensors (torch, or anything with __cuda_array_interface__) go to the engine as they are: memisonhost = 0
        on_device = hasattr(local_rows, "is_cuda") or hasattr(local_rows, "__cuda_array_interface__")
        self.W = engine_factory(local_rows if on_device else np.ascontiguousarray(local_rows))

    def forward(self):
        self.W.forward()

    def inverse(self):
        self.W.inverse()

    def soft_threshold(self, beta, do_thresh_appcoeffs=0, normalize=0):
        self.W.soft_threshold(beta, do_thresh_appcoeffs, normalize)  # elementwise: shard-local

    def norm1(self):
        """L1 norm of ALL coefficients of the whole batch: per-shard double partial + all-reduce(SUM)."""
        import torch
        local = float(self.W.norm1_f64())
        if not self.collective:
            return local
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([local], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def norm1_per_rank(self):
        """The per-shard partial sums of norm1(), in rank order, on every rank (all-gather of one float64 each): what the
        all-reduce of norm1() must add up to -- used by bench.py's untimed check of the N > 1 path."""
        import torch
        local = float(self.W.norm1_f64())
        if not self.collective:
            return [local]
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([local], dtype=torch.float64, device=dev)
        outs = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        return [float(o.item()) for o in outs]

    def gather_image(self, dst=0):
        """All shards' reconstructed rows stacked in rank order on rank `dst` (None elsewhere).

        A tensor gather, not a pickled object: over "nccl" (= RCCL over xGMI) the rows travel GPU to GPU from the engine's device
        buffer (zero-copy view) and only rank `dst` copies the stacked batch to the host; over "gloo" they travel as host tensors.
        Shards of different heights are padded to the tallest one for the collective (the counts come from an all-gather of one int)."""
        import torch
        if not self.collective:
            return self.W.get_image()
        nccl = self.dist.get_backend(self.group) == "nccl"
        dev = "cuda" if nccl else "cpu"
        if nccl and hasattr(self.W, "image_view"):
            self.W.sync()
            local = torch.as_tensor(self.W.image_view(), device="cuda")
        else:
            local = torch.from_numpy(np.ascontiguousarray(self.W.get_image())).to(dev)
        if local.dim() == 1:
            local = local.reshape(1, -1)
        rows = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(rows) for _ in range(self.world)]
        self.dist.all_gather(counts, rows, group=self.group)
        counts = [int(c.item()) for c in counts]
        hmax = max(counts)
        if local.shape[0] < hmax:  # pad to the tallest shard (at most one row more than any other: shard_rows)
            local = torch.cat([local, local.new_zeros((hmax - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
        local = local.contiguous()
        dst_global = self.dist.get_global_rank(self.group, dst) if self.group is not None else dst
        if self.rank == dst:
            parts = [torch.empty_like(local) for _ in range(self.world)]
            self.dist.gather(local, parts, dst=dst_global, group=self.group)
            return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)
        self.dist.gather(local, None, dst=dst_global, group=self.group)
        return None
