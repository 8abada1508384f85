"""Multi-GPU decomposition of the MAP objective: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" on CPU for
the tests).  The reference is single-process; these are the shardings its
objective admits:

  channels  rank r owns a block of channels of x, g, IRLS weights and of every
            LR frame (the reference's split_channels view,
            irls_map_solver.cpp:200-262).  No gradient traffic; a joint solve
            only all-reduces the scalars (cost, CG dot products).
  frames    rank r owns frames {k : k mod world == r} and a replica of x; the
            partial data-term gradients are summed with one all-reduce of C*N
            elements per evaluation; the regulariser term is evaluated once
            (rank 0) before the reduction.

The local evaluator is any callable  local_eval(x, terms) -> (cost, grad)
working on this rank's shard: srmap.Problem.eval_device on the GPU (the tests plug
a CPU evaluator over gloo).
"""
TERM_DATA, TERM_REG, TERM_ALL = 1, 2, 3


def frame_shard(num_frames, world, rank):
    """Frames owned by `rank` (round robin: shift phases spread evenly)."""
    return list(range(rank, num_frames, world))


def channel_shard(num_channels, world, rank):
    """Contiguous channel block [c0, c1) owned by `rank`."""
    base, rem = divmod(num_channels, world)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


class ShardedObjective:
    """ObjectiveFunction::ComputeAllTerms over all ranks.

    eval(x) returns (cost, grad): cost is the global cost on every rank; grad is
    the full gradient (frames mode) or this rank's channel block (channels mode).
    """

    def __init__(self, mode, local_eval, dist=None, all_reduce_tensor=None):
        assert mode in ("channels", "frames")
        self.mode, self.local_eval, self.dist = mode, local_eval, dist
        self._to_tensor = all_reduce_tensor

    def rank(self):
        return self.dist.get_rank() if self.dist is not None else 0

    def terms_for_rank(self):
        if self.mode == "frames" and self.rank() != 0:
            return TERM_DATA  # the regulariser is evaluated once, on rank 0
        return TERM_ALL

    def eval(self, x):
        cost, grad = self.local_eval(x, self.terms_for_rank())
        if self.dist is None or self.dist.get_world_size() == 1:
            return cost, grad
        import torch
        c = torch.tensor([cost], dtype=torch.float64, device=grad.device if hasattr(grad, "device") else "cpu")
        if self.mode == "frames" and grad is not None:
            self.dist.all_reduce(grad)      # sum of per-rank data-term gradients (+ reg from rank 0)
        self.dist.all_reduce(c)             # global cost
        return float(c.item()), grad
