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

  rows      rank r owns a band of HR rows of x, g, w and the matching LR rows of
            every frame (spatial sharding, the strong-scaling path).  Its problem
            lives on the band extended by `band_halo` rows on each inner side;
            before an evaluation the neighbours exchange those boundary rows of
            x (point-to-point), the cost counts owned rows only
            (srmap_problem_set_cost_rows) and is all-reduced; the gradient of the
            owned rows needs no exchange.
  grid      frames x channels (2-D): ranks form `channel_groups` x `frame_groups`;
            a rank owns one channel block and one frame shard, the gradient is
            all-reduced inside its channel group only, the cost over all ranks.

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


def grid_coords(world, rank, frame_groups):
    """(channel_group, frame_group) of `rank` in a channel_groups x frame_groups grid."""
    assert world % frame_groups == 0
    return rank // frame_groups, rank % frame_groups


def band_halo(scale, blur_ksize, max_abs_shift, reg_reach):
    """Rows of x beyond a band that the gradient and cost of its owned rows depend on:
    data term 2*|shift| + 2*(blur half width) (HR pixel -> shifted -> LR stencil -> warped x),
    regulariser `reg_reach` (BTV range R / TV 1), rounded up to a multiple of the scale."""
    hb = (max(blur_ksize, 1) - 1) // 2
    need = max(2 * int(max_abs_shift) + 2 * hb, int(reg_reach))
    return ((need + scale - 1) // scale) * scale


def row_band(hr_height, scale, world, rank, halo):
    """Owned HR rows [r0, r1) of `rank` (multiples of the scale, balanced) and the rows
    [e0, e1) its band problem covers (owned + halo, clipped to the image)."""
    lr_rows = hr_height // scale
    base, rem = divmod(lr_rows, world)
    i0 = rank * base + min(rank, rem)
    i1 = i0 + base + (1 if rank < rem else 0)
    r0, r1 = i0 * scale, (i1 * scale if rank < world - 1 else hr_height)
    return (r0, r1), (max(0, r0 - halo), min(hr_height, r1 + halo))


class BandObjective:
    """ObjectiveFunction::ComputeAllTerms with x sharded by HR row bands.

    `x_band` ([C][e1-e0][W], torch tensor) is this rank's extended band: the owned
    rows are authoritative, the halo rows are refreshed from the neighbours by
    exchange_halos().  local_eval(x_band) -> (owned_cost, grad_band) evaluates the band
    problem (cost restricted to the owned rows).  eval() returns the global cost and
    the gradient of the OWNED rows.
    """

    def __init__(self, owned, extent, local_eval, dist=None):
        self.r0, self.r1 = owned
        self.e0, self.e1 = extent
        self.local_eval, self.dist = local_eval, dist

    def exchange_halos(self, x_band):
        d = self.dist
        if d is None or d.get_world_size() == 1:
            return
        rank, world = d.get_rank(), d.get_world_size()
        up, dn = self.r0 - self.e0, self.e1 - self.r1   # halo rows above / below the owned rows
        o0, o1 = self.r0 - self.e0, self.r1 - self.e0   # owned rows inside the band
        reqs, recv = [], []
        # my top owned rows -> previous rank's bottom halo; my bottom owned rows -> next rank's top halo
        if rank > 0:
            send_up = x_band[:, o0:o0 + self._peer_halo(rank - 1, "dn"), :].contiguous()
            reqs.append(d.isend(send_up, rank - 1))
            buf = x_band.new_empty((x_band.shape[0], up, x_band.shape[2]))
            reqs.append(d.irecv(buf, rank - 1)); recv.append((buf, slice(0, up)))
        if rank < world - 1:
            send_dn = x_band[:, o1 - self._peer_halo(rank + 1, "up"):o1, :].contiguous()
            reqs.append(d.isend(send_dn, rank + 1))
            buf = x_band.new_empty((x_band.shape[0], dn, x_band.shape[2]))
            reqs.append(d.irecv(buf, rank + 1)); recv.append((buf, slice(o1, o1 + dn)))
        for r in reqs:
            r.wait()
        for buf, sl in recv:
            x_band[:, sl, :] = buf

    def set_peer_halos(self, halos):
        """halos[r] = (rows above, rows below) of every rank's band (from row_band)."""
        self._halos = halos

    def _peer_halo(self, peer, side):
        return self._halos[peer][0 if side == "up" else 1]

    def eval(self, x_band):
        self.exchange_halos(x_band)
        cost, grad = self.local_eval(x_band)
        o0, o1 = self.r0 - self.e0, self.r1 - self.e0
        g_owned = grad[:, o0:o1, :]
        if self.dist is not None and self.dist.get_world_size() > 1:
            import torch
            c = torch.tensor([cost], dtype=torch.float64, device=grad.device if hasattr(grad, "device") else "cpu")
            self.dist.all_reduce(c)
            cost = float(c.item())
        return cost, g_owned


class ShardedObjective:
    """ObjectiveFunction::ComputeAllTerms over all ranks.

    eval(x) returns (cost, grad): cost is the global cost on every rank; grad is
    the full gradient (frames mode) or this rank's channel block (channels / grid mode).
    """

    def __init__(self, mode, local_eval, dist=None, all_reduce_tensor=None, frame_groups=1, channel_group=None):
        """mode "grid": `frame_groups` ranks share a channel block; `channel_group` is the
        torch.distributed group of those ranks (dist.new_group), the gradient all-reduce runs inside it."""
        assert mode in ("channels", "frames", "grid")
        self.mode, self.local_eval, self.dist = mode, local_eval, dist
        self._to_tensor = all_reduce_tensor
        self.frame_groups, self.channel_group = frame_groups, channel_group

    def rank(self):
        return self.dist.get_rank() if self.dist is not None else 0

    def terms_for_rank(self):
        if self.mode == "frames" and self.rank() != 0:
            return TERM_DATA  # the regulariser is evaluated once, on rank 0
        if self.mode == "grid" and self.rank() % self.frame_groups != 0:
            return TERM_DATA  # ... once per channel block
        return TERM_ALL

    def eval(self, x):
        cost, grad = self.local_eval(x, self.terms_for_rank())
        if self.dist is None or self.dist.get_world_size() == 1:
            return cost, grad
        import torch
        c = torch.tensor([cost], dtype=torch.float64, device=grad.device if hasattr(grad, "device") else "cpu")
        if self.mode == "frames" and grad is not None:
            self.dist.all_reduce(grad)      # sum of per-rank data-term gradients (+ reg from rank 0)
        if self.mode == "grid" and grad is not None and self.frame_groups > 1:
            self.dist.all_reduce(grad, group=self.channel_group)  # within the channel block only
        self.dist.all_reduce(c)             # global cost
        return float(c.item()), grad
