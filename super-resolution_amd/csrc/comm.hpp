// comm.hpp -- internal interface of the communicator (comm.hip) used by the
// sharded evaluation and solver.
#pragma once
#include "srmap_internal.hpp"

namespace srmap {

int comm_rank(const srmap_comm* c);
// The communicator's side stream (non-blocking) and its two events, created on first use.
int comm_side(srmap_comm* c, hipStream_t* side, hipEvent_t* ev_x, hipEvent_t* ev_halo);
int comm_world(const srmap_comm* c);
bool comm_overlap(const srmap_comm* c);  // row shards: run the halo exchange under the interior tile rows
// In-place all-reduce of a device buffer (op 0 = sum, 1 = max), enqueued on `st` (RCCL) or staged through the
// caller's host callback (synchronises `st`).  No-op for world 1 / null communicator.
int comm_allreduce(srmap_comm* c, void* dev, size_t count, int dtype, int op, hipStream_t st);
// nseg segments: send[i] (send_seg elements each) -> rank dst, recv[i] (recv_seg elements each) <- rank src
// (rank < 0 or a zero size: that side is absent).
int comm_exchange(srmap_comm* c, const void* const* send, int dst, void* const* recv, int src, int nseg,
                  size_t send_seg, size_t recv_seg, int dtype, hipStream_t st);

// Both directions of a halo exchange in one RCCL group: `a` travels to rank `down` / arrives from `up`, `b` the other way.
int comm_exchange2(srmap_comm* c, const void* const* send_a, void* const* recv_a, size_t send_a_seg, size_t recv_a_seg,
                   const void* const* send_b, void* const* recv_b, size_t send_b_seg, size_t recv_b_seg, int up, int down,
                   int nseg, int dtype, hipStream_t st);
// Gradient and cost summed over the ranks of `c` in one RCCL group.
int comm_allreduce_grad_cost(srmap_comm* c, void* g, size_t count, int dtype, double* cost, hipStream_t st);

// Halo refresh of x for the shard (rows: boundary rows with the two row neighbours; channels: one plane with each
// channel neighbour when the problem carries halo planes).  x is this rank's [C][H][W] device buffer.
int shard_exchange_x(srmap_problem* p, srmap_comm* c, const srmap_shard_desc* sd, void* x_dev, hipStream_t st);
// The sharded evaluation without the final cost read-back: exchange, local evaluation, gradient / cost all-reduce
// for frame shards.  The (local, or for frame shards global) cost is left in p->d_cost[0].
int shard_eval(srmap_problem* p, srmap_comm* c, const srmap_shard_desc* sd, unsigned terms, void* x_dev, void* g_dev,
               hipStream_t st);

}  // namespace srmap
