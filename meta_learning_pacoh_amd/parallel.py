"""Multi-GPU sharding of the task dimension: rank r owns tasks r::world of the step's
(globally drawn, shared-seed) task batch and all P particles; partial sum_t mll[t,:] and partial score
[P,D] are summed with ONE all-reduce per step -- torch.distributed 'nccl' backend = RCCL over xGMI on
the GPU box, 'gloo' in the CPU tests.  Nothing else is exchanged; prior term, SVGD kernel and optimizer
are replicated (deterministic, identical on every rank).

The collective itself is torch.distributed's by default.  PACOH_COMM=rccl (or enable_direct_rccl()) switches the packed-buffer
reduce to the library's own pacoh_allreduce_sum: the same RCCL all-reduce, enqueued on the stream the kernels run on, so the
step has no cross-stream event hop; torch.distributed is then only the out-of-band channel for the communicator id."""
import os

import torch
import torch.distributed as dist

from . import _lib as L

_direct = None      # RcclComm once enable_direct_rccl() has run


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(indices, rank=None, world_size=None):
    """this rank's slice of the global task-index list (strided: balances ragged task sizes)"""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return indices[rank::world_size]


def broadcast_seed(seed):
    """rank 0's seed on every rank (None at world size 1: nothing to agree on).  Multi-rank runs draw the GLOBAL task batch and
    PACOH-VI's reparameterisation noise from host generators on every rank; unseeded, each rank would draw its own and the
    all-reduce would silently sum gradients of different objectives"""
    _, w = world()
    if w == 1:
        return seed
    box = [int(seed) if seed is not None else int.from_bytes(os.urandom(4), 'little') & 0x7fffffff]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def check_same_draws(idx_rows, sc_rows):
    """PACOH_CHECK_RANKS=1 (debug): assert that every rank drew the same pre-factors / learning rates for the chunk, i.e. that
    the host generators of the ranks are in step (the local index lists differ by construction, the scalars must not)"""
    _, w = world()
    if w == 1 or os.environ.get('PACOH_CHECK_RANKS', '0') != '1':
        return
    mine = [float(sum(r)) for r in sc_rows]          # (rows may be lists or a numpy array)
    box = [None] * w
    dist.all_gather_object(box, mine)
    assert all(b == box[0] for b in box), 'ranks drew different task batches: host RNG streams are out of step'


class RcclComm:
    """pacoh_comm_* handle of this rank (include/pacoh_gp.h, section 8e): created collectively by every rank"""

    def __init__(self):
        rank, w = world()
        uid = [L.comm_unique_id() if rank == 0 else None]
        if w > 1:
            dist.broadcast_object_list(uid, src=0)
        self.world_size = w
        self.handle = L.comm_init(uid[0], rank, w)

    def all_reduce_(self, buf):
        return L.allreduce_sum(buf, self.handle)

    def close(self):
        if self.handle is not None:
            L.comm_destroy(self.handle)
            self.handle = None


def enable_direct_rccl():
    """collective call (all ranks): route the step's all-reduce through pacoh_allreduce_sum from now on"""
    global _direct
    if _direct is None:
        _direct = RcclComm()
    return _direct


def disable_direct_rccl():
    global _direct
    if _direct is not None:
        _direct.close()
        _direct = None


def _direct_comm():
    if _direct is None and os.environ.get('PACOH_COMM', '') == 'rccl':
        enable_direct_rccl()
    return _direct


def packed_score_buffer(P, D, dtype, device):
    """one buffer holding score[P,D] followed by lik[P], so that the step's single all-reduce needs no packing copies:
    returns (buf, score view, lik view)"""
    buf = torch.empty(P * D + P, dtype=dtype, device=device)
    return buf, buf[:P * D].view(P, D), buf[P * D:]


def all_reduce_buffer_(buf):
    """buf := sum over ranks of buf, in place (identity at world size 1): the step's one exchange"""
    _, w = world()
    if w == 1:
        return buf
    comm = _direct_comm()
    if comm is not None:
        comm.all_reduce_(buf)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def all_reduce_sum_(lik, score, packed=None):
    """sum [P] and [P,D] over ranks with one collective on a packed buffer; identity at world size 1.
    `packed`: the buffer of packed_score_buffer() whose views lik / score are (reduced in place, no copies)"""
    _, w = world()
    if w == 1:
        return lik, score
    P, D = score.shape
    if packed is not None:
        comm = _direct_comm()
        if comm is not None:
            comm.all_reduce_(packed)
        else:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        return lik, score
    buf = torch.cat([score.reshape(-1), lik.reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf[P * D:].reshape(P), buf[:P * D].reshape(P, D)
