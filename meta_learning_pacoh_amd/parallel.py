"""Multi-GPU sharding of the task dimension: rank r owns tasks r::world of the step's
(globally drawn, shared-seed) task batch and all P particles; partial sum_t mll[t,:] and partial score
[P,D] are summed with ONE all-reduce per step -- torch.distributed 'nccl' backend = RCCL over xGMI on
the GPU box, 'gloo' in the CPU tests.  Nothing else is exchanged; prior term, SVGD kernel and optimizer
are replicated (deterministic, identical on every rank)."""
import torch
import torch.distributed as dist


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


def packed_score_buffer(P, D, dtype, device):
    """one buffer holding score[P,D] followed by lik[P], so that the step's single all-reduce needs no packing copies:
    returns (buf, score view, lik view)"""
    buf = torch.empty(P * D + P, dtype=dtype, device=device)
    return buf, buf[:P * D].view(P, D), buf[P * D:]


def all_reduce_sum_(lik, score, packed=None):
    """sum [P] and [P,D] over ranks with one collective on a packed buffer; identity at world size 1.
    `packed`: the buffer of packed_score_buffer() whose views lik / score are (reduced in place, no copies)"""
    _, w = world()
    if w == 1:
        return lik, score
    P, D = score.shape
    if packed is not None:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        return lik, score
    buf = torch.cat([score.reshape(-1), lik.reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf[P * D:].reshape(P), buf[:P * D].reshape(P, D)
