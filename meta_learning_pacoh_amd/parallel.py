"""Multi-GPU sharding of the task dimension: rank r owns tasks r::world of the step's
(globally drawn, shared-seed) task batch and all P particles; partial sum_t mll[t,:] and partial score
[P,D] are summed with ONE all-reduce per step over RCCL / xGMI.  Nothing else is exchanged; prior term,
SVGD kernel and optimizer are replicated (deterministic, identical on every rank).

Which call carries the exchange (round 3):
* torch.distributed backend 'nccl' (one rank per GPU -- the measured configuration): the library's own pacoh_allreduce_sum, i.e.
  RCCL's ncclAllReduce enqueued on the stream the kernels run on -- no cross-stream event hop, and CAPTURED INSIDE the step's
  hipGraph, so that a multi-rank step is one graph (and four steps per replay) exactly like a single-rank one.  The communicator is
  created once (its id travels through the c10d store) and proves itself before it is trusted: an eager all-reduce and a captured +
  replayed one on known values, agreed on by all ranks; if anything fails the learners fall back to
* torch.distributed.all_reduce between two graphs per step (also what the 'gloo' CPU tests and the two-ranks-on-one-GPU tests
  use: RCCL refuses two ranks per device).  PACOH_COMM=torch forces this path, PACOH_COMM=rccl the first one regardless of backend.
The packed buffer is P*D + P floats whatever the number of tasks: splitting a shard in halves does not shrink what has to be
reduced, it doubles it -- the exchange is latency-bound (203 KB at cfg #3) and sits between the last gradient kernel and the
update, so the lever is its latency (same stream, inside the graph), not pipelining."""
import os
import warnings

import torch
import torch.distributed as dist

from . import _lib as L
from .util import gc_paused

_direct = None      # RcclComm once enable_direct_rccl() has run; False = tried and given up (torch.distributed carries the exchange)


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


_comm_serial = 0


def agree_all(ok):
    """the same verdict on every rank: MIN over ranks of `ok` through torch.distributed (a host-side tensor on gloo, a device tensor on
    nccl).  Used wherever a per-rank decision would change which collectives a rank issues -- a rank must never issue (or replay) a
    collective its peers do not"""
    _, w = world()
    if w == 1:
        return bool(ok)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0)


class _Watchdog:
    """ends the process (exit code 13, no re-exec) if the block it guards does not finish: a collective that one rank never joins --
    ncclCommInitRank, the self-test's all-reduce -- cannot be cancelled, and the launcher ends the job when a rank exits"""

    def __init__(self, seconds, what):
        import threading
        rank, w = world()
        self.active = w > 1

        def gave_up():
            import sys
            sys.stderr.write('pacoh: %s did not finish within %d s on rank %d of %d -- rerun with PACOH_COMM=torch '
                             '(torch.distributed carries the exchange between two graphs per step)\n' % (what, seconds, rank, w))
            sys.stderr.flush()
            os._exit(13)
        self.timer = threading.Timer(seconds, gave_up)
        self.timer.daemon = True

    def __enter__(self):
        if self.active:
            self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def _exchange_unique_id(rank, w):
    """rank 0's fresh RCCL communicator id on every rank: through the c10d key-value store (host side, no collective, no device
    traffic) when torch.distributed has one, else as a broadcast object"""
    global _comm_serial
    _comm_serial += 1
    if w == 1:
        return L.comm_unique_id()
    store = None
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    if store is not None:
        key = 'pacoh_rccl_uid_%d' % _comm_serial
        if rank == 0:
            store.set(key, L.comm_unique_id())
        return bytes(store.get(key))
    uid = [L.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    return uid[0]


class RcclComm:
    """pacoh_comm_* handle of this rank (include/pacoh_gp.h, section 8e): created collectively by every rank.  graph_ok: the
    all-reduce survived capture into a hipGraph and two replays with the right sums on EVERY rank (self_test)"""

    SELF_TEST_TIMEOUT_S = 180

    def __init__(self, self_test=True):
        rank, w = world()
        self.rank, self.world_size = rank, w
        self.handle = None
        self.graph_ok = False
        # (the watchdog starts BEFORE the id exchange and ncclCommInitRank: both block until every rank has joined)
        with _Watchdog(self.SELF_TEST_TIMEOUT_S, 'creating the RCCL communicator'):
            self.handle = L.comm_init(_exchange_unique_id(rank, w), rank, w)
        if self_test:
            self.graph_ok = self._self_test()

    def all_reduce_(self, buf):
        return L.allreduce_sum(buf, self.handle)

    def _agree(self, ok, dev=None):
        """the same verdict on every rank: a rank must never replay a collective its peers failed to capture -- it would wait for
        them forever"""
        return agree_all(ok)

    def _self_test(self):
        """eager all-reduce, then the same call captured in a hipGraph and replayed twice, on values whose sums are known.  Three
        phases (eager / capture / replay), each closed by an agreement of all ranks: the next phase only starts if every rank got
        through the previous one"""
        w, rank = self.world_size, self.rank
        dev = torch.device('cuda', torch.cuda.current_device())
        total = w * (w + 1) / 2.0
        buf = graph = None

        def phase(fn):
            try:
                ok = bool(fn())
            except Exception as exc:                       # a capture the RCCL build does not support, a launch error, ...
                warnings.warn('pacoh: in-graph RCCL all-reduce self-test failed on rank %d: %r' % (rank, exc))
                ok = False
            return self._agree(ok, dev)

        def eager():
            nonlocal buf
            buf = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=dev)
            self.all_reduce_(buf)
            torch.cuda.synchronize()
            return (buf == total).all()

        def capture():
            nonlocal graph
            buf.fill_(float(rank + 1))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.all_reduce_(buf)                      # (warm-up on the capture stream's side, as capture_graph does)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            buf.fill_(float(rank + 1))
            g = torch.cuda.CUDAGraph()
            kw = {'capture_error_mode': 'thread_local'} if w > 1 else {}      # (torch.distributed's watchdog thread: see capture_graph)
            with warnings.catch_warnings(), gc_paused():   # (no garbage collection inside a capture: util.gc_paused)
                warnings.simplefilter('ignore')            # (one rank: RCCL enqueues nothing for an in-place sum -> "graph is empty")
                with torch.cuda.graph(g, **kw):
                    self.all_reduce_(buf)
            graph = g
            return True

        def replay():
            graph.replay()                                 # every rank: total
            graph.replay()                                 # every rank: w * total
            torch.cuda.synchronize()
            return (buf == w * total).all()

        # A collective that one rank never joins cannot be cancelled: if the test does not finish, say so and end the process (the
        # launcher then ends the job) instead of leaving a silent hang for the caller's own timeout to find
        with _Watchdog(self.SELF_TEST_TIMEOUT_S, 'the in-graph RCCL all-reduce self-test'):
            ok = phase(eager) and phase(capture) and phase(replay)
        del graph
        return ok

    def close(self):
        if self.handle is not None:
            L.comm_destroy(self.handle)
            self.handle = None


def enable_direct_rccl():
    """collective call (all ranks): route the step's all-reduce through pacoh_allreduce_sum from now on"""
    global _direct
    if not _direct:
        _direct = RcclComm()
    return _direct


def disable_direct_rccl():
    global _direct
    if _direct:
        _direct.close()
    _direct = None


def _want_direct():
    """policy (see the module comment): RCCL on the compute stream whenever every rank has its own GPU"""
    mode = os.environ.get('PACOH_COMM', '')
    if mode == 'torch':
        return False
    if mode == 'rccl':
        return True                                       # (also at world size 1: tests capture the collective on one GPU)
    return world()[1] > 1 and dist.get_backend() == 'nccl'


def _direct_comm():
    """the RcclComm carrying the step's exchange, or None (torch.distributed carries it).  Collective on first use: every rank
    reaches it at the same point of the program (the learners call collective_in_graph() when they set up their step)"""
    global _direct
    if _direct is None and _want_direct():
        comm, ok = None, True
        # Preconditions FIRST, agreed by all ranks, before anybody enters the blocking ncclCommInitRank: what can fail on ONE rank
        # alone (librccl not loadable there, a symbol missing, no id obtainable) would otherwise leave its healthy peers inside
        # pacoh_comm_init until their watchdog ends the job -- instead of the fallback to torch.distributed this function promises
        # (ADVICE r4).  pacoh_comm_unique_id exercises exactly that: dlopen + symbol binding + ncclGetUniqueId, no peer involved.
        try:
            L.comm_unique_id()
        except Exception as exc:
            warnings.warn('pacoh: rank %d cannot use RCCL directly (%r); all ranks fall back to torch.distributed' % (world()[0], exc))
            ok = False
        if not agree_all(ok):
            ok = False
        else:
            try:
                comm = RcclComm(self_test=False)
            except Exception as exc:
                warnings.warn('pacoh: RCCL communicator could not be created on rank %d (%r)' % (world()[0], exc))
                ok = False
        # EVERY rank reports, and all ranks take the same branch: one rank falling back to torch.distributed while its peers sit in
        # the communicator's self-test would build different step graphs and hang the first exchange
        if agree_all(ok) and comm is not None:
            comm.graph_ok = comm._self_test()              # (its three phases are agreed on by all ranks themselves)
            ok = comm.graph_ok or world()[1] == 1
        else:
            ok = False
        if ok:
            _direct = comm
        else:
            # the communicator does not exist everywhere, or does not survive graph capture here: use it eagerly?  No -- one code
            # path less to trust on hardware this build never saw: torch.distributed between two graphs per step
            if comm is not None:
                comm.close()
            if world()[0] == 0:
                warnings.warn('pacoh: torch.distributed carries the all-reduce (no usable in-graph RCCL communicator)')
            _direct = False
    return _direct or None


def collective_in_graph():
    """True: the step's exchange may be captured inside the step graph (world size 1: there is none, or a forced world-size-1
    communicator that passed its self-test); False: two graphs per step around an eager torch.distributed.all_reduce"""
    comm = _direct_comm()
    if comm is not None:
        return comm.graph_ok
    return world()[1] == 1


def packed_score_buffer(P, D, dtype, device):
    """one buffer holding score[P,D] followed by lik[P], so that the step's single all-reduce needs no packing copies:
    returns (buf, score view, lik view)"""
    buf = torch.empty(P * D + P, dtype=dtype, device=device)
    return buf, buf[:P * D].view(P, D), buf[P * D:]


def all_reduce_buffer_(buf):
    """buf := sum over ranks of buf, in place (identity at world size 1): the step's one exchange"""
    comm = _direct_comm()
    if comm is not None:
        comm.all_reduce_(buf)
    elif world()[1] > 1:
        with L._Timed('allreduce_torch'):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def all_reduce_sum_(lik, score, packed=None):
    """sum [P] and [P,D] over ranks with one collective on a packed buffer; identity at world size 1.
    `packed`: the buffer of packed_score_buffer() whose views lik / score are (reduced in place, no copies)"""
    if packed is not None:
        all_reduce_buffer_(packed)
        return lik, score
    _, w = world()
    if w == 1:
        return lik, score
    P, D = score.shape
    buf = torch.cat([score.reshape(-1), lik.reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf[P * D:].reshape(P), buf[:P * D].reshape(P, D)
