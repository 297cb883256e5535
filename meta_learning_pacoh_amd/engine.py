"""Device engine of the batched task-GP path: from flattened prior parameters theta[P, D] and a batch
of tasks to per-(task, particle) log marginal likelihoods and d(sum)/d(theta) -- the work the
reference does with a Python loop over tasks around VectorizedGP.forward / ExactGP + autograd
(meta_learn/random_gp.py:54-89,204-222; GPR_meta_mll.py:104-117).  Every arithmetic step is a HIP
kernel behind the C ABI (include/pacoh_gp.h); this file only sequences launches on the current
stream and owns the parameter layout.
"""
import os
import time
import warnings
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from . import parallel
from .util import gc_paused


def nn_param_layout(input_dim, output_dim, layer_sizes):
    """bias BEFORE weight per layer, weight row-major [out,in] (meta_learn/models.py:319-323,351-384)"""
    layout = OrderedDict()
    prev = input_dim
    for i, size in enumerate(layer_sizes):
        layout['fc_%i.bias' % (i + 1)] = size
        layout['fc_%i.weight' % (i + 1)] = size * prev
        prev = size
    layout['out.bias'] = output_dim
    layout['out.weight'] = output_dim * prev
    return layout


class ParamLayout:
    """Flattened prior-parameter vector.  Block order follows VectorizedGP.__init__
    (meta_learn/random_gp.py:33-51): mean block, kernel_nn block, lengthscale_raw, [outputscale_raw,]
    noise_raw.  `with_outputscale` adds the ScaleKernel parameter of PACOH-MAP (GPR_meta_mll.py:218)."""

    def __init__(self, input_dim, mean_module='NN', covar_module='NN', mean_nn_layers=(32, 32),
                 kernel_nn_layers=(32, 32), feature_dim=2, with_outputscale=False):
        # 'COS': gpytorch.kernels.CosineKernel on the raw inputs (module objects only, modules.py) -- ONE raw period parameter, kept
        # in the lengthscale slot; the device sees it replicated over the input dimensions (PACOH_KERNEL_COSINE in pacoh_gp.h)
        assert mean_module in ('NN', 'constant', 'zero') and covar_module in ('NN', 'SE', 'COS')
        self.input_dim, self.mean_module, self.covar_module = input_dim, mean_module, covar_module
        self.kernel_code = L.KERNEL_COSINE if covar_module == 'COS' else L.KERNEL_RBF
        self.mean_nn_layers, self.kernel_nn_layers = tuple(mean_nn_layers), tuple(kernel_nn_layers)
        self.feature_dim = feature_dim if covar_module == 'NN' else input_dim
        self.with_outputscale = with_outputscale
        blocks = OrderedDict()
        if mean_module == 'NN':
            for k, v in nn_param_layout(input_dim, 1, mean_nn_layers).items():
                blocks['mean_nn.' + k] = v
        elif mean_module == 'constant':
            blocks['constant_mean'] = 1
        if covar_module == 'NN':
            for k, v in nn_param_layout(input_dim, feature_dim, kernel_nn_layers).items():
                blocks['kernel_nn.' + k] = v
        blocks['lengthscale_raw'] = 1 if covar_module == 'COS' else self.feature_dim
        if with_outputscale:
            blocks['outputscale_raw'] = 1
        blocks['noise_raw'] = 1
        self.blocks = blocks
        self.slices = OrderedDict()
        idx = 0
        for k, v in blocks.items():
            self.slices[k] = (idx, idx + v)
            idx += v
        self.D = idx

    def parameter_shapes(self):
        return OrderedDict((k, torch.Size((v,))) for k, v in self.blocks.items())

    def block_range(self, prefix):
        keys = [k for k in self.blocks if k.startswith(prefix)]
        if not keys:
            return None
        return self.slices[keys[0]][0], self.slices[keys[-1]][1]

    def hyper_prior_mean_std(self, weight_prior_std, bias_prior_std):
        """meta_learn/random_gp.py:126-151"""
        mean, std = torch.zeros(self.D), torch.ones(self.D)
        for name, (lo, hi) in self.slices.items():
            if name == 'noise_raw':
                mean[lo:hi] = -1.0
            elif 'mean_nn' in name or 'kernel_nn' in name:
                std[lo:hi] = weight_prior_std if 'weight' in name else bias_prior_std
        return mean, std


class TaskBatch:
    """Tasks packed for the device: x[T, n_max, d], y[T, n_max], n_valid[T] (int32); ragged tasks are
    zero padded (the kernels ignore rows >= n_valid)."""

    def __init__(self, tasks, device, dtype=torch.float32):
        sizes = [int(x.shape[0]) for x, _ in tasks]
        self.T, self.n = len(tasks), max(sizes)
        d = tasks[0][0].shape[1]
        X = np.zeros((self.T, self.n, d), dtype=np.float64)
        Y = np.zeros((self.T, self.n), dtype=np.float64)
        for t, (x, y) in enumerate(tasks):
            X[t, :sizes[t]] = x
            Y[t, :sizes[t]] = np.asarray(y).reshape(-1)
        self.sizes = np.asarray(sizes)
        self.ragged = bool((self.sizes != self.n).any())
        self.x = torch.from_numpy(X).to(dtype).to(device)
        self.y = torch.from_numpy(Y).to(dtype).to(device)
        self.n_valid = torch.from_numpy(self.sizes.astype(np.int32)).to(device)

    def select(self, idx_tensor):
        out = TaskBatch.__new__(TaskBatch)
        out.T, out.n, out.ragged = int(idx_tensor.numel()), self.n, self.ragged
        out.x, out.y, out.n_valid = L.gather_tasks(self.x, self.y, self.n_valid, idx_tensor)
        out.sizes = None
        return out


class AsyncUploader:
    """Host -> device upload of small per-step arrays (task indices, reparameterisation noise) WITHOUT stalling the host:
    a ring of pinned staging buffers, each guarded by an event, and non-blocking copies on the current stream.  A plain
    `torch.from_numpy(a).to(device)` from pageable memory blocks the host until everything queued before it has run, i.e. it
    is a stream synchronisation every step: the GPU then idles while the host issues the first launches of the next step."""

    def __init__(self, device, dtype, slots=8):
        self.device, self.dtype, self.slots = device, dtype, slots
        self.pool, self.cap, self.events, self.k = None, 0, [None] * slots, 0

    def upload(self, array):
        a = array.contiguous() if torch.is_tensor(array) else torch.from_numpy(np.ascontiguousarray(array))
        if a.numel() > self.cap:                         # (re)allocate the whole ring with ONE pinned allocation
            for ev in self.events:
                if ev is not None:
                    ev.synchronize()
            self.cap = max(a.numel(), 1)
            self.pool = torch.empty(self.slots * self.cap, dtype=self.dtype).pin_memory()
        k = self.k
        self.k = (k + 1) % self.slots
        if self.events[k] is not None:
            self.events[k].synchronize()                 # the copy that last used this slot has executed (normally long ago)
        view = self.pool[k * self.cap:k * self.cap + a.numel()].reshape(a.shape)
        view.copy_(a)                                    # (converts dtype if needed)
        out = torch.empty(a.shape, dtype=self.dtype, device=self.device)
        out.copy_(view, non_blocking=True)
        ev = self.events[k] = self.events[k] or torch.cuda.Event()
        ev.record()
        return out


class StepFeed:
    """Per-step operands of a graph-captured meta-training step (include/pacoh_gp.h, "whole steps as hipGraphs"): the task draws
    of up to `chunk` steps idx_all[chunk, tb], their step scalars sc_all[chunk, SC_COUNT] and an optional per-step payload
    aux_all[chunk, ...] (PACOH-VI: the reparameterisation noise) are uploaded with one copy each from pinned memory; select() --
    the first launch of the captured step -- moves the current row into the fixed buffers idx / sc / aux the kernels read and
    advances the device-side counter."""

    _warned_threads = False

    def __init__(self, device, dtype, tb, chunk=1024, aux_shape=None):
        if not StepFeed._warned_threads:
            StepFeed._warned_threads = True
            from .util import host_cpu_budget
            if torch.get_num_threads() > host_cpu_budget():
                warnings.warn('torch uses %d intra-op CPU threads but this process may only use %d CPUs (affinity / cgroup quota): '
                              'the spinning pool can get the whole process throttled and starve the GPU of launches -- call '
                              'torch.set_num_threads(%d) or set OMP_NUM_THREADS' % (torch.get_num_threads(), host_cpu_budget(),
                                                                                  min(4, host_cpu_budget())))
        chunk = max(int(chunk), GRAPH_STEPS)              # (a several-steps graph reads GRAPH_STEPS consecutive rows)
        self.device, self.dtype, self.tb, self.chunk = device, dtype, int(tb), int(chunk)
        rows = chunk + 1                                  # (the pipelined SVGD step fetches one row ahead: pipeline())
        self.idx_all = torch.zeros(rows, tb, dtype=torch.int64, device=device) if tb > 0 else None
        self.idx = torch.zeros(tb, dtype=torch.int64, device=device) if tb > 0 else None
        self.sc_all = torch.zeros(rows, L.SC_COUNT, dtype=dtype, device=device)
        self.sc = torch.zeros(L.SC_COUNT, dtype=dtype, device=device)
        self.sc2 = self.batch = self.hyp = self._pipe = None
        self.ctr = torch.zeros(1, dtype=torch.int64, device=device)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=device)      # last-block ticket of pacoh_step_begin
        # two pinned staging sets, used alternately: the host prepares and enqueues chunk k+1 while the GPU still runs chunk k
        # (with one set it would have to wait for chunk k's upload, which sits in the stream behind chunk k-1's steps)
        self._h_idx = [torch.zeros(rows, max(tb, 1), dtype=torch.int64).pin_memory() for _ in range(2)]
        self._h_sc = [torch.zeros(rows, L.SC_COUNT, dtype=dtype).pin_memory() for _ in range(2)]
        self.aux_all = self.aux = self._h_aux = None
        if aux_shape is not None:
            self.aux_all = torch.zeros((rows,) + tuple(aux_shape), dtype=dtype, device=device)
            self.aux = torch.zeros(tuple(aux_shape), dtype=dtype, device=device)
            self._h_aux = [torch.zeros((rows,) + tuple(aux_shape), dtype=dtype).pin_memory() for _ in range(2)]
        self._ev, self._slot = [None, None], 0
        # the staging sets are FILLED through numpy views: plain memcpy on the calling thread.  torch's copy_ fans a 64 x 1024 index
        # block out over its intra-op pool (128 spinning workers on the 256-core test hosts, which have a CPU quota of 16: the
        # process was throttled for tens of milliseconds per chunk -- util.host_cpu_budget)
        self._n_idx = [t.numpy() for t in self._h_idx]
        self._n_sc = [t.numpy() for t in self._h_sc]
        self._n_aux = [t.numpy() for t in self._h_aux] if self._h_aux is not None else None

    def upload(self, idx_rows, sc_rows, aux_rows=None):
        """idx_rows: int array [k, tb]; sc_rows: k rows of L.step_scalars(); aux_rows: tensor [k, ...], k tensors [...], a callable
        (j, out_row) filling the pinned staging row of step j in place, 'device' (standard normal rows drawn by the device generator,
        no host copy), or None; resets the counter"""
        k = len(sc_rows)
        assert 0 < k <= self.chunk
        q = self._slot
        self._slot = 1 - q
        if self._ev[q] is not None:
            self._ev[q].synchronize()                    # the copies that last read this staging set have executed
        # Rows k .. GRAPH_STEPS repeat the last real row: the warm-up and capture runs of the several-steps graph read GRAPH_STEPS
        # rows whatever k is (meta_fit's first chunk is ONE step), the pipelined SVGD step one more (its update fetches the row
        # behind the one in flight), and they must see valid task indices and step scalars there -- not stale rows, not the zeros of a
        # fresh buffer (lr = 0 and bias correction 0 give NaN optimizer state)
        kk = max(k, GRAPH_STEPS) + 1
        hs = self._n_sc[q]
        hs[:k] = np.asarray(sc_rows, dtype=np.float64)
        hs[k:kk] = hs[k - 1]
        self.sc_all[:kk].copy_(self._h_sc[q][:kk], non_blocking=True)
        if self.tb > 0:
            hi = self._n_idx[q]
            hi[:k] = np.asarray(idx_rows).reshape(k, self.tb)
            hi[k:kk] = hi[k - 1]
            self.idx_all[:kk].copy_(self._h_idx[q][:kk], non_blocking=True)
        if self.aux_all is not None:
            ha = self._n_aux[q]
            on_device = isinstance(aux_rows, str)
            if on_device:                                # 'device': standard normal rows from the device generator (PACOH-VI, noise='device')
                assert aux_rows == 'device'
                self.aux_all[:kk].normal_()
            elif callable(aux_rows):                     # aux_rows(j, out): writes step j's payload straight into the staging row
                for j in range(k):
                    aux_rows(j, self._h_aux[q][j])
            elif torch.is_tensor(aux_rows):              # (any device / dtype / requires_grad: what copy_ used to accept)
                ha[:k] = aux_rows.detach().to('cpu', self._h_aux[q].dtype).numpy().reshape(ha[:k].shape)
            else:                                        # one tensor per step
                for j, row in enumerate(aux_rows):
                    ha[j] = row.detach().to('cpu', self._h_aux[q].dtype).numpy().reshape(ha[j].shape)
            if not on_device:
                ha[k:kk] = ha[k - 1]
                self.aux_all[:kk].copy_(self._h_aux[q][:kk], non_blocking=True)
        self.ctr.fill_(-1 if self.sc2 is not None else 0)      # (pipelined step: the first forward makes it 0 -- prologue())
        self._ev[q] = self._ev[q] or torch.cuda.Event()
        self._ev[q].record()

    def pipeline(self, tasks, engine, theta):
        """switch the feed to the pipelined SVGD step (csrc/step_tail.h): persistent batch buffers, transformed hyper-parameters and
        two rows of step scalars, filled one step ahead by the update launch.  prologue() after every upload()"""
        assert self.tb > 0
        dev, dt = tasks.x.device, tasks.x.dtype
        batch = TaskBatch.__new__(TaskBatch)
        batch.T, batch.n, batch.ragged, batch.sizes = self.tb, tasks.n, tasks.ragged, None
        batch.x = torch.empty(self.tb, tasks.n, tasks.x.shape[2], dtype=dt, device=dev)
        batch.y = torch.empty(self.tb, tasks.n, dtype=dt, device=dev)
        batch.n_valid = torch.empty(self.tb, dtype=torch.int32, device=dev) if tasks.ragged else None
        off_ls, f, off_os, off_noise, _ = engine._hyper_offsets()
        P = theta.shape[0]
        self.batch = batch
        self.hyp = (torch.empty(P, f, dtype=theta.dtype, device=theta.device),
                    torch.empty(P, dtype=theta.dtype, device=theta.device) if off_os >= 0 else None,
                    torch.empty(P, dtype=theta.dtype, device=theta.device))
        self.hyper = (off_ls, f, off_os, off_noise, engine.noise_floor, engine.layout.kernel_code)
        self.sc2 = torch.zeros(2, L.SC_COUNT, dtype=self.dtype, device=self.device)
        self._row0 = torch.zeros(1, dtype=torch.int64, device=self.device)      # constant: the prologue works on row 0
        self._pipe = (tasks, theta)

    def prologue(self):
        """row 0 of the uploaded chunk into the batch buffers and sc2[0], the hyper-parameters of the particles as they are now: one
        launch, once per chunk -- not part of the step.  (upload() has left the counter at -1: the forward of the first step makes it 0)"""
        tasks, theta = self._pipe
        sc, ctr, self.sc, self.ctr = self.sc, self.ctr, self.sc2[0], self._row0     # (step_begin reads its row number from feed.ctr)
        try:
            L.step_begin(self, tasks, (self.batch.x, self.batch.y, self.batch.n_valid), theta, self.hyper, self.hyp, advance=False)
        finally:
            self.sc, self.ctr = sc, ctr

    def select(self):
        L.step_select(self.idx_all, self.sc_all, self.ctr, self.idx, self.sc, self.aux_all, self.aux)

    def begin(self, tasks, engine=None, theta=None, advance=True, svgd=None):
        """select() + the step's task gather (+ the hyper-parameter transforms of theta through `engine`, + the SVGD distance
        matrix of svgd = (particles, workspace)) in ONE launch -> (TaskBatch | None, hypers | None); advance=False: the step's
        last launch advances the counter (L.step_begin)"""
        batch = hyp = None
        out = None
        if self.tb > 0:
            batch = TaskBatch.__new__(TaskBatch)
            batch.T, batch.n, batch.ragged, batch.sizes = self.tb, tasks.n, tasks.ragged, None
            batch.x = torch.empty(self.tb, tasks.n, tasks.x.shape[2], dtype=tasks.x.dtype, device=tasks.x.device)
            batch.y = torch.empty(self.tb, tasks.n, dtype=tasks.x.dtype, device=tasks.x.device)
            batch.n_valid = torch.empty(self.tb, dtype=torch.int32, device=tasks.x.device) if tasks.ragged else None
            out = (batch.x, batch.y, batch.n_valid)
        hyper = hyper_out = None
        if theta is not None and self.tb > 0:
            off_ls, f, off_os, off_noise, _ = engine._hyper_offsets()
            P = theta.shape[0]
            ls = torch.empty(P, f, dtype=theta.dtype, device=theta.device)
            os_ = torch.empty(P, dtype=theta.dtype, device=theta.device) if off_os >= 0 else None
            noise = torch.empty(P, dtype=theta.dtype, device=theta.device)
            hyper, hyper_out = (off_ls, f, off_os, off_noise, engine.noise_floor, engine.layout.kernel_code), (ls, os_, noise)
            hyp = hyper_out
        L.step_begin(self, tasks, out, theta if hyp is not None else None, hyper, hyper_out, advance=advance, svgd=svgd)
        return batch, hyp


def _begin_vi(self, tasks, engine, posterior, S):
    """StepFeed.begin for a PACOH-VI step with a diagonal posterior: select (incl. the step's noise) + task gather + the step's
    samples, their log q and transformed hyper-parameters in ONE launch -> (TaskBatch | None, hypers, theta[S, D], log_q[S])"""
    D = posterior.shape[1]
    dev, dt = posterior.device, posterior.dtype
    batch = out = None
    if self.tb > 0:
        batch = TaskBatch.__new__(TaskBatch)
        batch.T, batch.n, batch.ragged, batch.sizes = self.tb, tasks.n, tasks.ragged, None
        batch.x = torch.empty(self.tb, tasks.n, tasks.x.shape[2], dtype=tasks.x.dtype, device=tasks.x.device)
        batch.y = torch.empty(self.tb, tasks.n, dtype=tasks.x.dtype, device=tasks.x.device)
        batch.n_valid = torch.empty(self.tb, dtype=torch.int32, device=tasks.x.device) if tasks.ragged else None
        out = (batch.x, batch.y, batch.n_valid)
    off_ls, f, off_os, off_noise, _ = engine._hyper_offsets()
    hyp = (torch.empty(S, f, dtype=dt, device=dev), torch.empty(S, dtype=dt, device=dev) if off_os >= 0 else None,
           torch.empty(S, dtype=dt, device=dev))
    theta, log_q = torch.empty(S, D, dtype=dt, device=dev), torch.empty(S, dtype=dt, device=dev)
    L.step_begin_vi(self, tasks, out, posterior, theta, log_q, (off_ls, f, off_os, off_noise, engine.noise_floor, engine.layout.kernel_code),
                    hyp, advance=False)
    return batch, hyp, theta, log_q


StepFeed.begin_vi = _begin_vi


GRAPH_STEPS = 4      # steps per graph of the second graph the learners capture at world size 1 (hipGraphLaunch costs the host per
                     # launch, not per node: 0.006-0.012 ms per step with one step per graph, 0.003 with four -- what keeps the GPU fed
                     # when the host is busy with somebody else's job; tools/graph_steps_probe.py)


FIRST_CHUNK = 16     # steps in the first chunk of a call that has many more to issue (first_chunk)


def first_chunk(n_steps, chunk):
    """size of the FIRST chunk of a training call: the host prepares a chunk (task draws, step scalars, PACOH-VI's noise: 0.15 ms
    per step) before it can issue any of its steps, and while it prepares chunk k + 1 the GPU runs chunk k -- except for the first
    one, in front of which the GPU idles (meta_fit synchronises at every log line).  A small first chunk gets it started: 1.4 ms
    (SVGD, 200 steps) / 19 ms (VI, 128 steps) of idle time become 0.1 / 2.4 ms at the price of one more upload"""
    k = min(n_steps, chunk)
    if n_steps >= 4 * FIRST_CHUNK and k > FIRST_CHUNK:
        return FIRST_CHUNK
    return k


def replay_steps(n, graph_one, graph_many):
    """n steps through graph_many (GRAPH_STEPS steps per replay) and graph_one (the remainder)"""
    if graph_many is not None:
        for _ in range(n // GRAPH_STEPS):
            graph_many.replay()
        n -= (n // GRAPH_STEPS) * GRAPH_STEPS
    for _ in range(n):
        graph_one.replay()


def capture_graph(body, warmup=2, before=None):
    """hipGraph of body(): warm-up runs on a side stream first (workspaces get allocated outside the graph's pool), then the
    capture; before() (not captured) runs in front of every one of these runs -- the learners rewind their feed's step counter
    with it, so that no run reads past the rows the feed holds; the caller restores whatever state the runs changed"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            if before is not None:
                before()
            body()
    torch.cuda.current_stream().wait_stream(side)
    if before is not None:
        before()
    graph = torch.cuda.CUDAGraph()
    # several ranks: torch.distributed's watchdog thread polls events while this thread captures; with the default (global) capture
    # mode that is an error raised into the capture
    kw = {'capture_error_mode': 'thread_local'} if parallel.world()[1] > 1 else {}
    with gc_paused():                                     # (no garbage collection inside the capture: util.gc_paused)
        with torch.cuda.graph(graph, **kw):
            body()
    return graph


def build_step_graphs(body_likelihood, exchange, body_update, feed, many_ok=True):
    """the hipGraphs of one meta-training step = body_likelihood -> exchange (the all-reduce of the packed buffer) -> body_update:
    ((whole step,), four steps) when the exchange can be captured (world size 1, or RCCL on the compute stream:
    parallel.collective_in_graph()), else ((likelihood, update), None) around the eager torch.distributed call"""
    def rewind():
        if feed.sc2 is not None:
            feed.ctr.fill_(-1)                            # pipelined SVGD step: row 0 fetched again, counter = -1
            feed.prologue()
        else:
            feed.ctr.zero_()
    if parallel.collective_in_graph():
        def whole():
            body_likelihood()
            exchange()
            body_update()

        def several():
            for _ in range(GRAPH_STEPS):
                whole()
        return (capture_graph(whole, before=rewind),), (capture_graph(several, before=rewind) if many_ok else None)
    return (capture_graph(body_likelihood, before=rewind), capture_graph(body_update, before=rewind)), None


def run_step(graphs, graphed, body_likelihood, exchange, body_update):
    """one step: replayed from build_step_graphs()'s graphs, or the same launches issued one by one"""
    if graphed:
        graphs[0].replay()
        if len(graphs) > 1:
            exchange()
            graphs[1].replay()
    else:
        with L.roctx_range('likelihood'):                  # (no-ops unless PACOH_ROCTX=1)
            body_likelihood()
        with L.roctx_range('exchange'):
            exchange()
        with L.roctx_range('update'):
            body_update()


class StepMode:
    """Replay the captured step graph(s) or issue the same launches one by one?  Both produce identical bits.  The graph is the
    default and nearly always stays: it costs the host 0.003-0.01 ms per step (four steps per replay) against 0.1-0.2 ms for plain
    launches, so that a busy host cannot starve the GPU -- on the shared test hosts plain launches measured anywhere between 1 %
    faster than the replay (quiet host: the replay leaves a slightly larger gap between its kernel nodes on this ROCm) and 30 %
    slower (a neighbour's job on the same cores: 0.566 against 0.427 ms per cfg #3 step).  Plain launches are therefore chosen only
    when they beat the replay by 10 % in both looks' samples, which means that the replay itself is in trouble (a pathological
    graph launch).  The looks time the real training steps: a first one after PROBE steps of each kind (the first steps of a process
    are not representative: cold host, clocks ramping), a second, longer one once 4 * REPROBE more steps have run.
    PACOH_GRAPH=1 / 0 forces a mode."""
    PROBE = 6
    REPROBE = 12
    MARGIN = 0.90        # plain launches must beat the replay by 10 % (see above)

    def __init__(self):
        self.use_graph = None
        forced = os.environ.get('PACOH_GRAPH', '')
        self.forced = forced in ('0', '1')
        if self.forced:
            self.use_graph = forced == '1'
        self.timings = None
        self._since = 0
        self._looks = 0

    @staticmethod
    def _time(step, graphed, n):
        step(graphed)                                      # (not timed: first-use effects)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n - 1):
            step(graphed)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (n - 1)

    def run(self, n_steps, step, replay_many=None):
        """issue n_steps calls of step(graphed); some of them are timed to decide the mode (see the class comment).  replay_many(n)
        (optional) replays n steps with as few graph launches as the learner has graphs for (several steps per graph)"""
        done = 0
        if not self.forced:
            n = 0
            if self._looks == 0:
                n = self.PROBE
            elif self._looks == 1 and self._since >= 4 * self.REPROBE:
                n = self.REPROBE
            # the replay is timed the way it will be used: several steps per graph launch where the learner has such a graph (a
            # single-step replay pays the launch of the graph once per step -- on a slow host that alone decided for plain launches)
            ng = n if replay_many is None else -(-n // GRAPH_STEPS) * GRAPH_STEPS
            cost = 2 * (n + ng + (GRAPH_STEPS if replay_many is not None else 0))
            if n and n_steps >= max(cost, 5 * n):
                def graph_time():
                    if replay_many is None:
                        return self._time(step, True, ng)
                    replay_many(GRAPH_STEPS)               # (not timed: first-use effects)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    replay_many(ng)
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / ng
                # (each kind twice, interleaved, the faster sample counts: one hiccup of the host or the clocks during a handful of
                #  steps must not decide the mode for the rest of the run)
                t = [min(pair) for pair in zip(*[[self._time(step, False, n), graph_time()] for _ in range(2)])]
                self.timings = {'eager_ms': t[0] * 1e3, 'graph_ms': t[1] * 1e3, 'steps_each': n}
                self.use_graph = not (t[0] < self.MARGIN * t[1])
                # one decision for all ranks: with the collective captured in the step graph a rank that replays while a peer issues
                # eagerly still matches call for call, but the run would no longer be reproducible per rank -- and with
                # torch.distributed between two graphs the two modes must not mix at all.  Graph mode wins unless every rank prefers
                # plain launches.
                from . import parallel
                if parallel.world()[1] > 1:
                    self.use_graph = not parallel.agree_all(not self.use_graph)
                self._looks += 1
                done = cost
        graphed = True if self.use_graph is None else self.use_graph
        if graphed and replay_many is not None:
            replay_many(n_steps - done)
        else:
            for _ in range(n_steps - done):
                step(graphed)
        self._since += n_steps


class NotPSDError(RuntimeError):
    """a task's kernel matrix was not positive definite even with the jitter ladder -- what gpytorch's psd_safe_cholesky raises
    (gpytorch.utils.errors.NotPSDError) inside the reference's loss evaluation"""


class GPEngine:
    """Sequences the kernels of one LML(+grad) evaluation for all (task, particle) pairs."""

    def __init__(self, layout, noise_floor=0.0):
        self.layout = layout
        self.noise_floor = float(noise_floor)
        self._ws = {}

    # -- pieces -----------------------------------------------------------------------------------
    def _hyper_offsets(self):
        lay = self.layout
        off_os = lay.slices['outputscale_raw'][0] if lay.with_outputscale else -1
        off_c = lay.slices['constant_mean'][0] if lay.mean_module == 'constant' else -1
        return lay.slices['lengthscale_raw'][0], lay.feature_dim, off_os, lay.slices['noise_raw'][0], off_c

    def _hypers(self, theta):
        off_ls, f, off_os, off_noise, _ = self._hyper_offsets()
        return L.hyper_fwd(theta, off_ls, f, off_os, off_noise, self.noise_floor, kernel=self.layout.kernel_code)

    def _paired_nets(self):
        """both networks present with the same hidden shape -> (mean block offset, kernel block offset), else None"""
        lay = self.layout
        if lay.mean_module == 'NN' and lay.covar_module == 'NN' and lay.mean_nn_layers == lay.kernel_nn_layers:
            return lay.block_range('mean_nn.')[0], lay.block_range('kernel_nn.')[0]
        return None

    def _features(self, theta, x, T, n, theta_per_task=False, keep=False, svgd_tail=None):
        """kernel inputs z (+ divisor) and mean (+ mode) for B = T*P problems (b = t*P + p: task t, parameter row p).
        theta_per_task: theta holds T*S rows, S of its own per task -- the same kernels with P = T*S parameter rows, ONE
        problem per row (B = T*S) and inputs shared by S consecutive problems.  keep: park the activations the backward of the
        SAME step needs in self._ws['stash'] (lml_and_grad)"""
        lay = self.layout
        P, D = theta.shape
        B, x_div = (P, P // T) if theta_per_task else (T * P, P)
        pair = self._paired_nets()
        if pair is not None:                               # mean + kernel-feature network in one call (one launch on the fused path)
            stash = None
            if keep:
                key = ('stash', B, n)                      # one stash per batch shape: a captured graph keeps using its own
                stash = self._ws[key] = L.mlp2_stash(x, P, lay.input_dim, list(lay.mean_nn_layers), 1, lay.feature_dim, B, n,
                                                     self._ws.get(key))
            mean, z = L.mlp2_fwd(x, x_div, theta, P, lay.input_dim, list(lay.mean_nn_layers), pair[0], 1, pair[1],
                                 lay.feature_dim, B, n, ws_holder=self._ws, stash=stash,
                                 svgd_tail=svgd_tail[:3] if svgd_tail is not None else None)
            return z, 1, mean.reshape(B, n), L.MEAN_VECTOR
        if svgd_tail is not None:
            L.svgd_dist_advance(*svgd_tail[:3])            # (no paired forward launch to ride in)
        one_net = (lay.covar_module == 'NN') != (lay.mean_module == 'NN')      # the backward of ONE network can read a stash too
        if lay.covar_module == 'NN':
            lo, _ = lay.block_range('kernel_nn.')
            stash = None
            if keep and one_net:
                stash = self._ws[('stash1', B, n)] = L.mlp_stash(x, P, lay.input_dim, list(lay.kernel_nn_layers), lay.feature_dim, B, n,
                                                                 self._ws.get(('stash1', B, n)))
            z = L.mlp_fwd(x, x_div, theta[:, lo:], D, P, lay.input_dim, list(lay.kernel_nn_layers), lay.feature_dim, B, n,
                          ws_holder=self._ws, stash=stash)
            z_div = 1
        else:
            z, z_div = x, x_div
        if lay.mean_module == 'NN':
            lo, _ = lay.block_range('mean_nn.')
            stash = None
            if keep and one_net:
                stash = self._ws[('stash1', B, n)] = L.mlp_stash(x, P, lay.input_dim, list(lay.mean_nn_layers), 1, B, n,
                                                                 self._ws.get(('stash1', B, n)))
            mean = L.mlp_fwd(x, x_div, theta[:, lo:], D, P, lay.input_dim, list(lay.mean_nn_layers), 1, B, n,
                             ws_holder=self._ws, stash=stash).reshape(B, n)
            mode = L.MEAN_VECTOR
        elif lay.mean_module == 'constant':
            lo, hi = lay.slices['constant_mean']
            mean, mode = theta[:, lo:hi].contiguous().reshape(-1), L.MEAN_CONST
        else:
            mean, mode = None, L.MEAN_ZERO
        return z, z_div, mean, mode

    # -- public -----------------------------------------------------------------------------------
    def lml(self, theta, batch):
        """per-datapoint LML of every (task, particle): [T, P] (no gradients)"""
        P = theta.shape[0]
        T, n = batch.T, batch.n
        ls, os_, noise = self._hypers(theta)
        z, z_div, mean, mode = self._features(theta, batch.x, T, n)
        lml, _, _, info = L.gp_lml_fwd(z, z_div, mean, mode, batch.y, P, ls, os_, noise, T * P, P,
                                       n_valid=batch.n_valid if batch.ragged else None, kernel=self.layout.kernel_code)
        return lml.reshape(T, P), info

    def lml_and_grad(self, theta, batch, weight=1.0, lik_out=None, lik_scale=1.0, grad_out=None, fail_flag=None, hypers=None,
                     svgd_tail=None, opt=None):
        """returns (lml[T,P], grad[P,D]) with grad = d(weight * sum_t lml[t,p]) / d theta[p];
        lik_out[P] (optional) receives lik_scale * sum_t lml[t,p] from the same launch that reduces the hyper-gradients;
        grad_out[P,D] (optional, contiguous) is used for the gradient instead of a fresh tensor;
        fail_flag (optional int32[1]) is raised by that launch if any problem's Cholesky failed even with jitter;
        svgd_tail = (particles, workspace, counter, want_bandwidth): the pipelined SVGD step's distance matrix and counter
        increment, in extra workgroups of the forward launch where there is one (L.mlp2_fwd), and -- want_bandwidth -- its median
        bandwidth by one more workgroup of the hyper-parameter reduction (L.hyper_bwd);
        opt (L.adam_inline(...), one parameter row): the AdamW step is applied to every gradient entry where it is finished"""
        lay = self.layout
        P, D = theta.shape
        T, n = batch.T, batch.n
        B = T * P
        dev, dt = theta.device, theta.dtype
        ls, os_, noise = hypers if hypers is not None else self._hypers(theta)      # (hypers: already transformed by pacoh_step_begin)
        z, z_div, mean, mode = self._features(theta, batch.x, T, n, keep=True, svgd_tail=svgd_tail)
        g = None                                        # weight 1: the kernels take g_lml = NULL
        if float(weight) != 1.0:
            gkey = ('g', B, dt, dev)                    # ONE upstream-gradient vector per batch shape, refilled when the weight changes
            ent = self._ws.get(gkey)
            if ent is None or ent[1] != float(weight):
                g = ent[0] if ent is not None else torch.empty(B, dtype=dt, device=dev)
                g.fill_(float(weight))
                ent = self._ws[gkey] = (g, float(weight))
            g = ent[0]
        lml, d_z, d_mean, d_ls, d_os, d_noise, info = L.gp_lml_fwdbwd(
            z, z_div, mean, mode, batch.y, P, ls, os_, noise, B, P,
            n_valid=batch.n_valid if batch.ragged else None, g_lml=g, want_dz=(lay.covar_module == 'NN'), kernel=lay.kernel_code)
        grad = grad_out if grad_out is not None else torch.empty(P, D, dtype=dt, device=dev)   # every block is written below
        pair = self._paired_nets()
        off_ls, f, off_os, off_noise, off_c = self._hyper_offsets()
        hyper = dict(lml=lml if lik_out is not None else None, lik=lik_out, lik_scale=lik_scale,
                     info=info if fail_flag is not None else None, fail_flag=fail_flag)
        if svgd_tail is not None and svgd_tail[3]:
            hyper['svgd_bw'] = (svgd_tail[1],) + tuple(svgd_tail[0].shape)
        if opt is not None:                                 # PACOH-MAP at world size 1: the AdamW step rides in the gradient epilogue
            hyper['opt'] = opt
        if pair is not None:
            # both networks' backward + the hyper-parameter reduction (softplus chain rule, likelihood sums, failure flag): one call,
            # on the fused path two launches
            self._ws['mk'] = L.mlp2_bwd_hyper(batch.x, P, theta, P, lay.input_dim, list(lay.mean_nn_layers), pair[0], 1,
                                              d_mean.reshape(B, n, 1), pair[1], lay.feature_dim, d_z, grad, B, n, T, off_ls, f, off_os,
                                              off_noise, off_c, d_ls, d_os, d_noise, None, workspace=self._ws.get('mk'),
                                              stash=self._ws.get(('stash', B, n)), **hyper)
            return lml.reshape(T, P), grad, info
        d_const = d_mean if lay.mean_module == 'constant' else None
        if (lay.covar_module == 'NN') != (lay.mean_module == 'NN'):
            # ONE network: its backward and the hyper-parameter reduction in one call (one launch less on the fused path)
            if lay.covar_module == 'NN':
                (lo, _), layers, d_out, g_net = lay.block_range('kernel_nn.'), lay.kernel_nn_layers, lay.feature_dim, d_z
            else:
                (lo, _), layers, d_out, g_net = lay.block_range('mean_nn.'), lay.mean_nn_layers, 1, d_mean.reshape(B, n, 1)
            self._ws['k1'] = L.mlp_bwd_hyper(batch.x, P, theta, lo, P, lay.input_dim, list(layers), d_out, g_net, grad, B, n, T, off_ls, f,
                                             off_os, off_noise, off_c, d_ls, d_os, d_noise, d_const, kernel=lay.kernel_code,
                                             workspace=self._ws.get('k1'), stash=self._ws.get(('stash1', B, n)), **hyper)
            return lml.reshape(T, P), grad, info
        if lay.covar_module == 'NN':
            lo, _ = lay.block_range('kernel_nn.')
            self._ws['k'] = L.mlp_bwd(batch.x, P, theta[:, lo:], D, P, lay.input_dim, list(lay.kernel_nn_layers),
                                      lay.feature_dim, d_z, grad[:, lo:], D, False, B, n, self._ws.get('k'))
        if lay.mean_module == 'NN':
            lo, _ = lay.block_range('mean_nn.')
            self._ws['m'] = L.mlp_bwd(batch.x, P, theta[:, lo:], D, P, lay.input_dim, list(lay.mean_nn_layers), 1,
                                      d_mean.reshape(B, n, 1), grad[:, lo:], D, False, B, n, self._ws.get('m'))
        # hyper-parameters (+ constant mean): sum over tasks and softplus chain rule in one launch
        L.hyper_bwd(theta, T, off_ls, f, off_os, off_noise, off_c, d_ls, d_os, d_noise, d_const, grad, kernel=lay.kernel_code, **hyper)
        return lml.reshape(T, P), grad, info

    def predict(self, theta, ctx_x, ctx_y, tst_x, want_cov=False):
        """exact posterior predictive of ONE task for every particle: mu[P,m], var[P,m], cov[P,m,m]|None
        (normalised space, observation noise included)."""
        return self.predict_tasks(theta, ctx_x.unsqueeze(0), ctx_y.reshape(1, -1), tst_x.unsqueeze(0), want_cov=want_cov)

    def predict_tasks(self, theta, ctx_x, ctx_y, tst_x, want_cov=False, theta_per_task=False):
        """the same for T tasks of equal shape in one pass: ctx_x[T,n,d], ctx_y[T,n], tst_x[T,m,d] ->
        mu[T*P,m], var[T*P,m], cov[T*P,m,m]|None, info[T*P]; problem b = t*P + p.  theta_per_task: theta is [T*P, D], rows
        t*P .. t*P+P-1 belong to task t (fresh posterior samples per task)"""
        rows = theta.shape[0]
        T, n = ctx_x.shape[0], ctx_x.shape[1]
        m = tst_x.shape[1]
        P = rows // T if theta_per_task else rows
        ls, os_, noise = self._hypers(theta)
        zc, zc_div, mc, mode = self._features(theta, ctx_x.contiguous(), T, n, theta_per_task)
        zt, zt_div, mt, _ = self._features(theta, tst_x.contiguous(), T, m, theta_per_task)
        return L.gp_predict(zc, zc_div, mc, mode, ctx_y.contiguous(), P, zt, zt_div, mt, ls, os_, noise, T * P, rows, want_cov=want_cov,
                            kernel=self.layout.kernel_code)
