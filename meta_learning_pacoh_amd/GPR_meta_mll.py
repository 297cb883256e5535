"""PACOH-MAP on MI355X: same constructor / meta_fit / predict / state_dict API as the reference's
GPRegressionMetaLearned (meta_learn/GPR_meta_mll.py:12-264); the per-task ExactGP + autograd + AdamW
loop (:104-117) runs as a handful of HIP kernel launches per iteration over the whole task batch."""
import os
import time
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from . import parallel
from .abstract import RegressionModelMetaLearned
from .distributions import GaussianPredictive
from .engine import GPEngine, NotPSDError, ParamLayout, StepFeed, StepMode, TaskBatch, build_step_graphs, replay_steps, run_step
from .modules import apply_initial_values, resolve_covar_module, resolve_mean_module
from .util import StepLR


class GPRegressionMetaLearned(RegressionModelMetaLearned):

    def __init__(self, meta_train_data, learning_mode='both', lr_params=1e-3, weight_decay=0.0, feature_dim=2,
                 num_iter_fit=10000, covar_module='NN', mean_module='NN', mean_nn_layers=(32, 32),
                 kernel_nn_layers=(32, 32), task_batch_size=5, normalize_data=True, optimizer='Adam',
                 lr_decay=1.0, random_seed=None):
        """Arguments as in the reference (GPR_meta_mll.py:14-38)."""
        super().__init__(normalize_data, random_seed)
        assert learning_mode in ['learn_mean', 'learn_kernel', 'both', 'vanilla']
        # strings as the reference, or ZeroMean / ConstantMean / (Scale)RBFKernel objects (modules.py); other objects cannot run here
        mean_module, self._mean_init = resolve_mean_module(mean_module)
        covar_module, self._covar_init, self._learn_outputscale = resolve_covar_module(covar_module)
        assert mean_module in ['NN', 'constant', 'zero'] and covar_module in ['NN', 'SE', 'COS']
        assert optimizer in ['Adam', 'SGD']

        self.lr_params, self.weight_decay, self.feature_dim = lr_params, weight_decay, feature_dim
        self.num_iter_fit, self.task_batch_size, self.normalize_data = num_iter_fit, task_batch_size, normalize_data
        self.optimizer_name, self.learning_mode = optimizer, learning_mode

        meta_train_data = list(meta_train_data)
        self._check_meta_data_shapes(meta_train_data)
        self._compute_normalization_stats(meta_train_data)

        self._setup_gp_prior(mean_module, covar_module, learning_mode, feature_dim, mean_nn_layers, kernel_nn_layers)
        self.engine = GPEngine(self.layout, noise_floor=1e-3)        # GreaterThan(1e-3), GPR_meta_mll.py:54-55

        tasks = [self._prepare_data_per_task(x, y) for x, y in meta_train_data]
        self.tasks = TaskBatch(tasks, self.device, self.dtype)
        self._setup_optimizer(optimizer, lr_params, lr_decay)
        self.fitted = False

    # ------------------------------------------------------------------------------------------
    def _setup_gp_prior(self, mean_module, covar_module, learning_mode, feature_dim, mean_nn_layers, kernel_nn_layers):
        """GPR_meta_mll.py:207-251.  Shared parameters live in ONE flat vector theta[1, D] (layout:
        engine.ParamLayout); RNG consumption order = the reference's: kernel net first, then mean net,
        each torch.nn.Linear in construction order."""
        if covar_module == 'NN':
            assert learning_mode in ['learn_kernel', 'both'], 'neural network parameters must be learned'
        if mean_module == 'NN':
            assert learning_mode in ['learn_mean', 'both'], 'neural network parameters must be learned'
        self.layout = ParamLayout(self.input_dim, mean_module, covar_module, mean_nn_layers, kernel_nn_layers,
                                  feature_dim, with_outputscale=True)
        lay = self.layout
        theta = torch.zeros(lay.D)                     # raw GP hyper-parameters start at 0 (gpytorch default)

        def init_net(prefix, out_dim, layers):
            prev = self.input_dim
            names = ['fc_%i' % (i + 1) for i in range(len(layers))] + ['out']
            for name, size in zip(names, list(layers) + [out_dim]):
                lin = torch.nn.Linear(prev, size)      # same init + RNG stream as models.py:204-207
                lo, hi = lay.slices['%s.%s.bias' % (prefix, name)]
                theta[lo:hi] = lin.bias.detach()
                lo, hi = lay.slices['%s.%s.weight' % (prefix, name)]
                theta[lo:hi] = lin.weight.detach().reshape(-1)
                prev = size

        if covar_module == 'NN':
            init_net('kernel_nn', feature_dim, kernel_nn_layers)
        if mean_module == 'NN':
            init_net('mean_nn', 1, mean_nn_layers)
        apply_initial_values(theta, lay, dict(self._mean_init, **self._covar_init))     # (values carried by module objects)
        self.theta = theta.reshape(1, -1).to(self.dtype).to(self.device)

        # which segments of theta the optimizer updates (learning_mode, GPR_meta_mll.py:244-251)
        segs = []
        if learning_mode in ('learn_kernel', 'both'):
            if covar_module == 'NN':
                segs.append(lay.block_range('kernel_nn.'))
            segs.append(lay.slices['lengthscale_raw'])
            if self._learn_outputscale:                    # (a plain RBFKernel object has no output scale to learn)
                segs.append(lay.slices['outputscale_raw'])
        if learning_mode in ('learn_mean', 'both'):
            if mean_module == 'NN':
                segs.append(lay.block_range('mean_nn.'))
            elif mean_module == 'constant':
                segs.append(lay.slices['constant_mean'])
        segs.append(lay.slices['noise_raw'])           # the likelihood is always trained (:56)
        segs = sorted(segs)
        merged = [list(segs[0])]
        for lo, hi in segs[1:]:
            if lo == merged[-1][1]:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        self.train_segments = [tuple(s) for s in merged]
        self.shared_parameters = self.train_segments

    def _setup_optimizer(self, optimizer, lr, lr_decay):
        """AdamW with weight decay on EVERY group (GPR_meta_mll.py:255) / plain SGD; StepLR(1000, lr_decay)."""
        self.exp_avg = torch.zeros_like(self.theta)
        self.exp_avg_sq = torch.zeros_like(self.theta)
        self.opt_step = 0
        self.lr_scheduler = StepLR(lr, 1000, lr_decay)
        self._feed = self._graphs = None
        self._step_mode = StepMode()

    # ---- one meta-training iteration captured in hipGraph(s) ---------------------------------------------------------------------
    # A MAP iteration is ~10 launches of a few microseconds each, i.e. launch-bound.  The sequence (task gather -> features ->
    # fused GP LML+grad -> MLP backward -> hyper backward -> loss | AdamW) is captured once and replayed; the sampled task indices
    # and the step-dependent Adam scalars of up to GRAPH_CHUNK iterations are uploaded at once (engine.StepFeed).  With several
    # ranks the task batch is sharded (rank r evaluates idx[r::world]) and grad[1,D] + loss -- sums over tasks
    # (GPR_meta_mll.py:109-113) -- are all-reduced between the two graphs of a step.  Same kernels, same order, same results
    # eagerly (PACOH_NO_GRAPH=1).
    GRAPH_CHUNK = 1024

    def _setup_step(self):
        if getattr(self, '_feed', None) is not None:
            return
        D = self.layout.D
        tb_local = len(parallel.shard(np.arange(self.task_batch_size)))
        self._packed = torch.zeros(D + 1, dtype=self.dtype, device=self.device)      # grad[1, D] | loss: ONE all-reduce operand
        self._grad, self._g_loss = self._packed[:D].view(1, D), self._packed[D:]
        self._g_cum = torch.zeros((), dtype=self.dtype, device=self.device)
        self._fail = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._feed = StepFeed(self.device, self.dtype, tb_local, chunk=self.GRAPH_CHUNK)
        self._graphs = self._opt_blk = None
        # The reference's own regime -- a handful of small tasks per iteration -- runs K whole iterations per launch in one
        # workgroup (include/pacoh_gp.h, pacoh_map_persist): world size 1, Adam, a shape the kernel takes.  PACOH_MAP_PERSIST=0: the
        # launch sequence below (tests compare the two)
        self._persist = None
        if (self._adam_advances() and parallel.world()[1] == 1 and tb_local > 0 and os.environ.get('PACOH_MAP_PERSIST', '1') != '0'
                and not L.FORCE_DENSE):
            plan = L.MapPersistPlan(self.layout, self.tasks, tb_local, self.engine.noise_floor, self.train_segments, self.dtype)
            if plan.supported():
                self._persist = plan
        if self._persist is not None:
            self._pipelined = False
            self._task_ws = None
            return
        # Four launches per iteration (forward, GP, backward, slab reduction) where the AdamW step rides in the gradient epilogue
        # (_adam_inline) AND the networks run on the fused kernels: the epilogue then also fetches the next iteration's operands and
        # publishes the updated hyper-parameters' transforms (include/pacoh_gp.h, pacoh_step_next) -- no step_begin launch.
        # PACOH_MAP_PIPELINE=0: keep it
        pipe = os.environ.get('PACOH_MAP_PIPELINE', '1') != '0'
        inline = self._adam_inline() and pipe
        # ... and TWO where the whole task batch's forward + GP + backward is one launch (include/pacoh_gp.h, pacoh_map_task_step: one
        # workgroup per task, BASELINE config #2's regime; round 6: one workgroup for the whole batch with networks up to 128 wide, the
        # reference's PACOH-MAP launcher) in front of the slab reduction.  PACOH_MAP_TASK_FUSED=0: the four launches
        task_ws = plan = None
        if self._adam_inline(task_fused=True) and pipe and tb_local > 0 and os.environ.get('PACOH_MAP_TASK_FUSED', '1') != '0' and not L.FORCE_DENSE:
            plan = L.MapPersistPlan(self.layout, self.tasks, tb_local, self.engine.noise_floor, self.train_segments, self.dtype)
            task_ws = L.map_task_workspace(plan, tb_local, self.device, any_size=os.environ.get('PACOH_MAP_TASK_FUSED') == '1')
        self._pipelined = task_ws is not None or (inline and self._nets_fused(tb_local))
        if self._pipelined:
            self._feed.pipeline(self.tasks, self.engine, self.theta)
        self._task_ws, self._task_plan = (task_ws, plan) if self._pipelined else (None, None)

    def _nets_fused(self, tb_local):
        lay = self.layout
        nets = ([(lay.mean_nn_layers, 1)] if lay.mean_module == 'NN' else []) + \
               ([(lay.kernel_nn_layers, lay.feature_dim)] if lay.covar_module == 'NN' else [])
        return (len(nets) > 0 and tb_local > 0 and (len(nets) == 1 or lay.mean_nn_layers == lay.kernel_nn_layers) and
                all(L.mlp_fused_path(tb_local, 1, self.tasks.n, lay.input_dim, list(h), d_out, self.dtype) for h, d_out in nets))

    def _body_likelihood(self):
        if getattr(self, '_pipelined', False):
            if self._task_ws is not None:
                L.map_task_step(self._task_plan, self.theta, self._feed.batch, self._feed.hyp, self._grad, self._g_loss, -1.0, self._fail,
                                self._task_ws, self._opt_block())
                return
            self.engine.lml_and_grad(self.theta, self._feed.batch, weight=-1.0, lik_out=self._g_loss, lik_scale=-1.0,
                                     grad_out=self._grad, fail_flag=self._fail, hypers=self._feed.hyp, opt=self._opt_block())
            return
        # select + gather + hyper transforms: one launch; with Adam the step's last launch advances the feed's counter
        batch, hyp = self._feed.begin(self.tasks, self.engine, self.theta, advance=not self._adam_advances())
        if batch is None:                                  # more ranks than tasks in the batch: this rank contributes zeros
            self._packed.zero_()
            return
        # loss = -sum_t mll_t rides in the hyper-parameter reduction (lik_out), the gradient of it in grad_out; at world size 1 so does
        # the AdamW step (_adam_inline): every gradient entry is updated by the thread that finishes it
        self.engine.lml_and_grad(self.theta, batch, weight=-1.0, lik_out=self._g_loss, lik_scale=-1.0, grad_out=self._grad,
                                 fail_flag=self._fail, hypers=hyp, opt=self._opt_block() if self._adam_inline() else None)

    def _adam_advances(self):
        return self.optimizer_name == 'Adam' and len(self.train_segments) > 0

    def _adam_inline(self, task_fused=False):
        """the AdamW launch folded into the gradient epilogue (include/pacoh_gp.h, pacoh_adam_inline): no exchange between gradient
        and update, i.e. world size 1, and a task batch on this rank; PACOH_MAP_ADAM_INLINE=0 keeps the separate launch (A/B, tests).
        task_fused: behind pacoh_map_task_step, whose one slab reduction finishes every entry whatever the networks' shapes"""
        lay = self.layout
        one_call = task_fused or not (lay.mean_module == 'NN' and lay.covar_module == 'NN' and lay.mean_nn_layers != lay.kernel_nn_layers)
        # (two networks of different shapes on the general path: two backward calls and a separate reduction -- no single launch
        #  finishes every entry)
        return (self._adam_advances() and parallel.world()[1] == 1 and self._feed.tb > 0 and len(self.train_segments) <= 4 and one_call
                and os.environ.get('PACOH_MAP_ADAM_INLINE', '1') != '0')

    def _opt_block(self):
        blk = getattr(self, '_opt_blk', None)
        if blk is None:
            nxt = L.step_next(self._feed, self.tasks, self.engine.noise_floor) if getattr(self, '_pipelined', False) else None
            blk = self._opt_blk = L.adam_inline(self.theta, self.exp_avg, self.exp_avg_sq, self._feed.sc[L.SC_ADAM:L.SC_ADAM + 4],
                                                self.train_segments, step_counter=self._feed.ctr, loss_cum=self._g_cum.reshape(1),
                                                next_feed=nxt)
        return blk

    def _body_update(self):
        if self._adam_inline(task_fused=getattr(self, '_task_ws', None) is not None):
            return                                        # (done inside _body_likelihood's last launch)
        if not self._adam_advances():
            L.axpy(self._g_cum.reshape(1), self._g_loss, 1.0)
        for k, (lo, hi) in enumerate(self.train_segments):
            if self.optimizer_name == 'Adam':
                last = k == len(self.train_segments) - 1           # (the last launch also advances the feed and sums the loss)
                L.adam_step_dev(self.theta[0, lo:hi], self._grad[0, lo:hi], self.exp_avg[0, lo:hi], self.exp_avg_sq[0, lo:hi],
                                self._feed.sc[L.SC_ADAM:L.SC_ADAM + 4], step_counter=self._feed.ctr if last else None,
                                loss_cum=self._g_cum.reshape(1) if last else None, loss=self._g_loss if last else None)
            else:
                L.axpy(self.theta[0, lo:hi], self._grad[0, lo:hi], -self.lr_scheduler.lr)       # (eager only: host scalar)

    def _all_reduce(self):
        parallel.all_reduce_buffer_(self._packed)         # ONE exchange per iteration: grad [1, D] | loss, in place

    def _build_graphs(self):
        state = (self.theta, self.exp_avg, self.exp_avg_sq, self._feed.ctr, self._fail, self._g_cum)
        saved = [t.clone() for t in state]
        # (the large-context path allocates O(tasks x n^2) scratch per step inside the graph's pool: one step per graph there)
        self._graphs, self._graph_many = build_step_graphs(self._body_likelihood, self._all_reduce, self._body_update, self._feed,
                                                           many_ok=self.tasks.n <= 128)
        for t, sv in zip(state, saved):
            t.copy_(sv)
        if self._pipelined:
            self._feed.prologue()                         # (batch buffers, scalars and hyper-parameters of the restored parameters)

    def _run_step(self, graphed):
        run_step(self._graphs, graphed, self._body_likelihood, self._all_reduce, self._body_update)

    def _use_graph(self):
        # (large contexts run the HBM-resident path, whose launch sequence sets kernel attributes: keep it eager)
        return (self.optimizer_name == 'Adam' and os.environ.get('PACOH_NO_GRAPH', '0') != '1'
                and self.tasks.n <= L.gp_small_max_n(self.dtype, True) and not L.FORCE_DENSE)

    def _train_steps_persist(self, n_steps):
        """K iterations per launch: the task draws and step scalars of a chunk go up in one copy each (engine.StepFeed), one
        launch runs the chunk.  Same draws from rds_numpy, same scalars as the launch sequence."""
        while n_steps > 0:
            k = min(n_steps, self.GRAPH_CHUNK)
            idx = self.rds_numpy.randint(0, self.tasks.T, size=(k, self.task_batch_size))
            sc_rows = L.step_scalar_rows(1.0, self.lr_scheduler.lrs(k), self.opt_step + 1, weight_decay=self.weight_decay)
            self._feed.upload(idx, sc_rows)
            L.map_persist(self._persist, self.theta, self.exp_avg, self.exp_avg_sq, self.tasks, self._feed.idx_all, self._feed.sc_all, k,
                          self._g_loss, self._g_cum.reshape(1), self._fail)
            self.opt_step += k
            for _ in range(k):
                self.lr_scheduler.step()
            n_steps -= k

    def _train_steps(self, n_steps):
        self._setup_step()
        if self._persist is not None:
            return self._train_steps_persist(n_steps)
        graphed = self._use_graph()
        while n_steps > 0:
            k = min(n_steps, self.GRAPH_CHUNK) if self.optimizer_name == 'Adam' else 1     # SGD reads the host-side learning rate
            # rds_numpy.choice(task_dicts, size=B) == randint(0, T, B): with replacement (GPR_meta_mll.py:109); one call of shape
            # [k, B] consumes the numpy stream exactly like k calls
            idx = self.rds_numpy.randint(0, self.tasks.T, size=(k, self.task_batch_size))
            sc_rows = L.step_scalar_rows(1.0, self.lr_scheduler.lrs(k), self.opt_step + 1, weight_decay=self.weight_decay)
            rank, world = parallel.world()
            local = np.ascontiguousarray(idx[:, rank::world])
            parallel.check_same_draws(local, sc_rows)
            self._feed.upload(local if self._feed.tb > 0 else None, sc_rows)
            if self._pipelined:
                self._feed.prologue()
            if self._task_ws is not None:                # (theta may have been set from outside since the last call)
                L.map_task_setup(self._task_plan, self.theta, self._feed.tb, self._task_ws)
            if graphed and self._graphs is None:
                self._build_graphs()
                if self._task_ws is not None:            # (the capture runs stepped the image along with theta; theta was restored)
                    L.map_task_setup(self._task_plan, self.theta, self._feed.tb, self._task_ws)
            if graphed:
                # replay or eager launches, whichever is faster here (engine.StepMode); several steps per replay where possible
                many = (lambda n: replay_steps(n, self._graphs[0], self._graph_many)) if len(self._graphs) == 1 else None
                self._step_mode.run(k, self._run_step, many)
            else:
                for _ in range(k):
                    self._run_step(False)
            self.opt_step += k
            for _ in range(k):
                self.lr_scheduler.step()
            n_steps -= k

    # ------------------------------------------------------------------------------------------
    def meta_fit(self, valid_tuples=None, verbose=True, log_period=500, n_iter=None):
        """GPR_meta_mll.py:82-147: loss = -sum over the sampled tasks of the per-datapoint MLL."""
        assert (valid_tuples is None) or (all([len(valid_tuple) == 4 for valid_tuple in valid_tuples]))
        loss_val = float('nan')
        if len(self.train_segments) > 0:
            t = time.time()
            if n_iter is None:
                n_iter = self.num_iter_fit
            self._setup_step()
            self._g_cum.zero_()
            itr = 0
            while itr < n_iter:
                nxt = 1 if itr == 0 else min(n_iter, (itr // log_period + 1) * log_period)      # up to the next log line
                self._train_steps(nxt - itr)
                itr = nxt
                if itr == 1 or itr % log_period == 0:
                    duration = time.time() - t
                    avg_loss = self._g_cum / (log_period if itr > 1 else 1.0)
                    message = 'Iter %d/%d - Loss: %.6f - Time %.2f sec' % (itr, self.num_iter_fit, avg_loss.item(), duration)
                    self._check_numerics()
                    self._g_cum.zero_()
                    t = time.time()
                    if valid_tuples is not None:
                        valid_ll, valid_rmse, calibr_err = self.eval_datasets(valid_tuples)
                        message += ' - Valid-LL: %.3f - Valid-RMSE: %.3f - Calib-Err %.3f' % (valid_ll, valid_rmse, calibr_err)
                    self._last_log = message
                    if verbose:
                        self.logger.info(message)
            if n_iter > 0:
                loss_val = self._g_loss.item()
            self._check_numerics()
        else:
            self.logger.info('Vanilla mode - nothing to fit')
        self.fitted = True
        return loss_val

    def _check_numerics(self):
        """raise where the reference raises: gpytorch's psd_safe_cholesky -> NotPSDError (read at synchronisation points only)"""
        flag = getattr(self, '_fail', None)
        bad = flag is not None and int(flag.item()) != 0
        if not bad and parallel.world()[1] > 1 and getattr(self, '_g_loss', None) is not None:
            bad = not bool(torch.isfinite(self._g_loss).all())    # another rank's shard failed: its NaN loss came through the all-reduce
        if bad:
            if flag is not None:
                flag.zero_()
            raise NotPSDError('a task kernel matrix was not positive definite even after adding jitter (1e-6 .. 1e-4)')

    def predict(self, context_x, context_y, test_x, return_density=False):
        """GPR_meta_mll.py:149-190 -> (pred_mean, pred_std) numpy, or the predictive distribution."""
        cx, cy, tx = self._prepare_predict(context_x, context_y, test_x)
        mu, var, cov, _ = self.engine.predict(self.theta, cx, cy, tx, want_cov=return_density)
        dist = GaussianPredictive(mu, var, cov, self.y_mean.reshape(-1)[0], self.y_std.reshape(-1)[0], mixture=False)
        if return_density:
            return dist
        return dist.mean.cpu().numpy(), dist.stddev.cpu().numpy()

    def _eval_params(self, **kwargs):
        return (self.theta, False, False) if not kwargs else None

    # ------------------------------------------------------------------------------------------
    def state_dict(self):
        """same dict layout as the reference ({'optimizer', 'model'}, GPR_meta_mll.py:192-205)"""
        model = OrderedDict((name, self.theta[0, lo:hi].detach().cpu().clone()) for name, (lo, hi) in self.layout.slices.items())
        return {'optimizer': {'exp_avg': self.exp_avg.cpu().clone(), 'exp_avg_sq': self.exp_avg_sq.cpu().clone(),
                              'step': self.opt_step, 'epoch': self.lr_scheduler.epoch},
                'model': model}

    def load_state_dict(self, state_dict):
        for name, (lo, hi) in self.layout.slices.items():
            self.theta[0, lo:hi] = state_dict['model'][name].to(self.dtype).to(self.device)
        opt = state_dict['optimizer']
        self.exp_avg.copy_(opt['exp_avg'])
        self.exp_avg_sq.copy_(opt['exp_avg_sq'])
        self.opt_step = int(opt['step'])
        self.lr_scheduler.epoch = int(opt['epoch'])
