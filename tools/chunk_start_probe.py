"""Where does a SHORT timed region lose time?  The driver times `bench.py --steps 20`: 20 steps = 8.5 ms of GPU work behind a
synchronisation.  This probe times the pieces of such a chunk on the host and on the GPU (events), and the same 20 steps again
when the GPU was busy until just before (no idle gap), to separate host preparation, upload + prologue and clock ramp-up.
    python tools/chunk_start_probe.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import meta_learning_pacoh_amd as M  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
model = M.GPRegressionMetaLearnedSVGD(bench.make_tasks(1024, 64, 4), num_particles=20, covar_module='NN', mean_module='NN',
                                      task_batch_size=-1, lr=1e-3, random_seed=0)
os.environ['PACOH_GRAPH'] = '1'
model._step_mode.use_graph, model._step_mode.forced = True, True
model._train_steps(200)
torch.cuda.synchronize()
for idle_ms in (0.0, 0.0, 1.0, 5.0, 20.0, 0.0):
    model._train_steps(64)
    torch.cuda.synchronize()
    if idle_ms:
        time.sleep(idle_ms * 1e-3)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    idx_rows, sc_rows = model._draw_steps(K, model.lr_scheduler, model.opt_step + 1)
    t1 = time.perf_counter()
    e0.record()
    model._feed.upload(idx_rows, sc_rows)
    if model._pipelined:
        model._feed.prologue()
    e1.record()
    t2 = time.perf_counter()
    M.engine.replay_steps(K, model._graphs[0], model._graph_many)
    e2.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    model.opt_step += K
    for _ in range(K):
        model.lr_scheduler.step()
    print('idle %4.1f ms | host: draw %.3f  upload+prologue %.3f  replays %.3f  wait %.3f  total %.3f ms = %.4f ms/step | '
          'GPU: upload+prologue %.3f  steps %.3f ms = %.4f ms/step'
          % (idle_ms, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, (t4 - t0) * 1e3 / K,
             e0.elapsed_time(e1), e1.elapsed_time(e2), e1.elapsed_time(e2) / K))
