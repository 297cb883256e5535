"""Where does the task-fused likelihood launch (pacoh_svgd_task_step) stop paying?  PACOH-SVGD steps at n points per task, P particles and
tb tasks per step (B = tb x P problems), forced onto either path (PACOH_SVGD_TASK_FUSED=1 / 0), graph-replayed ms per step:
    python tools/task_fused_crossover.py [n] [P] [layers]        -> the table GPR_meta_svgd.TASK_FUSED_MAX_PROBLEMS is set from"""
import os
import subprocess
import sys

n = sys.argv[1] if len(sys.argv) > 1 else '20'
P = sys.argv[2] if len(sys.argv) > 2 else '10'
layers = sys.argv[3] if len(sys.argv) > 3 else '4'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, time, numpy as np, torch; sys.path.insert(0, %r)\n"
        "import meta_learning_pacoh_amd as M\n"
        "n, P, L, tb = %s, %s, %s, int(sys.argv[1])\n"
        "rs = np.random.RandomState(1)\n"
        "tasks = [(rs.uniform(-3, 3, (n, 1)), rs.normal(size=(n, 1))) for _ in range(max(tb, 20))]\n"
        "m = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=P, task_batch_size=tb, random_seed=1, mean_nn_layers=(32,) * L, kernel_nn_layers=(32,) * L, bandwidth=0.1)\n"
        "m._train_steps(300); torch.cuda.synchronize()\n"
        "t0 = time.perf_counter(); m._train_steps(600); torch.cuda.synchronize()\n"
        "print('RESULT %%d %%s %%.4f' %% (tb * P, m._task_ws is not None, (time.perf_counter() - t0) / 600 * 1e3))\n" % (root, n, P, layers))
print('n = %s points, %s particles, %s x 32 networks: ms per SVGD step (graph replay)' % (n, P, layers))
print('%8s %10s %10s %10s' % ('problems', 'fused', 'general', 'default'))
for tb in (2, 8, 16, 32, 64, 102, 160, 256):
    row = []
    for fused in ('1', '0', None):
        env = dict(os.environ)
        env.pop('PACOH_SVGD_TASK_FUSED', None)
        if fused is not None:
            env['PACOH_SVGD_TASK_FUSED'] = fused
        r = subprocess.run([sys.executable, '-c', code, str(tb)], cwd=root, env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
        row.append(line[0].split() if line else ['RESULT', '?', '?', 'FAILED'])
    print('%8s %10s %10s %10s   (default takes the fused path: %s)' % (row[0][1], row[0][3], row[1][3], row[2][3], row[2][2]), flush=True)
