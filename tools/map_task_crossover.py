"""PACOH-MAP iterations at T tasks x 32 points per iteration (BASELINE config #2's modules: SE kernel + NN(32,32) mean) on the task-fused
two-launch iteration (pacoh_map_task_step) and on the four-launch sequence (PACOH_MAP_TASK_FUSED=0): ms per iteration, graph replay
    python tools/map_task_crossover.py"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, time, torch; sys.path.insert(0, %r); import bench\n"
        "import meta_learning_pacoh_amd as M\n"
        "T = int(sys.argv[1])\n"
        "m = M.GPRegressionMetaLearned(bench.sinusoid_tasks(27, T, 32), covar_module='SE', mean_module='NN', task_batch_size=T, random_seed=1)\n"
        "m._train_steps(300); torch.cuda.synchronize()\n"
        "t0 = time.perf_counter(); m._train_steps(600); torch.cuda.synchronize()\n"
        "print('RESULT %%s %%.4f' %% (getattr(m, '_task_ws', None) is not None, (time.perf_counter() - t0) / 600 * 1e3))\n" % root)
print('%8s %10s %10s %10s' % ('tasks', 'fused', '4-launch', 'default'))
for T in (64, 256, 512, 768, 1024, 2048):
    row = []
    for fused in ('1', '0', None):
        env = dict(os.environ, PACOH_MAP_PERSIST='0')
        env.pop('PACOH_MAP_TASK_FUSED', None)
        if fused is not None:
            env['PACOH_MAP_TASK_FUSED'] = fused
        r = subprocess.run([sys.executable, '-c', code, str(T)], cwd=root, env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
        row.append(line[0].split() if line else ['RESULT', '?', 'FAILED'])
    print('%8d %10s %10s %10s   (default takes the fused path: %s)' % (T, row[0][2], row[1][2], row[2][2], row[2][1]), flush=True)
