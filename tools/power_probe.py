"""Clock and power of the GPU while the cfg #3 step runs back to back (is the step power-limited?): samples rocm-smi beside a
long run of the training loop, idle before / busy during.
    python tools/power_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(2)
import bench  # noqa: E402
import meta_learning_pacoh_amd as M  # noqa: E402


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--showperflevel'], capture_output=True, text=True, timeout=20).stdout
    except Exception as exc:
        return repr(exc)
    keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ('sclk', 'mclk', 'Power', 'Temperature (Sensor junction)', 'Performance Level'))]
    return ' | '.join(keep)


model = M.GPRegressionMetaLearnedSVGD(bench.make_tasks(1024, 64, 4), num_particles=20, covar_module='NN', mean_module='NN',
                                      task_batch_size=-1, lr=1e-3, random_seed=0)
model._train_steps(200)
torch.cuda.synchronize()
time.sleep(1.0)
print('idle :', smi())
stop = False


def run():
    while not stop:
        model._train_steps(1000)
        torch.cuda.synchronize()


th = threading.Thread(target=run)
th.start()
for _ in range(4):
    time.sleep(1.5)
    print('busy :', smi())
stop = True
th.join()
t0 = time.perf_counter()
model._train_steps(1000)
torch.cuda.synchronize()
print('step %.4f ms' % ((time.perf_counter() - t0)))
