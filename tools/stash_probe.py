"""VERDICT r3 #4b: does the activation stash survive in the 256 MB Infinity Cache between forward and backward when the step's task
batch is processed in chunks?  One pass = features -> GP LML+grad -> networks' backward over 1024 tasks x 20 particles (cfg #3), as
`chunks` consecutive sub-batches of 1024 / chunks tasks sharing ONE stash buffer (670 MB / chunks); the pass is captured in a
hipGraph and replayed, so that the extra launches cost the GPU only.  Run under rocprofv3 --pmc FETCH_SIZE for the backward's HBM reads.
usage: python tools/stash_probe.py <chunks> [passes=100]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import GPRegressionMetaLearnedSVGD, util  # noqa: E402

torch.set_num_threads(util.host_cpu_budget())
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = 1024
rs = np.random.RandomState(0)
tasks = []
for _ in range(T):
    x = rs.uniform(-3, 3, size=(64, 4))
    tasks.append((x, np.sin(x[:, :1]) + 0.1 * rs.randn(64, 1)))
m = GPRegressionMetaLearnedSVGD(tasks, num_particles=20, task_batch_size=T, random_seed=1)
eng, theta = m.engine, m.particles
per = T // chunks
batches = [m.tasks.select(torch.arange(c * per, (c + 1) * per, device=theta.device)) for c in range(chunks)]
grad = torch.empty_like(theta)
lik = torch.empty(theta.shape[0], dtype=theta.dtype, device=theta.device)


def one_pass():
    for b in batches:
        eng.lml_and_grad(theta, b, weight=1.0, lik_out=lik, lik_scale=1.0, grad_out=grad)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    one_pass()
    one_pass()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    one_pass()
torch.cuda.synchronize()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / passes
print('%d chunk(s) of %d tasks x 20 particles (stash %.0f MB per chunk): %.4f ms per pass of 20480 problems, finite=%s'
      % (chunks, per, 670.0 / chunks, dt * 1e3, bool(torch.isfinite(grad).all())))
