"""Throughput of the other BASELINE.json configurations (parity-test cases, not the headline bench line)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meta_learning_pacoh_amd as M
from meta_learning_pacoh_amd import _lib as L


def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

out = {}
# cfg 2: PACOH-MAP, 256 tasks, n=32, d=1, SE kernel + NN mean, full batch per step
rs = np.random.RandomState(27)
tasks = [(rs.uniform(-5, 5, (32, 1)), rs.normal(size=(32, 1))) for _ in range(256)]
m = M.GPRegressionMetaLearned(tasks, covar_module='SE', mean_module='NN', task_batch_size=256, random_seed=1)
dt = timeit(lambda: m.meta_fit(n_iter=50, verbose=False, log_period=10**9), reps=4, warm=1) / 50
out['cfg2_map_256x32_d1'] = {'ms_per_iter': dt * 1e3, 'evals_per_s': 256 / dt}
# demo-sized MAP (cfg 1): 5 tasks of 5 points per iteration
tasks5 = [(rs.uniform(-5, 5, (5, 1)), rs.normal(size=(5, 1))) for _ in range(20)]
m1 = M.GPRegressionMetaLearned(tasks5, random_seed=1)
dt = timeit(lambda: m1.meta_fit(n_iter=50, verbose=False, log_period=10**9), reps=3, warm=1) / 50
out['cfg1_map_demo_5x5'] = {'ms_per_iter': dt * 1e3, 'evals_per_s': 5 / dt}
# cfg 4: PACOH-VI, 512 tasks, n=128, S=10, NN mean + kernel
tasks = [(rs.uniform(-5, 5, (128, 1)), rs.normal(size=(128, 1))) for _ in range(512)]
v = M.GPRegressionMetaLearnedVI(tasks, svi_batch_size=10, random_seed=1)
dt = timeit(lambda: v.meta_fit(n_iter=20, verbose=False, log_period=10**9), reps=4, warm=1) / 20
out['cfg4_vi_512x128_S10'] = {'ms_per_iter': dt * 1e3, 'evals_per_s': 5120 / dt}
# cfg 5: large context, 256 tasks, n=512, d=8, fp64: Gram build + dense Cholesky LML
X = torch.randn(256, 512, 8, dtype=torch.float64, device='cuda'); Y = torch.randn(256, 512, dtype=torch.float64, device='cuda')
ls = torch.full((1, 8), 0.6931, dtype=torch.float64, device='cuda'); nz = torch.tensor([0.313], dtype=torch.float64, device='cuda')
def cfg5():
    K = L.gram_rbf_ard(X, 1, X, 1, ls, None, nz, True, 256, 1)
    L.mvn_logprob_dense(K, Y, 1.0 / 512)
dt = timeit(cfg5, reps=5, warm=2)
Kbuf = L.gram_rbf_ard(X, 1, X, 1, ls, None, nz, True, 256, 1)
dt_gram = timeit(lambda: L.gram_rbf_ard(X, 1, X, 1, ls, None, nz, True, 256, 1), reps=10, warm=2)
out['cfg5_dense_256x512_d8_fp64'] = {'ms_per_pass': dt * 1e3, 'evals_per_s': 256 / dt, 'gram_ms': dt_gram * 1e3,
                                     'gram_GBs': 256 * (512 * 8 * 8 + 512 * 512 * 8) / dt_gram / 1e9,
                                     'chol_ms': (dt - dt_gram) * 1e3}
# cfg 5, full LML + gradient through the HBM-resident path (pacoh_gp_lml_dense): gram, Cholesky, triangular inverse, Z^T Z, contractions
os1 = torch.ones(1, dtype=torch.float64, device='cuda')
def cfg5_grad():
    L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, 256, 1)
dt = timeit(cfg5_grad, reps=5, warm=2)
out['cfg5_dense_lml_plus_grad_256x512_d8_fp64'] = {'ms_per_pass': dt * 1e3, 'evals_per_s': 256 / dt}
Xf, Yf, lsf, nzf, osf = X.float(), Y.float(), ls.float(), nz.float(), os1.float()
dt = timeit(lambda: L.gp_lml_fwdbwd(Xf, 1, None, L.MEAN_ZERO, Yf, 1, lsf, osf, nzf, 256, 1), reps=5, warm=2)
out['cfg5_dense_lml_plus_grad_256x512_d8_fp32'] = {'ms_per_pass': dt * 1e3, 'evals_per_s': 256 / dt}
print(json.dumps(out, indent=1))
