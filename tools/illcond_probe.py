"""accuracy probe: lml of a few fp32 / fp64 problems against the fp64 oracle, LDS/register-resident and (PACOH_PROBE_DENSE=1) HBM-resident path"""
import sys, os, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_dense_path as t
from meta_learning_pacoh_amd import _lib as L
DEV = 'cuda'
L.FORCE_DENSE = os.environ.get('PACOH_PROBE_DENSE') == '1'
for dt in (torch.float32, torch.float64):
    for case in [(2, 2, 64, 2, True), (2, 2, 32, 2, True), (2, 2, 31, 2, True), (2, 2, 96, 2, True), (2, 2, 200, 3, True)]:
        T, P, n, f, pez = case
        z, mean, y, ls, os_, noise = t.make_problem(T, P, n, f, dt, seed=7 * n + f, per_eval_z=pez, noise_lo=0.0)
        ref = t.oracle_mll(z.double(), mean.double(), y.double(), ls.double(), os_.double(), noise.double(), T, P, pez)
        out = L.gp_lml_fwdbwd(z.to(DEV), 1 if pez else P, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, ls.to(DEV), os_.to(DEV), noise.to(DEV), T * P, P, want_dz=pez)
        print(os.environ.get('PACOH_LIB', 'cur')[-12:], dt, case, 'maxrel lml %.2e' % t.maxrel(out[0], ref))
