import sys, time, torch
sys.path.insert(0, '.')
import bench
import meta_learning_pacoh_amd as M
tasks = bench.make_tasks(1024, 64, 4)
model = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=20, covar_module='NN', mean_module='NN', task_batch_size=-1, lr=1e-3, random_seed=0)
def step():
    idx_local, pre = model._sample_task_batch()
    model.svgd_step(idx_local, pre)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue time per step %.3f ms; total per step %.3f ms' % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200): step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
