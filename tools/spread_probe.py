"""Does the hardware pack an under-filled grid onto a few CUs?  Per-kernel times (HIP events, eager pass) of the cfg #3 step at
`tasks` tasks x 20 particles while extra dynamic LDS per workgroup caps how many workgroups a CU takes (PACOH_LDS_PAD_GP /
PACOH_LDS_PAD_MLP, csrc/switches.h).    python tools/spread_probe.py [tasks]"""
import json
import os
import subprocess
import sys

tasks = sys.argv[1] if len(sys.argv) > 1 else '128'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, json, torch; sys.path.insert(0, %r); import bench\n"
        "import meta_learning_pacoh_amd as M; from meta_learning_pacoh_amd import _lib as L\n"
        "bench.TASKS = %s\n"
        "wl = bench.wl_cfg3(1, 'weak', M, L)\n"
        "wl['run'](200); torch.cuda.synchronize()\n"
        "import time; t0 = time.perf_counter(); wl['run'](400); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 400 * 1e3\n"
        "pp = bench.profile_pass(wl, L, 40)\n"
        "print(json.dumps({'ms_per_step': round(ms, 4), 'kernel_ms': pp['kernel_ms']}))\n" % (root, tasks))
print('cfg #3 step at %s tasks x 20 particles: graph-replayed ms per step, per-kernel ms (eager pass)' % tasks)
if len(sys.argv) > 2 and sys.argv[2] == 'tiles':
    # the MLP kernels' tile plan at an under-filled grid ("check the tile-loop quantisation of mlp_fused_*")
    variants = [{}] + [{'PACOH_FUSED_FWD_TPW': str(v)} for v in (4, 8, 16)] + [{'PACOH_FUSED_FWD_PB': '2'}, {'PACOH_FUSED_BWD_PB': '2'},
                                                                              {'PACOH_FUSED_FWD_PB': '2', 'PACOH_FUSED_FWD_TPW': '4'},
                                                                              {'PACOH_FUSED_FWD_PB': '2', 'PACOH_FUSED_FWD_TPW': '8'}]
else:
    variants = [{'PACOH_LDS_PAD_GP': str(gp), 'PACOH_LDS_PAD_MLP': str(mlp)} for gp, mlp in
                ((0, 0), (4096, 0), (8192, 0), (12288, 0), (20480, 0), (0, 8192), (0, 16384), (0, 24576), (0, 40960), (8192, 16384))]
for var in variants:
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, **var), capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    print('%-60s %s' % (' '.join('%s=%s' % kv for kv in var.items()) or '(default)', line[0] if line else 'FAILED ' + r.stderr[-300:]), flush=True)
