"""per-kernel per-dispatch averages of rocprofv3 --pmc counter_collection csv files (several passes under one directory)
usage: python tools/pmc_kernel.py <dir> <kernel-name substring>"""
import csv
import glob
import os
import sys

acc = {}
for path in sorted(glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if sys.argv[2] not in row['Kernel_Name']:
                continue
            c = acc.setdefault(row['Counter_Name'], [0.0, set()])
            c[0] += float(row['Counter_Value'])
            c[1].add((path, row['Dispatch_Id']))
d = {k: v[0] / max(1, len(v[1])) for k, v in acc.items()}
for k in sorted(d):
    print('%-30s %16.0f' % (k, d[k]))
wc, waves = d.get('SQ_WAVE_CYCLES'), d.get('SQ_WAVES')
if wc and waves:
    print('wave cycles per wave (x4 = shader cycles) %10.0f' % (wc / waves))
    for k in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_INSTS_MFMA', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_SMEM'):
        if k in d:
            print('%-28s per wave %10.1f' % (k, d[k] / waves))
    for k in ('SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_INST_LDS', 'SQ_INST_CYCLES_VMEM'):
        if k in d:
            print('%-28s share of wave cycles %6.1f %%' % (k, 100.0 * d[k] / wc))
if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'SQ_BUSY_CYCLES' in d:
    print('MFMA busy / SQ busy (per-XCD/SE sums) %6.1f %%' % (100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES']))
