"""rocprofv3 --pmc SQ_* counter_collection csv files (several passes) -> per-kernel per-dispatch averages and a few ratios.
usage: python tools/pmc_sq_summary.py <dir searched recursively for *counter_collection.csv> <out.txt>"""
import csv
import glob
import os
import re
import sys

KEEP = ('gp_mfma_kernel', 'mlp_mfma_bwd_small_kernel', 'mlp_mfma_fwd_kernel', 'gram_kernel<float, 2', 'svgd_update_kernel', 'reduce_slab_kernel')


def main(src, out):
    acc = {}
    for path in sorted(glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r'^void ', '', row['Kernel_Name'])
                name = re.sub(r'\(.*$', '', name).replace('pacoh::', '')
                if not any(k in name for k in KEEP):
                    continue
                c = acc.setdefault(name, {}).setdefault(row['Counter_Name'], [0.0, set()])
                c[0] += float(row['Counter_Value'])
                c[1].add((path, row['Dispatch_Id']))
    lines = ['rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (one pass per counter group);',
             'per-dispatch averages, summed over the XCDs/SEs as rocprofv3 reports them', '']
    for name in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', [0.0, {0}])[0] / max(1, len(acc[k].get('SQ_WAVE_CYCLES', [0, {0}])[1]))):
        d = {k: v[0] / max(1, len(v[1])) for k, v in acc[name].items()}
        lines.append(name)
        for k in sorted(d):
            lines.append('    %-28s %16.0f' % (k, d[k]))
        wc, waves = d.get('SQ_WAVE_CYCLES'), d.get('SQ_WAVES')
        if wc and waves:
            lines.append('    -> wave cycles per wave                  %10.0f' % (wc / waves))
            for k, label in (('SQ_INSTS_VALU', 'VALU instructions per wave'), ('SQ_INSTS_LDS', 'LDS instructions per wave'),
                             ('SQ_INSTS_SALU', 'SALU instructions per wave'), ('SQ_INSTS_VALU_MFMA_MOPS_F32', 'MFMA f32 MOPS per wave')):
                if k in d:
                    lines.append('    -> %-38s %10.1f' % (label, d[k] / waves))
            for k, label in (('SQ_ACTIVE_INST_VALU', 'VALU-active share of wave cycles'), ('SQ_WAIT_INST_ANY', 'waiting-for-instruction share'),
                             ('SQ_WAIT_ANY', 'waiting (any counter) share'), ('SQ_ACTIVE_INST_LDS', 'LDS-active share')):
                if k in d:
                    lines.append('    -> %-38s %9.1f %%' % (label, 100.0 * d[k] / wc))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'SQ_BUSY_CYCLES' in d:
            lines.append('    -> MFMA-busy / SQ-busy cycles             %9.1f %%' % (100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES']))
        lines.append('')
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines))
    print('\n'.join(lines[:60]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
