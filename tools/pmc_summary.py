"""Turn rocprofv3 --pmc counter_collection csv files (FETCH_SIZE and WRITE_SIZE passes) into profiles/*_pmc_hbm_traffic.json.
usage: python tools/pmc_summary.py <dir with *counter_collection.csv (searched recursively)> <out.json>
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: raw counters are in KB and FETCH_SIZE reports half of a wide coalesced read
stream on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import os
import re
import sys


def main(src, out, passes=0):
    acc = {}
    for path in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r'^void ', '', row['Kernel_Name']).replace('(anonymous namespace)::', '').replace('pacoh::', '')
                name = re.sub(r'\(.*$', '', name)
                d = acc.setdefault(name, {})
                c = d.setdefault(row['Counter_Name'], [0.0, set()])
                c[0] += float(row['Counter_Value'])
                c[1].add((path, row['Dispatch_Id']))
    kernels = {}
    for name, d in acc.items():
        if 'FETCH_SIZE' not in d and 'WRITE_SIZE' not in d:
            continue
        f = d.get('FETCH_SIZE', [0.0, {0}])
        w = d.get('WRITE_SIZE', [0.0, {0}])
        fk, wk = f[0] / max(1, len(f[1])), w[0] / max(1, len(w[1]))
        kernels[name] = {'FETCH_SIZE_KB': round(fk, 2), 'WRITE_SIZE_KB': round(wk, 2), 'launches': max(len(f[1]), len(w[1])),
                         'hbm_bytes_per_launch': int((2 * fk + wk) * 1024)}
    per_pass = {}
    if passes:      # bytes per pass of the profiled program (a kernel launched several times per pass, e.g. the jitter-ladder retries that
                    # exit at once, is summed: what one LML + gradient evaluation of the batch moves)
        for name, d in acc.items():
            if 'FETCH_SIZE' in d or 'WRITE_SIZE' in d:
                per_pass[name] = int((2 * d.get('FETCH_SIZE', [0.0])[0] + d.get('WRITE_SIZE', [0.0])[0]) * 1024 / passes)
    note = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) of `bench.py --steps 3 --warmup 1 '
            '--no-cpu-baseline`, averaged per launch; raw counters are in KB. hbm_bytes applies the gfx950 correction of '
            'MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read stream): hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.')
    # the kernel sources this profile was taken on (bench.py marks its `traffic` figures stale when they have changed since)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from meta_learning_pacoh_amd._build import source_hash
    with open(out, 'w') as fh:
        json.dump({'note': note, 'source_hash': source_hash(), 'kernels': kernels, 'per_pass': per_pass, 'passes': passes}, fh, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:12]:
        print('%-60s %12d B/launch' % (k[:60], v['hbm_bytes_per_launch']))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
