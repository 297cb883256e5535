"""rocprofv3 --kernel-trace csv -> per-launch timeline statistics: kernel time, gaps between consecutive kernels, split by the
phase markers of tools/host_issue_time.py (graph replay first, eager launches second).
usage: python tools/trace_gaps.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys

rows = []
for path in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
print('kernels:', len(rows))
# split into bursts separated by idle periods > 2 ms (the synchronisations between the phases)
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 2_000_000:
        bursts.append(cur)
        cur = []
    cur.append(b)
bursts.append(cur)
for bi, bu in enumerate(bursts):
    if len(bu) < 50:
        continue
    busy = sum(e - s for s, e, _ in bu)
    span = bu[-1][1] - bu[0][0]
    gaps = sorted(b[0] - a[1] for a, b in zip(bu, bu[1:]))
    print('burst %d: %d kernels, span %.3f ms, busy %.3f ms (%.1f %%), gap median %.2f us, p90 %.2f us, max %.1f us'
          % (bi, len(bu), span / 1e6, busy / 1e6, 100.0 * busy / span, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
    names = {}
    for s, e, n in bu:
        n = n.split('(')[0][-60:]
        names.setdefault(n, []).append(e - s)
    for n, v in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print('      %-62s x%-5d avg %.2f us' % (n, len(v), sum(v) / len(v) / 1e3))
