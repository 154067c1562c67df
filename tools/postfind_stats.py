#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel trace AFTER MIOpen's solver search: rows from the first replayed forward on
(rocprofv3's own --stats table of a bench.py run is dominated by `naive_conv_*` kernels of the search at engine build).
usage: postfind_stats.py <..._kernel_trace.csv> <out.csv>"""
import csv
import sys


def main(src, dst):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    naive = [i for i, r in enumerate(rows) if 'naive_conv' in r['Kernel_Name']]
    rows = rows[(naive[-1] + 1) if naive else 0:]
    agg = {}
    for r in rows:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(r['Kernel_Name'], [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / total, 3), a[2], a[3]])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
