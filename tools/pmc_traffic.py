#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC slots).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -- python3 tools/k1_bench.py --iters 10
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -- python3 tools/k1_bench.py --iters 10
  python tools/pmc_traffic.py <dirF>/*/*_counter_collection.csv <dirW>/*/*_counter_collection.csv out.json

Corrections of MI355X_MICROARCH.md (HBM section): counter values are KiB; on gfx950 FETCH_SIZE reports exactly half the
bytes of a wide coalesced streaming read (16 B per lane) -- doubled here; WRITE_SIZE is exact for 16-B-per-lane stores.
Calibration in this access pattern: bicubic4_kernel writes N*C*H*W*4 bytes, band_topk_kernel reads them once."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            acc[name.split('(')[0].split('<')[0]].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}   # mean per dispatch, bytes


def main(fetch_csv, write_csv, out):
    rd = {k: 2.0 * v for k, v in per_kernel(fetch_csv, 'FETCH_SIZE').items()}   # gfx950: half the bytes are reported
    wr = per_kernel(write_csv, 'WRITE_SIZE')
    names = sorted(set(rd) | set(wr))
    table = {n: {'read': round(rd.get(n, 0.0)), 'write': round(wr.get(n, 0.0))} for n in names}
    three = [n for n in names if n.startswith(('band_topk_kernel', 'merge_collect_kernel'))]   # og_generate_limbs_f32, flags 0
    single = [n for n in names if n.startswith('generate_limbs_kernel')]
    res = {'hbm_bytes_per_launch': round(sum(table[n]['read'] + table[n]['write'] for n in three)),
           'single_launch_hbm_bytes_per_launch': round(sum(table[n]['read'] + table[n]['write'] for n in single)) if single else None,
           'algorithmic_bytes_per_launch': 8 * 27889280,
           'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/k1_bench.py --iters 10 --rotate 3, bs8 '
                   '640x640); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads), '
                   'values are KiB; hbm_bytes_per_launch = band_topk_kernel + merge_collect_kernel '
                   '(og_generate_limbs_f32, flags 0); calibration: bicubic4_kernel write = 222.8 MB expected',
           'per_kernel': table}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:4])
