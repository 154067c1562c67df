#!/usr/bin/env python3
"""Counter summary of the production decoder kernels (K1-fused: band_topk_kernel<...,true> + merge_collect_kernel<...,true>) from the
rocprofv3 --pmc passes of tools/r06_run1.sh: mean per launch of every counter and kernel -> <dir>/k1f_pmc_summary.csv (+ .json with
the derived figures bench.py reports: VALU issue fraction, HBM bytes).

Units (MI355X_MICROARCH.md, constants table): SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles summed over waves;
SQ_BUSY_CYCLES is summed over the SEs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE reports
half the bytes of wide coalesced reads on gfx950 (doubled here, as tools/pmc_traffic.py does)."""
import collections
import csv
import glob
import json
import sys

N_SIMD = 256 * 4


def main(out):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + '/k1f_*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'band_topk' not in k and 'merge_collect' not in k:
                continue
            name = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            acc[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    # the first launches of a process include the warm-up on other inputs: all launches of k1_bench.py run the same shape, keep all
    with open(out + '/k1f_pmc_summary.csv', 'w') as f:
        f.write('kernel,counter,mean_per_launch,launches\n')
        for (k, c), v in sorted(acc.items()):
            f.write(f'"{k}",{c},{sum(v) / len(v):.0f},{len(v)}\n')
    mean = {kc: sum(v) / len(v) for kc, v in acc.items()}
    derived = {}
    for kern in sorted({k for k, _ in mean}):
        g = lambda c: mean.get((kern, c))   # noqa: E731
        d = {}
        if g('GRBM_GUI_ACTIVE') and g('SQ_ACTIVE_INST_VALU'):
            simd_cycles = g('GRBM_GUI_ACTIVE') / 8 * N_SIMD          # cycles of the launch x SIMDs of the chip
            d['launch_cycles'] = round(g('GRBM_GUI_ACTIVE') / 8)
            d['valu_issue_frac'] = round(4 * g('SQ_ACTIVE_INST_VALU') / simd_cycles, 4)
            d['valu_insts_per_simd'] = round(g('SQ_INSTS_VALU') / N_SIMD, 1)
            if g('SQ_WAVE_CYCLES'):
                d['wave_occupancy_per_simd'] = round(4 * g('SQ_WAVE_CYCLES') / simd_cycles, 3)
        if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None:
            d['hbm_read_bytes'] = round(2 * 1024 * g('FETCH_SIZE'))
            d['hbm_write_bytes'] = round(1024 * g('WRITE_SIZE'))
        derived[kern] = d
    json.dump({'note': 'tools/r06_run1.sh: rocprofv3 --pmc passes over tools/k1_bench.py --forms fused --bench-inputs (bs8 640x640, '
                       'decoder inputs of bench.py); valu_issue_frac = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)',
               'per_kernel': derived}, open(out + '/k1f_pmc_summary.json', 'w'), indent=1)
    print(open(out + '/k1f_pmc_summary.csv').read())
    print(json.dumps(derived, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
