#!/usr/bin/env python3
"""One replayed forward out of a rocprofv3 kernel trace of bench.py: hardware queue, start (us since the stem kernel),
duration and grid of every kernel -- the timeline DESIGN.md section 8 discusses.
usage: forward_timeline.py <..._kernel_trace.csv> <out.csv>"""
import csv
import sys


def short(n):
    for key, name in (('halo_kernel<16', 'C1 halo 16x16'), ('halo_kernel<40', 'C1 halo 40x4'), ('conv3x3_kernel', 'C2 split-K'),
                      ('stem7x7', 'stem7x7'), ('upsample2_add', 'upsample2_add'), ('bias_act', 'bias_act'),
                      ('nhwc_slice', 'nhwc_slice_to_nchw'), ('band_topk', 'K1 band_topk'), ('merge_bands', 'K1 merge_bands'),
                      ('collect_limbs', 'K2 collect_limbs'), ('greedy_group', 'K3 greedy_group'), ('bicubic4', 'K1a bicubic4')):
        if key in n:
            return name
    if n.startswith('_ZN2ck') or 'igemm' in n:
        return 'MIOpen conv'
    return n[:40]


def main(src, dst):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    stems = [i for i, r in enumerate(rows) if 'stem7x7' in r['Kernel_Name']]
    fw = rows[stems[-3]:stems[-2]]          # a forward in the middle of the timed region (decoder of the batch before included)
    t0 = int(fw[0]['Start_Timestamp'])
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['queue', 'start_us', 'duration_us', 'kernel', 'workgroups'])
        for r in fw:
            wg = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * max(1, int(r['Grid_Size_Y']) // int(r['Workgroup_Size_Y'])) \
                * max(1, int(r['Grid_Size_Z']) // int(r['Workgroup_Size_Z']))
            w.writerow([r['Queue_Id'], round((int(r['Start_Timestamp']) - t0) / 1e3, 1),
                        round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1), short(r['Kernel_Name']), wg])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
