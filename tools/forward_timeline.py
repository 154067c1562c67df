#!/usr/bin/env python3
"""One replayed forward out of a rocprofv3 kernel trace of bench.py: hardware queue, start (us since the stem kernel),
duration and grid of every kernel -- the timeline DESIGN.md section 8 discusses.
usage: forward_timeline.py <..._kernel_trace.csv> <out.csv>"""
import csv
import sys


def short(n):
    for key, name in (('conv3x3_tiled_kernel<16', 'C1 tiled 16x16'), ('conv3x3_tiled_kernel<20', 'C1 tiled 20x4'), ('conv3x3_tiled_kernel<40', 'C1 tiled 40x4'),
                      ('conv3x3s2_tiled', 'C1 tiled stride 2'), ('conv1x1_tiled_kernel<128', 'pointwise'), ('conv1x1_tiled_kernel<64', 'heads'),
                      ('conv_band_kernel<2', 'C3 band 5x5'), ('conv_band_kernel<4', 'C3 band 10x10 s2'), ('conv_band_kernel<7', 'C3 band 10x10'), ('merge_collect', 'K1 merge_collect'), ('halo_kernel<16', 'C1 halo 16x16'), ('halo_kernel<40', 'C1 halo 40x4'), ('conv3x3_kernel', 'C2 split-K'),
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


def summary(dst):
    """Coverage of the forward's span: how long at least one chip-filling kernel (>= 256 workgroups) is running, how long only
    latency-bound kernels (< 256 workgroups) are, and how long nothing is."""
    rows = list(csv.DictReader(open(dst)))
    ev = []
    for r in rows:
        s, d, big = float(r['start_us']), float(r['duration_us']), int(r['workgroups']) >= 256
        ev.append((s, 1, big))
        ev.append((s + d, -1, big))
    ev.sort()
    nb = ns = 0
    t_prev, acc = ev[0][0], {'bulk': 0.0, 'small only': 0.0, 'idle': 0.0}
    for t, d, big in ev:
        acc['bulk' if nb else ('small only' if ns else 'idle')] += t - t_prev
        t_prev = t
        if big:
            nb += d
        else:
            ns += d
    span = ev[-1][0] - ev[0][0]
    print(f'forward span {span:.0f} us: ' + ', '.join(f'{k} {v:.0f} us ({100 * v / span:.0f} %)' for k, v in acc.items()))
    by = {}
    for r in rows:
        k = by.setdefault(r['kernel'], [0, 0.0])
        k[0] += 1
        k[1] += float(r['duration_us'])
    for name, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f'  {name:28s} x{n:3d}  {t:8.1f} us in all, {t / n:7.1f} us each')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
    summary(sys.argv[2])
