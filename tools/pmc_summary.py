#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) per kernel -> JSON for profiles/.
Usage: pmc_summary.py <dir with p*/**/**counter_collection.csv> <out.json> [note]"""
import collections
import csv
import glob
import json
import sys


# MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports HALF of the bytes of a wide coalesced read (16 B per lane; 128-byte
# requests tallied at 64 B) -- "double it"; other access patterns are to be calibrated on a known byte count.
#   * proj_gather_max_kernel reads 400 contiguous bytes per half-wave with 16 B per lane.  Calibrated at the one point
#     where nearly everything misses L2 (cfg5, full-length documents of uniformly drawn words: L2 hit rate 0.042): its
#     4.51 M requests x 0.958 misses x 128 B = 553 MB, and at least 0.958 x 493 MB of row lines MUST cross the L2, while
#     FETCH_SIZE reads 278 MB -- exactly the guide's half.  Doubled.
#   * the projection GEMMs' table-row pieces (64 B per row and K chunk) matched their known byte count 1:1 (round 2
#     calibration): not doubled.  WRITE_SIZE is taken as reported.
#   * mf_adam_kernel (the ID-table sweeps) streams p, m, v as one float4 per lane -- the guide's case word for word -- and
#     WRITES exactly the elements it reads: its true read bytes are at least its write bytes, and FETCH_SIZE reads half
#     of WRITE_SIZE on every workload (round 6: cfg5 42.5 vs 83.5 MB, cfg2 14.8 vs 27.9, the B = 8,192 flush form 85.1
#     vs 164.1).  Doubled.
FETCH_X2 = ('r4r::proj_gather_max_kernel', 'r4r::mf_adam_kernel')


def hbm_bytes(kernel, k):
    """rocprofv3 reports KB."""
    x2 = kernel.startswith(FETCH_X2)
    k['fetch_size_correction'] = 2 if x2 else 1
    k['hbm_bytes_per_launch'] = int(((2 if x2 else 1) * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024)


def main(root, out, note=''):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))
    for f in sorted(glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True)):
        for row in csv.DictReader(open(f)):
            kn = row['Kernel_Name'].split('(')[0]
            if not kn.startswith(('r4r::', 'void r4r::')):
                continue
            a = agg[kn.replace('void ', '')][row['Counter_Name']]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
            a[2] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    res = {'note': note, 'kernels': {}}
    for kn, d in agg.items():
        k = {c: round(v / n, 3) for c, (v, n, _) in d.items()}
        durs = [us / n for _, (v, n, us) in d.items()]
        k['avg_duration_us_under_pmc'] = round(sum(durs) / len(durs), 2)
        k['launches_under_pmc'] = max(n for _, (v, n, us) in d.items())      # per counter pass (a one-off flush launch: 1)
        if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
            hbm_bytes(kn, k)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in k and 'GRBM_GUI_ACTIVE' in k and k['GRBM_GUI_ACTIVE']:
            k['mfma_pipe_busy_frac'] = round(k['SQ_VALU_MFMA_BUSY_CYCLES'] / (k['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
        if 'TCC_HIT_sum' in k and 'TCC_MISS_sum' in k:
            k['l2_hit_rate'] = round(k['TCC_HIT_sum'] / (k['TCC_HIT_sum'] + k['TCC_MISS_sum']), 4)
        res['kernels'][kn] = k
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(res['kernels']), 'kernels')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else '')
