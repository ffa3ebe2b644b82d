#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) per kernel -> JSON for profiles/.
Usage: pmc_summary.py <dir with p*/**/**counter_collection.csv> <out.json> [note]"""
import collections
import csv
import glob
import json
import sys


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
        if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
            # rocprofv3 reports KB.  Calibration on this access pattern (64-byte gathered segments):
            # FETCH_SIZE matched the known byte count 1:1, so no x2 correction is applied here.
            k['hbm_bytes_per_launch'] = int((k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in k and 'GRBM_GUI_ACTIVE' in k and k['GRBM_GUI_ACTIVE']:
            k['mfma_pipe_busy_frac'] = round(k['SQ_VALU_MFMA_BUSY_CYCLES'] / (k['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
        if 'TCC_HIT_sum' in k and 'TCC_MISS_sum' in k:
            k['l2_hit_rate'] = round(k['TCC_HIT_sum'] / (k['TCC_HIT_sum'] + k['TCC_MISS_sum']), 4)
        res['kernels'][kn] = k
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(res['kernels']), 'kernels')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else '')
