#!/usr/bin/env python3
"""Convert the HDF5 epoch files the reference's data_scripts/make_quick_data.py wrote
(datasets a..h, gzip, make_quick_data.py:21-32) into the .npz files reviews4rec_amd.data_fast
reads.  Needs h5py, i.e. a machine where the reference itself runs; the MI355X image has none.

    python tools/hdf5_to_npz.py quick_data_deepconn/Electronics/5_core/train.hdf5 [...]
"""
import sys

import numpy as np


def convert(path):
    import h5py
    with h5py.File(path, 'r') as f:
        arrays = {k: f[k][:] for k in 'abcdefgh'}
    for k in 'abcdefg':
        arrays[k] = arrays[k].astype(np.int64, copy=False)
    arrays['h'] = arrays['h'].astype(np.float64, copy=False)
    out = path[:-5] + '.npz' if path.endswith('.hdf5') else path + '.npz'
    np.savez(out, **arrays)
    print(path, '->', out, {k: v.shape for k, v in arrays.items()})


if __name__ == '__main__':
    for p in sys.argv[1:]:
        convert(p)
