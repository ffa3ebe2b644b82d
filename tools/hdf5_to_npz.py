#!/usr/bin/env python3
"""Convert the HDF5 epoch files the reference's data_scripts/make_quick_data.py wrote
(datasets a..h, gzip, make_quick_data.py:21-32) into the .npz files reviews4rec_amd.data_fast
reads.  (Not needed to train: data_fast.DataLoader reads the .hdf5 files directly.  A converted .npz loads
faster -- no inflate.)  Uses h5py where it exists, reviews4rec_amd.hdf5_lite otherwise.

    python tools/hdf5_to_npz.py quick_data_deepconn/Electronics/5_core/train.hdf5 [...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def convert(path):
    from reviews4rec_amd.data_fast import read_split
    arrays = read_split(path)
    for k in 'abcdefg':
        arrays[k] = arrays[k].astype(np.int64, copy=False)
    arrays['h'] = arrays['h'].astype(np.float64, copy=False)
    out = path[:-5] + '.npz' if path.endswith('.hdf5') else path + '.npz'
    np.savez(out, **arrays)
    print(path, '->', out, {k: v.shape for k, v in arrays.items()})


if __name__ == '__main__':
    for p in sys.argv[1:]:
        convert(p)
