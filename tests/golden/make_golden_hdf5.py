#!/opt/conda/bin/python3.9
"""HDF5 fixtures written by the REAL h5py / libhdf5 (this container only: /opt/conda/bin/python3.9 carries
h5py 3.3.0 on HDF5 1.10.6; the image's main interpreter and the GPU box have none).

What is written under tests/golden/hdf5/ (data only):

  deepconn_train.hdf5   the Tiny dataset's train split in the reference's quick-data layout: the 8 datasets
  narre_train.hdf5      a..h created exactly as data_scripts/make_quick_data.py:21-32 creates them (i8 / f8,
                        maxshape = shape, compression="gzip") and filled the way :34-44 fills them -- one sliced
                        assignment per loader batch, ragged last batch included.  The batches are the reference
                        loader's own (tests/golden/tiny/<mt>_streams.npz, produced by make_golden_tiny.py from
                        the reference's data.py).
  variants.hdf5         what else libhdf5 may hand a reader of such files: a contiguous (uncompressed) dataset,
                        gzip + shuffle, gzip + fletcher32, a dataset of 175 small chunks (a two-level chunk
                        B-tree), a dataset only partly written (absent chunks read as the fill value), a
                        non-zero fill value, f4 / i4 / big-endian types, more than 8 links in the root group
                        (several symbol-table nodes) and a nested group.
  expected.npz          the arrays h5py reads back from those files (the reader's known answers)

Usage:  /opt/conda/bin/python3.9 tests/golden/make_golden_hdf5.py
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'hdf5')
KEYS = 'abcdefgh'


def quick_file(mt, name):
    z = np.load(os.path.join(HERE, 'tiny', '%s_streams.npz' % mt))
    nb = 1 + max(int(k.split('/')[1]) for k in z.files if k.startswith('train/') and k.split('/')[1].isdigit())
    batches = [[z['train/%d/%s' % (b, s)] for s in ('0', '1', '2', '3', '4', '5', '6', 'y')] for b in range(nb)]
    n = sum(len(b[0]) for b in batches)
    shape = [n] + list(batches[0][0].shape[1:])
    path = os.path.join(OUT, name)
    with h5py.File(path, 'w') as f:
        dset = {}
        dset['a'] = f.create_dataset('a', shape, dtype='i8', maxshape=shape, compression='gzip')
        dset['b'] = f.create_dataset('b', [n, 10], dtype='i8', maxshape=[n, 10], compression='gzip')
        dset['c'] = f.create_dataset('c', [n, 10], dtype='i8', maxshape=[n, 10], compression='gzip')
        dset['d'] = f.create_dataset('d', shape, dtype='i8', maxshape=shape, compression='gzip')
        dset['e'] = f.create_dataset('e', shape, dtype='i8', maxshape=shape, compression='gzip')
        dset['f'] = f.create_dataset('f', [n], dtype='i8', maxshape=[n], compression='gzip')
        dset['g'] = f.create_dataset('g', [n], dtype='i8', maxshape=[n], compression='gzip')
        dset['h'] = f.create_dataset('h', [n], dtype='f8', maxshape=[n], compression='gzip')
        at = 0
        for fields in batches:
            bsz = len(fields[0])
            for k, v in zip(KEYS, fields):
                dset[k][at:at + bsz] = v
            at += bsz
    return path


def variants():
    rng = np.random.default_rng(20200725)
    path = os.path.join(OUT, 'variants.hdf5')
    with h5py.File(path, 'w') as f:
        f.create_dataset('contig', data=rng.integers(-2 ** 40, 2 ** 40, size=(13, 7), dtype=np.int64))
        f.create_dataset('shuf', data=rng.integers(0, 50000, size=(40, 33), dtype=np.int64), compression='gzip', shuffle=True)
        f.create_dataset('fletch', data=rng.standard_normal(301), compression='gzip', fletcher32=True)
        f.create_dataset('many', data=rng.integers(0, 1000, size=(700, 5), dtype=np.int64), chunks=(4, 5), compression='gzip')
        d = f.create_dataset('partial', (100, 6), dtype='i8', chunks=(16, 6), compression='gzip')
        d[40:56] = rng.integers(1, 99, size=(16, 6))
        d = f.create_dataset('filled', (9, 4), dtype='i8', chunks=(4, 4), fillvalue=-7)
        d[0:2] = 5
        f.create_dataset('f4', data=rng.standard_normal((6, 3)).astype(np.float32), compression='gzip')
        f.create_dataset('i4', data=rng.integers(-9, 9, size=17, dtype=np.int32))
        f.create_dataset('be', data=rng.integers(0, 2 ** 31, size=(5, 2)).astype('>i8'))
        f.create_dataset('edge', data=rng.integers(0, 9, size=(10, 7, 3), dtype=np.int64), chunks=(4, 4, 2), compression='gzip')
        f.create_dataset('empty', (0, 12), dtype='i8', maxshape=(0, 12), compression='gzip')
        g = f.create_group('grp')
        g.create_dataset('x', data=np.arange(11, dtype=np.float64))
    return path


def main():
    os.makedirs(OUT, exist_ok=True)
    expected = {}
    for mt, name in (('deepconn', 'deepconn_train.hdf5'), ('NARRE', 'narre_train.hdf5')):
        with h5py.File(quick_file(mt, name), 'r') as f:
            for k in KEYS:
                expected['%s/%s' % (name, k)] = f[k][:]
    with h5py.File(variants(), 'r') as f:
        for k in ('contig', 'shuf', 'fletch', 'many', 'partial', 'filled', 'f4', 'i4', 'edge', 'empty', 'grp/x'):
            expected['variants.hdf5/' + k] = f[k][:]
        expected['variants.hdf5/be'] = f['be'][:].astype(np.int64)      # (values; the file keeps them big-endian)
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **expected)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == '__main__':
    main()
