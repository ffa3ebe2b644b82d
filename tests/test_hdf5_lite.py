"""reviews4rec_amd.hdf5_lite -- the package's own reader of the reference's HDF5 epoch files
(data_fast.py:31-45, data_scripts/make_quick_data.py:21-44) -- against files written by the REAL h5py / libhdf5
(tests/golden/hdf5/, generator: tests/golden/make_golden_hdf5.py) and, where an interpreter with h5py exists
(this container's /opt/conda/bin/python3.9; not the GPU box), against files it writes on the spot."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR, TINY_DIR

H5 = os.path.join(GOLDEN_DIR, 'hdf5')
H5PY_PYTHON = '/opt/conda/bin/python3.9'


def _have_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    return subprocess.run([H5PY_PYTHON, '-c', 'import h5py'], capture_output=True).returncode == 0


def test_every_fixture_dataset_reads_back_bit_for_bit():
    from reviews4rec_amd import hdf5_lite
    exp = np.load(os.path.join(H5, 'expected.npz'))
    assert len(exp.files) == 28
    for key in exp.files:
        fn, name = key.split('/', 1)
        with hdf5_lite.File(os.path.join(H5, fn)) as f:
            ds = f[name]
            want = exp[key]
            got = ds[:]
            assert ds.shape == want.shape and got.shape == want.shape, key
            assert got.dtype == want.dtype and got.dtype.isnative, key
            assert np.array_equal(got, want), key
            if want.ndim and len(want):
                assert len(ds) == len(want)
                n = len(want)
                for lo, hi in ((0, 1), (1, n), (n // 3, 2 * n // 3 + 1), (n - 1, n), (2, 2), (n, n + 5), (-3, None)):
                    assert np.array_equal(ds[lo:hi], want[lo:hi]), (key, lo, hi)
                assert np.array_equal(ds[n // 2], want[n // 2]) and np.array_equal(ds[-1], want[-1])
                assert np.array_equal(ds[::2], want[::2])
                if want.ndim > 1:
                    assert np.array_equal(ds[1:n, 0], want[1:n, 0]) and np.array_equal(ds[0, 1:], want[0, 1:])


def test_the_quick_data_files_have_the_reference_writers_layout():
    """make_quick_data.py:21-32: a / d / e [n, T] (NARRE: [n, R, W]), b / c [n, 10], f / g / h [n]; i8, h f8; gzip chunks."""
    from reviews4rec_amd import hdf5_lite
    for fn, tail in (('deepconn_train.hdf5', (37,)), ('narre_train.hdf5', (10, 6))):
        with hdf5_lite.File(os.path.join(H5, fn), 'r') as f:
            assert f.keys() == list('abcdefgh') and 'a' in f and 'z' not in f
            n = len(f['a'])
            assert n == 235
            for k in 'ade':
                assert f[k].shape == (n,) + tail
            for k in 'bc':
                assert f[k].shape == (n, 10)
            for k in 'fgh':
                assert f[k].shape == (n,)
            assert all(f[k].dtype == np.int64 for k in 'abcdefg') and f['h'].dtype == np.float64
            assert all(f[k].chunks is not None and f[k]._filters == [(1, [4])] for k in 'abcdefgh')   # deflate, level 4
            with pytest.raises(KeyError):
                f['i']


def test_the_fast_loader_reads_the_reference_files_directly(tmp_path, monkeypatch):
    """data_fast.DataLoader on train.hdf5 -- the file itself, no conversion -- yields the reference loader's batches."""
    from reviews4rec_amd.data_fast import DataLoader, read_split
    monkeypatch.chdir(tmp_path)
    for mt, fn, root in (('deepconn', 'deepconn_train.hdf5', 'quick_data_deepconn'), ('NARRE', 'narre_train.hdf5', 'quick_data_narre')):
        os.makedirs('%s/Tiny/5_core' % root)
        shutil.copy(os.path.join(H5, fn), '%s/Tiny/5_core/train.hdf5' % root)
        hp = {'batch_size': 16, 'data_dir': 'data/Tiny/5_core/', 'model_type': mt}
        fast = DataLoader(hp, 'train.hdf5', device=torch.device('cpu'))
        z = np.load(os.path.join(TINY_DIR, mt + '_streams.npz'))
        assert len(fast) == int(z['len'][0])
        k = -1
        for k, (data, y) in enumerate(fast.iter()):
            for s in range(7):
                assert data[s].dtype == torch.int64 and np.array_equal(data[s].numpy(), z['train/%d/%d' % (k, s)])
            assert y.dtype == torch.float32 and np.array_equal(y.numpy(), z['train/%d/y' % k])
        assert k + 1 == len(fast)
        arrays = read_split('%s/Tiny/5_core/train.hdf5' % root)
        assert arrays['h'].dtype == np.float64 and arrays['a'].dtype == np.int64


def test_converter_tool_writes_the_same_arrays(tmp_path):
    src = str(tmp_path / 'train.hdf5')
    shutil.copy(os.path.join(H5, 'deepconn_train.hdf5'), src)
    tool = os.path.join(os.path.dirname(GOLDEN_DIR), '..', 'tools', 'hdf5_to_npz.py')
    subprocess.run([sys.executable, tool, src], check=True, capture_output=True)
    z, exp = np.load(str(tmp_path / 'train.npz')), np.load(os.path.join(H5, 'expected.npz'))
    for k in 'abcdefgh':
        assert np.array_equal(z[k], exp['deepconn_train.hdf5/' + k]) and z[k].dtype == exp['deepconn_train.hdf5/' + k].dtype


def test_what_is_not_read_raises_instead_of_returning_wrong_data(tmp_path):
    from reviews4rec_amd import hdf5_lite
    p = str(tmp_path / 'x.hdf5')
    open(p, 'wb').write(b'PK\x03\x04' + b'\0' * 4000)                  # an .npz renamed, say
    with pytest.raises(hdf5_lite.Hdf5Error, match='not an HDF5 file'):
        hdf5_lite.File(p)
    open(p, 'wb').close()
    with pytest.raises(hdf5_lite.Hdf5Error, match='not an HDF5 file'):
        hdf5_lite.File(p)
    with pytest.raises(hdf5_lite.Hdf5Error, match='reads only'):
        hdf5_lite.File(os.path.join(H5, 'variants.hdf5'), 'w')
    raw = open(os.path.join(H5, 'deepconn_train.hdf5'), 'rb').read()
    open(p, 'wb').write(raw[:len(raw) // 2])                          # a copy cut short: chunk addresses past the end
    with pytest.raises((hdf5_lite.Hdf5Error, Exception)):
        with hdf5_lite.File(p) as f:
            for k in 'abcdefgh':
                f[k][:]
    # an unknown filter id in the pipeline message: refuse the dataset (patch deflate's id 1 -> 4, szip)
    with hdf5_lite.File(os.path.join(H5, 'deepconn_train.hdf5')) as f:
        pos = [d for t, _, d, _ in f._messages(f._root._load()['f']) if t == 0x0B][0] + 8
    assert raw[pos:pos + 2] == b'\x01\x00'
    open(p, 'wb').write(raw[:pos] + b'\x04\x00' + raw[pos + 2:])
    with hdf5_lite.File(p) as f:
        assert np.array_equal(f['g'][:], np.load(os.path.join(H5, 'expected.npz'))['deepconn_train.hdf5/g'])
        with pytest.raises(hdf5_lite.Hdf5Error, match='szip'):
            f['f']


@pytest.mark.skipif(not _have_h5py(), reason='no interpreter with h5py on this machine')
def test_against_files_h5py_writes_on_the_spot(tmp_path):
    """Random shapes, chunkings and filters, written by h5py now, read here; includes a chunk B-tree three levels
    deep (5,000 chunks) and the libver='latest' refusal."""
    from reviews4rec_amd import hdf5_lite
    script = r'''
import sys, numpy as np, h5py
rng = np.random.default_rng(int(sys.argv[2]))
exp = {}
with h5py.File(sys.argv[1] + '/live.hdf5', 'w') as f:
    for i in range(12):
        nd = int(rng.integers(1, 4))
        shape = tuple(int(x) for x in rng.integers(1, 60, size=nd))
        chunks = tuple(int(rng.integers(1, s + 1)) for s in shape)
        dt = ['i8', 'f8', 'i4', 'f4', 'u2', 'i1'][int(rng.integers(0, 6))]
        a = (rng.standard_normal(shape) * 100).astype(dt)
        kw = [dict(), dict(chunks=chunks), dict(chunks=chunks, compression='gzip'),
              dict(chunks=chunks, compression='gzip', shuffle=True, compression_opts=9),
              dict(chunks=chunks, fletcher32=True)][int(rng.integers(0, 5))]
        f.create_dataset('d%d' % i, data=a, **kw)
        exp['d%d' % i] = a
    a = rng.integers(0, 1 << 40, size=(5000, 3), dtype=np.int64)
    f.create_dataset('deep', data=a, chunks=(1, 3), compression='gzip')
    exp['deep'] = a
np.savez(sys.argv[1] + '/live.npz', **exp)
with h5py.File(sys.argv[1] + '/latest.hdf5', 'w', libver='latest') as f:
    f.create_dataset('a', data=np.arange(10), chunks=(5,), compression='gzip')
'''
    for seed in (1, 2, 3):
        subprocess.run([H5PY_PYTHON, '-c', script, str(tmp_path), str(seed)], check=True, capture_output=True)
        exp = np.load(str(tmp_path / 'live.npz'))
        with hdf5_lite.File(str(tmp_path / 'live.hdf5')) as f:
            assert f.keys() == sorted(exp.files)
            for k in exp.files:
                assert np.array_equal(f[k][:], exp[k]) and f[k][:].dtype == exp[k].dtype, (seed, k)
            assert np.array_equal(f['deep'][1234:4321], exp['deep'][1234:4321])
    with pytest.raises(hdf5_lite.Hdf5Error, match='latest'):
        with hdf5_lite.File(str(tmp_path / 'latest.hdf5')) as f:
            f['a'][:]
