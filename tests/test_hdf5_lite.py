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


def _sample_arrays(seed=7, n=1000):
    rng = np.random.default_rng(seed)
    return {'a': rng.integers(0, 50000, size=(n, 37)), 'b': rng.integers(0, 9, size=(n, 10)), 'h': rng.random(n) * 5,
            'f4': rng.random((7, 3)).astype(np.float32), 'e0': np.zeros((0, 12), np.int64),
            'u2': rng.integers(0, 60000, size=(33, 2, 3)).astype(np.uint16), 'i1': rng.integers(-9, 9, size=5).astype(np.int8)}


def test_writer_round_trip_through_the_reader(tmp_path):
    """write_file -> File: gzip chunks (one node, several nodes, two levels of nodes) and contiguous storage."""
    from reviews4rec_amd import hdf5_lite
    d = _sample_arrays()
    for kw in (dict(), dict(chunk_bytes=4096), dict(chunk_bytes=296), dict(compression=None)):
        p = str(tmp_path / 'w.hdf5')
        hdf5_lite.write_file(p, d, **kw)
        with hdf5_lite.File(p) as f:
            assert f.keys() == sorted(d)
            for k, want in d.items():
                assert np.array_equal(f[k][:], want) and f[k][:].dtype == want.dtype, (kw, k)
            assert np.array_equal(f['a'][123:777], d['a'][123:777])
            if kw.get('chunk_bytes') == 296:
                assert f['a'].chunks == (1, 37) and len(f['a']._chunk_list(0, 1000)) == 1000      # 1,000 chunks: 16 leaves + a root
    with pytest.raises(hdf5_lite.Hdf5Error):
        hdf5_lite.write_file(str(tmp_path / 'x.hdf5'), {'s': np.array(['a', 'b'])})
    with pytest.raises(hdf5_lite.Hdf5Error):
        hdf5_lite.write_file(str(tmp_path / 'x.hdf5'), {'a/b': np.zeros(3)})
    assert not os.path.exists(str(tmp_path / 'x.hdf5'))


def test_save_split_writes_the_reference_container(tmp_path, monkeypatch):
    from reviews4rec_amd import data_fast, hdf5_lite
    monkeypatch.chdir(tmp_path)
    z = np.load(os.path.join(TINY_DIR, 'deepconn_streams.npz'))
    nb = int(z['len'][0])
    data = [np.concatenate([z['train/%d/%d' % (k, s)] for k in range(nb)]) for s in range(7)]
    y = np.concatenate([z['train/%d/y' % k] for k in range(nb)])
    data_fast.save_split('quick_data_deepconn/Tiny/5_core/train.hdf5', data, y)
    with hdf5_lite.File('quick_data_deepconn/Tiny/5_core/train.hdf5') as f:
        assert f.keys() == list('abcdefgh') and f['h'].dtype == np.float64 and f['a'].dtype == np.int64
        assert f['a']._filters == [(1, [4])]
    exp = np.load(os.path.join(H5, 'expected.npz'))
    got = data_fast.read_split('quick_data_deepconn/Tiny/5_core/train.hdf5')
    for k in 'abcdefgh':                                     # the same arrays the h5py-written fixture holds
        assert np.array_equal(got[k], exp['deepconn_train.hdf5/' + k])
    fast = data_fast.DataLoader({'batch_size': 16, 'data_dir': 'data/Tiny/5_core/', 'model_type': 'deepconn'}, 'train.hdf5',
                                device=torch.device('cpu'))
    for k, (fields, yy) in enumerate(fast.iter()):
        assert all(np.array_equal(fields[s].numpy(), z['train/%d/%d' % (k, s)]) for s in range(7))


@pytest.mark.skipif(not _have_h5py(), reason='no interpreter with h5py on this machine')
def test_h5py_reads_what_the_writer_writes(tmp_path):
    """The other direction of the interchange: the reference's loader (h5py.File(...)[k][:], [a:b]) on our files."""
    from reviews4rec_amd import hdf5_lite
    d = _sample_arrays()
    np.savez(str(tmp_path / 'w.npz'), **d)
    hdf5_lite.write_file(str(tmp_path / 'w0.hdf5'), d)
    hdf5_lite.write_file(str(tmp_path / 'w1.hdf5'), d, chunk_bytes=296)
    hdf5_lite.write_file(str(tmp_path / 'w2.hdf5'), d, compression=None)
    script = r'''
import sys, h5py, numpy as np
exp = np.load(sys.argv[1] + '/w.npz')
for fn in ('w0', 'w1', 'w2'):
    with h5py.File(sys.argv[1] + '/' + fn + '.hdf5', 'r') as f:
        assert sorted(f.keys()) == sorted(exp.files)
        assert len(f['a']) == 1000
        for k in exp.files:
            assert f[k].dtype == exp[k].dtype and f[k].shape == exp[k].shape and f[k].maxshape == exp[k].shape
            assert np.array_equal(f[k][:], exp[k]), (fn, k)
        assert np.array_equal(f['a'][200:400], exp['a'][200:400])
        assert (f['a'].compression == 'gzip' and f['a'].compression_opts == 4) if fn != 'w2' else f['a'].chunks is None
print('ok')
'''
    r = subprocess.run([H5PY_PYTHON, '-c', script, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stderr
    h5ls = os.path.join(os.path.dirname(H5PY_PYTHON), 'h5ls')
    if os.path.exists(h5ls):                                  # libhdf5's own tool walks every structure
        r = subprocess.run([h5ls, '-r', '-v', str(tmp_path / 'w1.hdf5')], capture_output=True, text=True)
        assert r.returncode == 0 and 'Dataset {1000/1000, 37/37}' in r.stdout, r.stdout + r.stderr
