"""reviews4rec_amd.data (counterpart of the reference's data.py) against batch streams the
REFERENCE's loader produced on the Tiny dataset (tests/golden/make_golden_tiny.py): every batch of
train / test / val ``iter()`` and of ``iter_negs`` is compared slot by slot, bit for bit.
CPU tests run the host-side evaluation of the pools; the -m gpu tests the r4r_batch_build kernel."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import TINY_DIR, materialise_tiny, tiny_hp

STREAM_MODELS = ['deepconn', 'NARRE', 'MF_dot', 'transnet++']


def _check_streams(mt, tmp_path, device):
    from reviews4rec_amd.data import load_data
    hp = tiny_hp(mt, materialise_tiny(tmp_path))
    train, test, val, hp = load_data(hp, device=device)
    z = np.load(os.path.join(TINY_DIR, mt + '_streams.npz'))
    assert [len(train), len(test), len(val)] == z['len'].tolist()
    review = mt not in ('bias_only', 'MF', 'MF_dot', 'NeuMF')
    for tag, stream in (('train', train.iter()), ('test', test.iter(eval=True)), ('val', val.iter(eval=True)),
                        ('negs', test.iter_negs(review))):
        n = 0
        for k, (data, y) in enumerate(stream):
            for s, d in enumerate(data):
                key = '%s/%d/%d' % (tag, k, s)
                if d is None:
                    assert key not in z.files
                    continue
                assert d.dtype == torch.int64 and d.device.type == torch.device(device).type
                want = z[key]
                assert tuple(d.shape) == want.shape, (key, tuple(d.shape), want.shape)
                assert np.array_equal(d.cpu().numpy(), want), key
            assert y.dtype == torch.float32
            assert np.array_equal(y.cpu().numpy(), z['%s/%d/y' % (tag, k)])
            n += 1
        assert n == int(z[tag + '/n']), tag
    return train, test, val, hp


@pytest.mark.parametrize('mt', STREAM_MODELS)
def test_streams_match_the_reference_loader_host(mt, tmp_path):
    _check_streams(mt, tmp_path, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('mt', STREAM_MODELS)
def test_streams_match_the_reference_loader_device(mt, tmp_path):
    _check_streams(mt, tmp_path, 'cuda')


def test_counts_and_maps_like_the_reference(tmp_path):
    """count_train_counts / calculate_reviewed_map (data.py:36-79) against the dataset's own pickles."""
    from reviews4rec_amd.data import load_data
    from reviews4rec_amd.utils import load_obj
    root = materialise_tiny(tmp_path)
    train, test, val, hp = load_data(tiny_hp('deepconn', root), device='cpu')
    uc, ic = load_obj(root + 'user_count'), load_obj(root + 'item_count')
    assert {int(k): v for k, v in train.user_count.items()} == uc
    assert {int(k): v for k, v in train.item_count.items()} == ic
    assert test.user_count is train.user_count and val.store is train.store
    tiui = load_obj(root + 'this_index_user_item')
    u2i, i2u = train.u_to_i_map, train.i_to_u_map
    for u, per in tiui.items():
        for i, (ku, ki) in per.items():
            assert u2i[u][ku] == i and i2u[i][ki] == u
    assert train.get_count_user(10 ** 6) == 0 and (hp['total_users'], hp['total_items']) == (48, 24)


def test_simple_mode_yields_plain_arrays(tmp_path):
    """iter_review(simple=True) (data.py:282-291) feeds the quick-data writer: numpy, f8 ratings."""
    from reviews4rec_amd.data import load_data
    train, _, _, hp = load_data(tiny_hp('deepconn', materialise_tiny(tmp_path)), device='cpu')
    z = np.load(os.path.join(TINY_DIR, 'deepconn_streams.npz'))
    data, y = next(iter(train.iter_review(simple=True)))
    assert all(isinstance(d, np.ndarray) for d in data) and y.dtype == np.float64
    for s in range(7):
        assert np.array_equal(data[s], z['train/0/%d' % s])


def test_quick_data_writer_round_trip(tmp_path, monkeypatch):
    """tools/make_quick_data.py (make_quick_data.py's counterpart) -> data_fast.DataLoader yields the
    reference loader's train stream again."""
    import importlib.util
    root = materialise_tiny(tmp_path)
    spec = importlib.util.spec_from_file_location(
        'make_quick_data', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools',
                                        'make_quick_data.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.chdir(tmp_path)
    data_root = os.path.relpath(os.path.dirname(os.path.dirname(os.path.dirname(root))), str(tmp_path)) + '/'
    hp = dict(tiny_hp('deepconn', root), data_dir=data_root + 'Tiny/5_core/')
    # the writer uses the reference's fixed shapes (input_length 1000); shrink them for the fixture
    real_load = __import__('reviews4rec_amd.data', fromlist=['load_data']).load_data

    def small_load(h, load_negs=True, device=None):
        h.update(input_length=hp['input_length'], narre_num_words=hp['narre_num_words'])
        return real_load(h, load_negs=load_negs, device=device)

    monkeypatch.setattr('reviews4rec_amd.data.load_data', small_load)
    mod.main(['make_quick_data.py', 'Tiny', '5', '100', 'deepconn', data_root])
    from reviews4rec_amd.data_fast import DataLoader
    fast = DataLoader(dict(hp, total_users=48, total_items=24), 'train.hdf5', device=torch.device('cpu'))
    z = np.load(os.path.join(TINY_DIR, 'deepconn_streams.npz'))
    assert len(fast) == int(z['len'][0])
    for k, (data, y) in enumerate(fast.iter()):
        for s in range(7):
            assert np.array_equal(data[s].numpy(), z['train/%d/%d' % (k, s)])
        assert np.array_equal(y.numpy(), z['train/%d/y' % k])
