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


def _random_dataset(seed, U=17, I=11, V=40):
    """Random interactions in the reference's schema with the awkward cases on purpose: users / items without
    any train review, empty reviews, owners with more than ten reviews, reviews longer than narre_num_words."""
    rng = np.random.default_rng(seed)
    pairs = [(u, i) for u in range(U) for i in range(I) if rng.random() < 0.45 and u != 3 and i != 2]
    rng.shuffle(pairs)
    cut = int(0.75 * len(pairs))
    rev = lambda: [] if rng.random() < 0.15 else [int(t) for t in rng.integers(0, V, size=int(rng.integers(1, 23)))]
    train = [[u, i, float(rng.integers(1, 6))] for u, i in pairs[:cut]]
    held = [[u, i, float(rng.integers(1, 6))] for u, i in pairs[cut:]] + [[3, 2, 4.0]]   # a pair of two unseen ids
    user_reviews, item_reviews, tiui = {u: [] for u in range(U)}, {i: [] for i in range(I)}, {}
    for u, i, r in train:
        text = rev()
        tiui.setdefault(u, {})[i] = [len(user_reviews[u]), len(item_reviews[i])]
        user_reviews[u].append(text)
        item_reviews[i].append(text)
    test_reviews = {}
    for u, i, r in held:
        test_reviews.setdefault(u, {})[i] = rev()
    negs = {}
    for u in range(U):
        mine = [i for uu, i, r in held if uu == u]
        if len(mine) >= 1:
            others = [int(x) for x in rng.choice(I, size=5, replace=False)]
            negs[u] = [[mine[0]], others]
    return dict(train=train, held=held, user_reviews=user_reviews, item_reviews=item_reviews, tiui=tiui,
                test_reviews=test_reviews, negs=negs, U=U, I=I)


def _reference_fields(ds, hp, u, i, pos_i, train):
    """The five review slots of ONE rating, restated literally from data.py:144-236 + 273-279 (list surgery,
    nothing shared with reviews4rec_amd.data).  pos_i: the item remove_overlap is called with (iter_negs: the
    positive's)."""
    narre, T, R, W = hp['model_type'] == 'NARRE', hp['input_length'], hp['narre_num_reviews'], hp['narre_num_words']
    u_r, i_r = [list(r) for r in ds['user_reviews'][u]], [list(r) for r in ds['item_reviews'][i]]
    u2i = [0] * len(ds['user_reviews'][u])
    for item, (ku, ki) in ds['tiui'].get(u, {}).items():
        u2i[ku] = item
    i2u = [0] * len(ds['item_reviews'][pos_i])
    for user, per in ds['tiui'].items():
        if pos_i in per:
            i2u[per[pos_i][1]] = user
    if train:
        ku, ki = ds['tiui'][u][pos_i]
        this = [u_r[ku]]
        what = [x for k, x in enumerate(u2i) if k != ku]
        who = [x for k, x in enumerate(i2u) if k != ki]
        u_r = [r for k, r in enumerate(u_r) if k != ku]
        i_r = [r for k, r in enumerate(i_r) if k != ki]
    else:
        this, what, who = [list(ds['test_reviews'][u][pos_i])], list(u2i), list(i2u)
    who = (who + [hp['total_users'] + 1] * 10)[:10]
    what = (what + [hp['total_items'] + 1] * 10)[:10]

    def doc(reviews):
        if narre:
            rows = [(r + [0] * W)[:W] for r in reviews]
            return (rows + [[0] * W] * R)[:R]
        flat = [t for r in reviews for t in r]
        return (flat + [0] * T)[:T]
    return doc(this), who, what, doc(u_r), doc(i_r)


@pytest.mark.parametrize('mt', ['deepconn', 'NARRE'])
@pytest.mark.parametrize('seed', [1, 2, 3])
def test_pool_arithmetic_equals_the_list_surgery_on_random_datasets(mt, seed):
    """Property test of the token-pool index arithmetic (host evaluation) against a literal restatement of the
    reference's per-rating list surgery, on random datasets with the awkward cases (no reviews, empty reviews, more
    than ten reviews, over-long reviews, held-out pairs of unseen ids), train / held-out / negatives streams."""
    from reviews4rec_amd.data import DataLoader
    ds = _random_dataset(seed)
    hp = dict(model_type=mt, batch_size=7, input_length=19, narre_num_reviews=10, narre_num_words=5,
              total_users=ds['U'], total_items=ds['I'])
    train = DataLoader(hp, ds['train'], ds['user_reviews'], ds['item_reviews'], ds['negs'],
                       this_index_user_item=ds['tiui'], device='cpu')
    held = DataLoader(hp, ds['held'], ds['user_reviews'], ds['item_reviews'], ds['negs'],
                      test_reviews=ds['test_reviews'], train_loader=train, device='cpu')
    for loader, rows, is_train in ((train, ds['train'], True), (held, ds['held'], False)):
        at = 0
        for data, y in loader.iter():
            for b in range(y.shape[0]):
                u, i, r = rows[at]
                want = _reference_fields(ds, hp, u, i, i, is_train)
                for s in range(5):
                    assert data[s][b].tolist() == want[s], (mt, seed, is_train, at, s)
                assert (int(data[5][b]), int(data[6][b]), float(y[b])) == (u, i, r)
                at += 1
        assert at == len(rows)
    at = 0
    users = list(ds['negs'])
    for data, y in held.iter_negs(True):
        for b in range(y.shape[0]):
            u = users[at]
            cands = [ds['negs'][u][0][0]] + ds['negs'][u][1]
            for c, i2 in enumerate(cands):
                want = _reference_fields(ds, hp, u, i2, cands[0], False)
                for s in range(5):
                    assert data[s][b, c].tolist() == want[s], (mt, seed, 'negs', at, c, s)
                assert (int(data[5][b, c]), int(data[6][b, c])) == (u, i2)
            at += 1
    assert at == len(users)


@pytest.mark.gpu
@pytest.mark.parametrize('mt', ['deepconn', 'NARRE'])
def test_device_batches_equal_host_batches_on_random_datasets(mt):
    """r4r_batch_build against the host evaluation of the same pools, every stream, random awkward datasets."""
    from reviews4rec_amd.data import DataLoader
    for seed in (4, 5):
        ds = _random_dataset(seed, U=60, I=25)
        hp = dict(model_type=mt, batch_size=16, input_length=37, narre_num_reviews=10, narre_num_words=6,
                  total_users=ds['U'], total_items=ds['I'])
        mk = lambda dev: (lambda tr: (tr, DataLoader(hp, ds['held'], ds['user_reviews'], ds['item_reviews'], ds['negs'],
                                                     test_reviews=ds['test_reviews'], train_loader=tr, device=dev)))(
            DataLoader(hp, ds['train'], ds['user_reviews'], ds['item_reviews'], ds['negs'],
                       this_index_user_item=ds['tiui'], device=dev))
        (ct, ch), (gt, gh) = mk('cpu'), mk('cuda')
        for a, b in ((ct.iter(), gt.iter()), (ch.iter(), gh.iter()), (ch.iter_negs(True), gh.iter_negs(True))):
            for (cd, cy), (gd, gy) in zip(a, b):
                for s in range(7):
                    assert torch.equal(cd[s], gd[s].cpu()), (seed, s)
                assert torch.equal(cy, gy.cpu())
