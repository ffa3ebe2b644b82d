"""Shared test helpers: golden-fixture loading and synthetic batches."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

GOLDEN_CASES = ['mf_bias_only', 'mf_dot', 'mf_full', 'deepconn_e20', 'deepconn_e64',
                'deepconnpp_e20', 'narre_e16', 'transnet_e16', 'transnetpp_e16',
                'neumf_gmf', 'neumf_mlp', 'neumf_full']
TRAINABLE_CASES = [c for c in GOLDEN_CASES if not c.startswith('transnet')]

_INT_KEYS = ('latent_size', 'word_embed_size', 'input_length', 'total_users', 'total_items',
             'narre_num_reviews', 'narre_num_words', 'batch_size')
_FLOAT_KEYS = ('dropout', 'lr', 'weight_decay')


class Golden:
    """One npz fixture written by tests/golden/make_golden.py."""

    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.z = {k: z[k] for k in z.files}
        hp = dict(zip(self.z['hp_keys'].tolist(), self.z['hp_vals'].tolist()))
        for k in _INT_KEYS:
            hp[k] = int(hp[k])
        for k in _FLOAT_KEYS:
            hp[k] = float(hp[k])
        self.hp = hp
        self.vocab = int(self.z['vocab'])

    def group(self, prefix, device='cpu'):
        pre = prefix + '/'
        return {k[len(pre):]: torch.from_numpy(v.copy()).to(device)
                for k, v in self.z.items() if k.startswith(pre)}

    def params(self, tag='w', device='cpu'):
        return self.group(tag, device)

    def batch(self, k, device='cpu'):
        d = self.group('b%d' % k, device)
        data = [d[str(s)] for s in range(7)]
        return data, torch.from_numpy(self.z['y%d' % k].copy()).to(device)

    def neg_batch(self, device='cpu'):
        d = self.group('neg', device)
        return [d[str(s)] for s in range(7)]

    def arr(self, key):
        return torch.from_numpy(self.z[key].copy())

    def has(self, key):
        return key in self.z


def synthetic_review_batch(B, T, V, U, I, seed=0, R=None, W=None, device='cpu'):
    """Amazon-shaped random batch in the 7-slot layout of data_fast.py:101-109."""
    rng = np.random.default_rng(seed)

    def docs(shape):
        tok = rng.integers(1, V, size=shape)
        fill = rng.integers(1, shape[-1] + 1, size=shape[:-1])
        return np.where(np.arange(shape[-1]) < fill[..., None], tok, 0)

    if R is None:
        ur, ir = docs((B, T)), docs((B, T))
    else:
        ur, ir = docs((B, R, W)), docs((B, R, W))
    data = [docs((B, T)), rng.integers(0, U + 2, size=(B, 10)), rng.integers(0, I + 2, size=(B, 10)),
            ur, ir, rng.integers(0, U, size=(B,)), rng.integers(0, I, size=(B,))]
    y = rng.integers(1, 6, size=(B,)).astype(np.float32)
    return [torch.from_numpy(np.ascontiguousarray(d.astype(np.int64))).to(device) for d in data], \
        torch.from_numpy(y).to(device)


class OracleModule(torch.nn.Module):
    """CPU stand-in with the drop-in models' duck type (Model(hp) -> forward(data)),
    computing with the oracle.  Lets the host loop / batcher / data-parallel logic be
    tested where no GPU exists.  TEST ONLY."""

    def __init__(self, hyper_params, params=None, vocab=50, seed=0):
        super().__init__()
        import oracle
        self.hyper_params = hyper_params
        P = params if params is not None else oracle.init_params(hyper_params, vocab_size=vocab, seed=seed)
        self.names = list(P)
        self.plist = torch.nn.ParameterList([
            torch.nn.Parameter(v.clone(), requires_grad=not k.endswith('word2vec.weight')) for k, v in P.items()])

    def as_dict(self):
        return dict(zip(self.names, self.plist))

    def forward(self, data):
        import oracle
        return oracle.model_forward(self.as_dict(), data, self.hyper_params, train=self.training)


TINY_DIR = os.path.join(GOLDEN_DIR, 'tiny')
TINY_FILES = ('train', 'test', 'val', 'user_reviews', 'item_reviews', 'test_reviews', 'this_index_user_item',
              'num_users_items', 'word2vec', 'user_count', 'item_count', 'negs')


def materialise_tiny(root):
    """Write tests/golden/tiny/dataset.json back into the .pkl files of a reference dataset directory
    (preprocess_random_split.py:296-316, make_negative_sets.py:83) -> the data_dir string."""
    import json
    import pickle
    ds = json.load(open(os.path.join(TINY_DIR, 'dataset.json')))

    def ints(d, depth):
        if depth == 0 or not isinstance(d, dict):
            return d
        return {int(k): ints(v, depth - 1) for k, v in d.items()}

    ds['user_reviews'], ds['item_reviews'] = ints(ds['user_reviews'], 1), ints(ds['item_reviews'], 1)
    ds['test_reviews'], ds['this_index_user_item'] = ints(ds['test_reviews'], 2), ints(ds['this_index_user_item'], 2)
    ds['user_count'], ds['item_count'] = ints(ds['user_count'], 1), ints(ds['item_count'], 1)
    ds['negs'] = ints(ds['negs'], 1)
    root = os.path.join(str(root), 'data', 'Tiny', '5_core')
    os.makedirs(root, exist_ok=True)
    for name in TINY_FILES:
        with open(os.path.join(root, name + '.pkl'), 'wb') as f:
            pickle.dump(ds[name], f, 2)
    return root + '/'


def tiny_hp(model_type, data_dir, **kw):
    """hyper_params of a tests/golden/tiny fixture (the generator's tiny_hp) + the data directory."""
    import json
    name = model_type + ('_eval' if model_type in ('deepconn', 'NARRE', 'MF_dot', 'transnet++') else '_e2e')
    hp = dict(json.load(open(os.path.join(TINY_DIR, name + '.json')))['hp'])
    for k in ('total_users', 'total_items', 'total_words'):
        hp.pop(k, None)                                   # load_data sets them (data.py:469-471)
    hp['data_dir'] = data_dir
    hp.update(kw)
    return hp
