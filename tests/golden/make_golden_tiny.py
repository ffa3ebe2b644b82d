#!/usr/bin/env python3
"""Golden fixtures for the host surface around the models: the pickle data loader, evaluation
and HR@1 -- produced by RUNNING THE REFERENCE's data.py / eval.py / main.py (this container only).

What is written under tests/golden/tiny/ (data only, no reference source):

  dataset.json            a seeded 'Tiny' dataset in the reference's on-disk schema
                          (preprocess_random_split.py:296-316: train / test / val rating lists,
                          user_reviews, item_reviews, test_reviews, this_index_user_item,
                          num_users_items, word2vec, user_count, item_count; negs as
                          make_negative_sets.py:62-75 builds them).  tests materialise it back into
                          the .pkl files data.load_data reads.
  <mt>_streams.npz        for model_type mt in deepconn / NARRE / MF_dot: every batch of
                          train_loader.iter(), test_loader.iter(eval=True) and
                          test_loader.iter_negs(review) (data.py:250-447), slot by slot
  <mt>_eval.npz + .json   a seeded reference model (post xavier_init weights), and what
                          eval.evaluate (eval.py:11-62) returned on the test loader -- metrics, both
                          count -> [SE] maps, the count dicts after their setdefault side effect --
                          and eval.eval_ranking's HR@1 (eval.py:64-92)
  <mt>_e2e.npz + .json    (bias_only, MF_dot, MF, NeuMF -- main_NeuMF's three stages --, deepconn, deepconn++, NARRE)
                          main.main(hyper_params) end to end (main.py:400-414: load_data, xavier_init,
                          train_complete over 3 epochs with per-epoch validation, best-model reload,
                          test MSE + HR@1), dropout 0: the post-init weights (captured by wrapping
                          utils.xavier_init) and every metrics dict the run logged

Usage:  cd /tmp && python /root/repo/tests/golden/make_golden_tiny.py
"""
import json
import os
import pickle
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
sys.path.insert(0, REF)
sys.modules.setdefault('surprise', types.ModuleType('surprise'))     # data.py:4 imports it; only used at :104-106

import numpy as np
import torch

torch.set_num_threads(1)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tiny')

U, I, V = 48, 24, 50
RANK_USERS, HEAVY = 20, (20, 21)
FILES = ('train', 'test', 'val', 'user_reviews', 'item_reviews', 'test_reviews', 'this_index_user_item',
         'num_users_items', 'word2vec', 'user_count', 'item_count', 'negs')


def make_dataset(seed=20200725, E=16):
    """Same construction order as preprocess_random_split.py:205-316, on random interactions."""
    rng = np.random.default_rng(seed)
    pairs = []
    for u in range(U):
        n = 14 if u < RANK_USERS else (16 if u in HEAVY else int(rng.integers(3, 7)))   # HEAVY: > 10 train reviews
        for i in rng.choice(I, size=min(n, I), replace=False):
            pairs.append((u, int(i)))
    rng.shuffle(pairs)

    def review():
        if rng.random() < 0.1:
            return []                                    # percent_reviews_to_keep empties reviews (:195)
        return [int(t) for t in rng.integers(0, V, size=int(rng.integers(1, 15)))]   # 0 = UNK is a legal token

    by_user = {}
    for u, i in pairs:
        by_user.setdefault(u, []).append(i)
    train, test, val = [], [], []
    for u, items in by_user.items():
        if u < RANK_USERS:                               # ranking users: 7 test ratings, one 5.0 and six low
            held = items[:7]
            for k, i in enumerate(held):
                test.append([u, i, 5.0 if k == 0 else float(rng.integers(1, 5)), review()])
            rest = items[7:]
        else:
            rest = items
        for i in rest:
            r = float(rng.choice([1, 2, 3, 4, 5], p=[.05, .05, .10, .22, .58]))
            x = rng.random()
            (train if (x < 0.8 or u in HEAVY) else (test if x < 0.9 else val)).append([u, i, r, review()])
    user_reviews = {u: [] for u in range(U)}
    item_reviews = {i: [] for i in range(I)}
    tiui = {}
    for u, i, r, rev in train:
        tiui.setdefault(u, {})[i] = [len(user_reviews[u]), len(item_reviews[i])]
        user_reviews[u].append(rev)
        item_reviews[i].append(rev)
    test_reviews = {}
    for u, i, r, rev in test + val:
        test_reviews.setdefault(u, {})[i] = rev
    user_count, item_count = {}, {}
    for u, i, r, rev in train:
        user_count[u] = user_count.get(u, 0) + 1
        item_count[i] = item_count.get(i, 0) + 1
    # make_negative_sets.py:41-75
    negs, pos_of, neg_of = {}, {}, {}
    for u, i, r, rev in test:
        pos_of.setdefault(u, [])
        neg_of.setdefault(u, [])
        (pos_of if r >= 4.9 else neg_of)[u].append(i)
    for u in pos_of:
        if len(pos_of[u]) == 0 or len(set(neg_of[u])) < 5:
            continue
        pos = [pos_of[u][int(rng.integers(len(pos_of[u])))]]
        neg = set()
        while len(neg) < 5:
            neg.add(neg_of[u][int(rng.integers(len(neg_of[u])))])
        negs[u] = [pos, [int(x) for x in neg]]
    word2vec = rng.uniform(0.0, 1.0, size=(V, E)).astype(np.float32).tolist()
    strip = lambda rows: [[u, i, r] for u, i, r, rev in rows]
    return {'train': strip(train), 'test': strip(test), 'val': strip(val), 'user_reviews': user_reviews,
            'item_reviews': item_reviews, 'test_reviews': test_reviews, 'this_index_user_item': tiui,
            'num_users_items': [U, I, V - 1], 'word2vec': word2vec, 'user_count': user_count,
            'item_count': item_count, 'negs': negs}


def write_pickles(ds, root):
    os.makedirs(root, exist_ok=True)
    for name in FILES:
        with open(os.path.join(root, name + '.pkl'), 'wb') as f:
            pickle.dump(ds[name], f, 2)


def to_jsonable(ds):
    def conv(x):
        if isinstance(x, dict):
            return {str(k): conv(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [conv(v) for v in x]
        return x
    return {k: conv(v) for k, v in ds.items()}


def tiny_hp(mt, root, **kw):
    hp = {'dataset': 'Tiny', 'k_core': 5, 'percent_reviews_to_keep': 100, 'weight_decay': 1e-6, 'lr': 0.002,
          'epochs': 3, 'batch_size': 16, 'shuffle_data_every_epoch': False, 'latent_size': 6,
          'word_embed_size': 16, 'input_length': 37, 'dropout': 0.0, 'model_type': mt, 'narre_num_reviews': 10,
          'narre_num_words': 6, 'only_reviews': False, 'data_dir': root + '/'}
    hp.update(kw)
    return hp


def model_class(mt):
    if mt in ('deepconn', 'deepconn++'):
        from pytorch_models.DeepCoNN import DeepCoNN as Model
    elif mt in ('transnet', 'transnet++'):
        from pytorch_models.TransNet import TransNet as Model
    elif mt == 'NARRE':
        from pytorch_models.NARRE import NARRE as Model
    else:
        from pytorch_models.MF import MF as Model
    return Model


def record_stream(out, tag, batches):
    n = 0
    for k, (data, y) in enumerate(batches):
        for s, d in enumerate(data):
            if d is None:
                continue
            out['%s/%d/%d' % (tag, k, s)] = d.numpy().copy()
        out['%s/%d/y' % (tag, k)] = y.numpy().copy()
        n += 1
    out[tag + '/n'] = np.array(n)


def int_keys(d):
    return {str(k): v for k, v in d.items()}


def run_streams_and_eval(mt, root):
    import data as ref_data
    import eval as ref_eval
    from loss import MSELoss
    from utils import load_user_item_counts, xavier_init
    review = mt not in ('bias_only', 'MF', 'MF_dot', 'NeuMF')
    hp = tiny_hp(mt, root)
    train_loader, test_loader, val_loader, hp = ref_data.load_data(hp)
    out = {}
    record_stream(out, 'train', train_loader.iter())
    record_stream(out, 'test', test_loader.iter(eval=True))
    record_stream(out, 'val', val_loader.iter(eval=True))
    record_stream(out, 'negs', test_loader.iter_negs(review))
    out['len'] = np.array([len(train_loader), len(test_loader), len(val_loader)])
    np.savez_compressed(os.path.join(OUT, mt + '_streams.npz'), **out)

    # evaluate / eval_ranking with a seeded reference model on FRESH loaders (NARRE's pad_only pads the
    # shared review lists in place, data.py:159-170; a fresh load keeps this leg independent of the one above)
    hp = tiny_hp(mt, root)
    train_loader, test_loader, val_loader, hp = ref_data.load_data(hp)
    torch.manual_seed(7)
    model = model_class(mt)(hp)
    xavier_init(model)
    w = {'w/' + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    user_count, item_count = load_user_item_counts(hp)
    user_count[10 ** 6] = 3                                # an id the pass never sees must survive untouched
    crit = MSELoss(hp)
    metrics, ucm, icm = ref_eval.evaluate(model, crit, test_loader, hp, user_count, item_count, review)
    hr = ref_eval.eval_ranking(model, test_loader, hp, review)
    model.eval()
    with torch.no_grad():
        scores = []
        for d, y in test_loader.iter_negs(review):
            o = model(d)
            scores.append((o[0] if isinstance(o, (list, tuple)) else o).numpy().copy())
    w['neg_scores'] = np.concatenate(scores, 0)
    np.savez_compressed(os.path.join(OUT, mt + '_eval.npz'), **w)
    json.dump({'metrics': metrics, 'ranking': hr, 'user_count_mse_map': int_keys(ucm),
               'item_count_mse_map': int_keys(icm), 'user_count_after': int_keys(user_count),
               'item_count_after': int_keys(item_count),
               'hp': {k: v for k, v in hp.items() if k != 'data_dir'}},
              open(os.path.join(OUT, mt + '_eval.json'), 'w'), indent=1)
    print('%-10s streams %d/%d/%d batches, negs %d rows; evaluate %s, %s' % (
        mt, len(train_loader), len(test_loader), len(val_loader), len(w['neg_scores']), metrics, hr))


def run_e2e(mt, root, cwd):
    """main.main(hyper_params) (main.py:400-414) with two observers wrapped around utils functions."""
    import main as ref_main
    import utils as ref_utils
    os.chdir(cwd)
    os.makedirs('saved_models', exist_ok=True)
    os.makedirs('saved_logs', exist_ok=True)
    hp = tiny_hp(mt, root, log_file='saved_logs/e2e_' + mt, model_path='saved_models/e2e_' + mt)
    captured, logged = {}, []
    real_init, real_log = ref_utils.xavier_init, ref_utils.log_end_epoch

    def observing_init(model):
        real_init(model)
        tag = 'w/' if mt != 'NeuMF' else 'w_%s/' % type(model).__name__       # NeuMF: xavier_init runs on GMF and on MLP
        for k, v in model.state_dict().items():
            captured[tag + k] = v.detach().numpy().copy()

    def observing_log(hyper_params, metrics, epoch, time_elapsed, metrics_on='(VAL)'):
        logged.append({'epoch': epoch, 'on': metrics_on, 'metrics': dict(metrics)})
        real_log(hyper_params, metrics, epoch, time_elapsed, metrics_on=metrics_on)

    ref_utils.xavier_init, ref_utils.log_end_epoch = observing_init, observing_log
    try:
        torch.manual_seed(11)
        final = ref_main.main(hp)
    finally:
        ref_utils.xavier_init, ref_utils.log_end_epoch = real_init, real_log
    best = torch.load(hp['model_path'])
    for k, v in best.items():
        captured['best/' + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, mt + '_e2e.npz'), **captured)
    json.dump({'final': final, 'logged': logged, 'hp': {k: v for k, v in hp.items() if k not in ('data_dir', 'log_file', 'model_path')}},
              open(os.path.join(OUT, mt + '_e2e.json'), 'w'), indent=1)
    print('%-10s e2e: %s' % (mt, final))


def main():
    os.makedirs(OUT, exist_ok=True)
    cwd = tempfile.mkdtemp(prefix='r4r_cwd_')
    os.chdir(cwd)                                         # hyper_params.py:87-88 creates saved_* in the cwd
    ds = make_dataset()
    json.dump(to_jsonable(ds), open(os.path.join(OUT, 'dataset.json'), 'w'))
    root = os.path.join(cwd, 'data', 'Tiny', '5_core')
    write_pickles(ds, root)
    print('Tiny: %d train / %d test / %d val ratings, %d ranking users' % (
        len(ds['train']), len(ds['test']), len(ds['val']), len(ds['negs'])))
    for mt in ('deepconn', 'NARRE', 'MF_dot', 'transnet++'):
        run_streams_and_eval(mt, root)
    for mt in ('bias_only', 'MF_dot', 'MF', 'NeuMF', 'deepconn', 'deepconn++', 'NARRE'):
        run_e2e(mt, root, cwd)


if __name__ == '__main__':
    main()
