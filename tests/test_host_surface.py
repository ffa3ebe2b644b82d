"""Host-surface counterparts (hyper_params / data_fast / main.train / eval.evaluate),
tested on the CPU with an oracle-backed stand-in model: the loop is model-agnostic."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLDEN_DIR, Golden, OracleModule, synthetic_review_batch


def test_common_path_matches_reference_strings():
    from reviews4rec_amd import hyper_params as H
    ref = json.load(open(os.path.join(GOLDEN_DIR, 'common_paths.json')))
    for mt, expected in ref.items():
        hp = H.default_hyper_params()
        hp['model_type'] = mt
        assert H.get_common_path(hp) == expected, mt


def test_hyper_params_defaults_and_finalize(tmp_path):
    from reviews4rec_amd import hyper_params as H
    hp = H.default_hyper_params()
    # the reference's shipped defaults (hyper_params.py:57-66)
    assert (hp['lr'], hp['weight_decay'], hp['batch_size'], hp['epochs']) == (0.002, 1e-6, 128, 2)
    assert (hp['latent_size'], hp['word_embed_size'], hp['input_length'], hp['dropout']) == (10, 64, 1000, 0.6)
    hp['model_type'] = 'NARRE'                             # crashes in the reference (missing only_reviews)
    H.finalize(hp, root=str(tmp_path))
    assert hp['data_dir'] == 'data/InstantVideo/5_core/'
    assert os.path.isdir(tmp_path / 'saved_logs') and hp['log_file'].endswith(hp['common_path'])


def test_batcher_contract(tmp_path):
    """Same shapes / dtypes / ragged tail as data_fast.DataLoader.iter (SURVEY 8c probe:
    batches of 128, 128, 44; 7 int64 tensors + float32 ratings; no shuffle)."""
    from reviews4rec_amd import data_fast
    hp = dict(batch_size=128, model_type='deepconn', data_dir='data/Tiny/5_core/')
    data, y = synthetic_review_batch(300, 50, 40, 30, 20, seed=3)
    data = [d.numpy() for d in data]
    loader = data_fast.DataLoader.from_arrays(hp, data, y.numpy(), device=torch.device('cpu'))
    assert len(loader) == 3
    sizes = []
    at = 0
    for batch, yy in loader.iter():
        assert len(batch) == 7 and all(t.dtype == torch.int64 for t in batch) and yy.dtype == torch.float32
        n = yy.shape[0]
        for t, src in zip(batch, data):
            assert np.array_equal(t.numpy(), src[at:at + n])
        at += n
        sizes.append(n)
    assert sizes == [128, 128, 44]
    # on-disk round trip (npz with the HDF5 writer's dataset names a..h, i8 / f8)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        data_fast.save_split('quick_data_deepconn/Tiny/5_core/train.npz', data, y.numpy())
        z = np.load('quick_data_deepconn/Tiny/5_core/train.npz')
        assert sorted(z.files) == list('abcdefgh') and z['a'].dtype == np.int64 and z['h'].dtype == np.float64
        disk = data_fast.DataLoader(hp, 'train.hdf5', device=torch.device('cpu'))
        assert len(disk) == 3 and disk.total == 300
        b0, y0 = next(disk.iter())
        assert np.array_equal(b0[3].numpy(), data[3][:128]) and np.allclose(y0.numpy(), y.numpy()[:128])
    finally:
        os.chdir(cwd)


class TorchAdam(torch.optim.Adam):
    pass


@pytest.mark.parametrize('case', ['mf_dot', 'deepconn_e20'])
def test_train_loop_reproduces_reference_trajectory(case):
    """main.train over a 2-batch reader == the golden 2-step trajectory (dropout 0)."""
    from reviews4rec_amd import main as M
    from reviews4rec_amd.loss import MSELoss
    g = Golden(case)

    class Reader:
        def __len__(self):
            return 2

        def iter(self, eval=False):
            for k in (0, 1):
                yield g.batch(k)

    model = OracleModule(g.hp, params=g.params())
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=g.hp['lr'],
                           weight_decay=g.hp['weight_decay'])
    metrics = M.train(model, MSELoss(g.hp), opt, Reader(), g.hp)
    n = g.arr('y0').shape[0] + g.arr('y1').shape[0]
    expected = round(float(g.arr('se0').sum() + g.arr('se1').sum()) / n, 4)
    assert metrics['MSE'] == pytest.approx(expected, abs=1e-4)
    # after the first step the weights are the golden w1; the second batch then matches se1
    model2 = OracleModule(g.hp, params=g.params('w1'))
    model2.eval()
    with torch.no_grad():
        d, y = g.batch(1)
        se1 = (model2(d) - y) ** 2
    torch.testing.assert_close(se1, g.arr('se1'), rtol=1e-4, atol=1e-5)


def test_evaluate_metrics_and_count_maps():
    from reviews4rec_amd.eval import evaluate
    from reviews4rec_amd.loss import MSELoss
    g = Golden('deepconn_e20')
    model = OracleModule(g.hp, params=g.params())

    class Reader:
        def iter(self, eval=False):
            for k in (0, 1):
                yield g.batch(k)

    d0, y0 = g.batch(0)
    d1, y1 = g.batch(1)
    se = torch.cat([(g.arr('eval0') - y0) ** 2, (g.arr('eval1') - y1) ** 2])
    user_count = {int(d0[5][0]): 3}
    metrics, ucm, icm = evaluate(model, MSELoss(g.hp), Reader(), g.hp, user_count, {}, review=True)
    assert metrics == {'MSE': round(float(se.sum()) / se.numel(), 4)}
    assert sum(len(v) for v in ucm.values()) == se.numel() == sum(len(v) for v in icm.values())
    assert 3 in ucm and 0 in icm                              # train-frequency keys (eval.py:45-53)
    assert sorted(x for v in icm.values() for x in v) == pytest.approx(sorted(se.tolist()), abs=1e-5)
    # against the reference's per-example loop restated (eval.py:42-53): same keys, same lists in
    # the same order, same side effect on the count dictionaries
    users = torch.cat([d0[5], d1[5]]).tolist()
    items = torch.cat([d0[6], d1[6]]).tolist()
    ref_uc, ref_ic, ref_ucm, ref_icm = {int(d0[5][0]): 3}, {}, {}, {}
    model.eval()
    with torch.no_grad():
        mine = torch.cat([(model(d0) - y0) ** 2, (model(d1) - y1) ** 2]).tolist()
    for u, i, e in zip(users, items, mine):
        ref_uc.setdefault(u, 0)
        ref_ic.setdefault(i, 0)
        ref_ucm.setdefault(ref_uc[u], []).append(e)
        ref_icm.setdefault(ref_ic[i], []).append(e)
    assert ucm == ref_ucm and icm == ref_icm and user_count == ref_uc


def test_eval_ranking_hr_at_1():
    from reviews4rec_amd.eval import eval_ranking
    g = Golden('mf_dot')
    model = OracleModule(g.hp, params=g.params()).eval()

    class Reader:
        def iter_negs(self, review):
            yield g.neg_batch(), torch.zeros(3)

    scores = g.arr('neg_eval')
    expected = round(100.0 * float((scores.argmax(-1) == 0).sum()) / 3, 2)
    assert eval_ranking(model, Reader(), g.hp) == {'HR@1': expected}


def test_state_dict_flush_hook_holds_the_engine_weakly_and_survives_pickling():
    """engine.flush_before_state_dict (the temporally blocked sweeps' pre-hook): state_dict() of the model and of every
    submodule flushes first; the hook does not keep the engine alive; a pickled / deep-copied model keeps working
    (its copy of the hook is inert: the copy has no engine)."""
    import copy
    import gc
    import pickle
    import torch
    from reviews4rec_amd.engine import flush_before_state_dict

    class Engine:
        flushes = 0

        def flush(self):
            self.flushes += 1

    model = torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.Sequential(torch.nn.Linear(2, 1)))
    eng = Engine()
    hooks = flush_before_state_dict(eng, model)
    assert len(hooks) == len(list(model.modules()))
    model.state_dict()
    assert eng.flushes >= 1
    before = eng.flushes
    model[1].state_dict()                                    # a submodule's own state_dict() flushes as well
    assert eng.flushes > before
    clone, deep = pickle.loads(pickle.dumps(model)), copy.deepcopy(model)
    n = eng.flushes
    clone.state_dict()
    deep.state_dict()
    assert eng.flushes == n                                  # the copies' hooks are inert
    del eng
    gc.collect()
    model.state_dict()                                       # the engine is gone: nothing to flush, nothing raised
