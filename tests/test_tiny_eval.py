"""eval.evaluate / eval.eval_ranking / main.main against what the REFERENCE's eval.py and main.py
returned on the Tiny dataset (tests/golden/make_golden_tiny.py ran them; fixtures are data only).

CPU tests drive this package's eval.py + data.py with the oracle standing in for the model (the host
logic is what is under test); the -m gpu tests run the HIP models and the fused native engines."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import TINY_DIR, OracleModule, materialise_tiny, tiny_hp

EVAL_MODELS = ['deepconn', 'NARRE', 'MF_dot', 'transnet++']
E2E_MODELS = ['bias_only', 'MF_dot', 'MF', 'NeuMF', 'deepconn', 'deepconn++', 'NARRE']
MSE_TOL = 2e-4          # metrics are rounded to 4 decimals (eval.py:56): one unit of rounding + fp32 noise


def _fixture(mt, kind):
    meta = json.load(open(os.path.join(TINY_DIR, '%s_%s.json' % (mt, kind))))
    z = np.load(os.path.join(TINY_DIR, '%s_%s.npz' % (mt, kind)))
    return meta, z


def _weights(z, prefix='w/'):
    return {k[len(prefix):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(prefix)}


def _counts(root):
    from reviews4rec_amd.utils import load_user_item_counts
    user_count, item_count = load_user_item_counts({'data_dir': root})
    user_count[10 ** 6] = 3          # the generator planted it: an id the pass never sees stays untouched
    return user_count, item_count


def _check_eval(meta, metrics, ucm, icm, user_count, item_count, hr, rtol):
    for k, v in meta['metrics'].items():
        assert metrics[k] == pytest.approx(v, abs=MSE_TOL), k
    assert set(metrics) == set(meta['metrics'])
    for mine, want in ((ucm, meta['user_count_mse_map']), (icm, meta['item_count_mse_map'])):
        assert sorted(mine) == sorted(int(k) for k in want)            # train-frequency keys (eval.py:45-53)
        for k, vals in want.items():
            assert len(mine[int(k)]) == len(vals)
            np.testing.assert_allclose(mine[int(k)], vals, rtol=rtol, atol=1e-5)   # same SEs in the same order
    assert {int(k): v for k, v in user_count.items()} == {int(k): v for k, v in meta['user_count_after'].items()}
    assert {int(k): v for k, v in item_count.items()} == {int(k): v for k, v in meta['item_count_after'].items()}
    assert hr == meta['ranking']


@pytest.mark.parametrize('mt', EVAL_MODELS)
def test_evaluate_and_ranking_host_logic_vs_reference(mt, tmp_path):
    from reviews4rec_amd.data import load_data
    from reviews4rec_amd.eval import evaluate, eval_ranking
    from reviews4rec_amd.loss import MSELoss
    meta, z = _fixture(mt, 'eval')
    root = materialise_tiny(tmp_path)
    train, test, val, hp = load_data(tiny_hp(mt, root), device='cpu')
    model = OracleModule(hp, params=_weights(z))
    user_count, item_count = _counts(root)
    review = mt not in ('bias_only', 'MF', 'MF_dot', 'NeuMF')
    metrics, ucm, icm = evaluate(model, MSELoss(hp), test, hp, user_count, item_count, review)
    hr = eval_ranking(model, test, hp, review)
    _check_eval(meta, metrics, ucm, icm, user_count, item_count, hr, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('engine', ['native', 'module'])
@pytest.mark.parametrize('mt', EVAL_MODELS)
def test_evaluate_and_ranking_on_the_device_vs_reference(mt, engine, tmp_path):
    """The reference's evaluate() / eval_ranking() numbers from the HIP path: batches built on the device
    from the token pools, scored by the fused native engine's eval forward (or the op-by-op modules)."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from reviews4rec_amd.data import load_data
    from reviews4rec_amd.eval import evaluate, eval_ranking
    from reviews4rec_amd.loss import MSELoss
    meta, z = _fixture(mt, 'eval')
    root = materialise_tiny(tmp_path)
    train, test, val, hp = load_data(tiny_hp(mt, root), device='cuda')
    model = reviews4rec_amd.get_model_class(mt)(hp)
    model.load_state_dict(_weights(z), strict=True)
    model = model.cuda()
    eng = M.make_engine(dict(hp, engine='native'), model) if engine == 'native' else None
    user_count, item_count = _counts(root)
    review = mt not in ('bias_only', 'MF', 'MF_dot', 'NeuMF')
    metrics, ucm, icm = evaluate(model, MSELoss(hp), test, hp, user_count, item_count, review, engine=eng)
    hr = eval_ranking(model, test, hp, review, engine=eng)
    _check_eval(meta, metrics, ucm, icm, user_count, item_count, hr, rtol=1e-4)
    # the [rows, 6] ranking scores themselves
    scores = []
    with torch.no_grad():
        for data, y in test.iter_negs(review):
            if eng is not None:
                scores.append(eng.predict(data, None)[0].clone())
            else:
                o = model(data)
                scores.append(o[0] if isinstance(o, (list, tuple)) else o)
    np.testing.assert_allclose(torch.cat(scores).cpu().numpy(), z['neg_scores'], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('mt', ['deepconn', 'NARRE', 'MF_dot', 'transnet++'])
def test_validation_launch_size_does_not_change_the_scores(mt, tmp_path):
    """evaluate() / eval_ranking() score larger slices of the stream than the reference's batch_size when a
    native engine does the scoring (eval._launch_size): same SEs in the same order, same HR@1 as with the
    reference's slices -- a rating's score does not depend on what shares its launch."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from reviews4rec_amd.data import load_data
    from reviews4rec_amd.eval import evaluate, eval_ranking, _launch_size, EVAL_LAUNCH
    from reviews4rec_amd.loss import MSELoss
    meta, z = _fixture(mt, 'eval')
    root = materialise_tiny(tmp_path)
    train, test, val, hp = load_data(tiny_hp(mt, root), device='cuda')
    model = reviews4rec_amd.get_model_class(mt)(hp)
    model.load_state_dict(_weights(z), strict=True)
    model = model.cuda()
    eng = M.make_engine(dict(hp, engine='native'), model)
    review = mt not in ('bias_only', 'MF', 'MF_dot', 'NeuMF')
    assert _launch_size(test, hp, eng, EVAL_LAUNCH) == EVAL_LAUNCH * hp['batch_size']
    assert _launch_size(test, hp, None, EVAL_LAUNCH) is None           # module path: the reference's slices
    got = {}
    for label, size in (('default', None), ('reference', hp['batch_size']), ('odd', 7)):
        h = dict(hp) if size is None else dict(hp, eval_batch_size=size)
        uc, ic = _counts(root)
        got[label] = evaluate(model, MSELoss(hp), test, h, uc, ic, review, engine=eng) + (eval_ranking(model, test, h, review, engine=eng),)
    for label in ('reference', 'odd'):
        assert set(got[label][0]) == set(got['default'][0])
        for k in got[label][0]:                                       # MSE (+ TransNet's means of per-slice means)
            assert got[label][0][k] == pytest.approx(got['default'][0][k], abs=1e-4), k
        for a, b in ((got[label][1], got['default'][1]), (got[label][2], got['default'][2])):
            assert sorted(a) == sorted(b)
            for k in a:
                np.testing.assert_allclose(a[k], b[k], rtol=1e-6, atol=1e-7)
        assert got[label][3] == got['default'][3]


@pytest.mark.gpu
@pytest.mark.parametrize('mt', E2E_MODELS)
def test_main_end_to_end_vs_reference(mt, tmp_path, monkeypatch):
    """main.main(hyper_params) == the reference's main.main on the same dataset directory from the same
    post-xavier_init weights (dropout 0): validation MSE of every epoch, test MSE, HR@1, and the
    best-on-validation weights it saved."""
    from reviews4rec_amd import main as M
    meta, z = _fixture(mt, 'e2e')
    root = materialise_tiny(tmp_path)
    hp = tiny_hp(mt, root, log_file=str(tmp_path / 'log'), model_path=str(tmp_path / 'model'))
    def fixture_init(model):                   # the reference drew these with its own RNG stream
        # (NeuMF: main_NeuMF runs xavier_init on the GMF and on the MLP stage model; NeuMF itself is built by init())
        init = _weights(z, 'w/' if mt != 'NeuMF' else 'w_%s/' % type(model).__name__)
        model.load_state_dict({k: v.to(next(model.parameters()).device) for k, v in init.items()}, strict=True)

    logged = []
    real_log = M.log_end_epoch

    def observing_log(hyper_params, metrics, epoch, t, metrics_on='(VAL)'):
        logged.append({'epoch': epoch, 'on': metrics_on, 'metrics': dict(metrics)})
        real_log(hyper_params, metrics, epoch, t, metrics_on=metrics_on)

    monkeypatch.setattr(M, 'xavier_init', fixture_init)
    monkeypatch.setattr(M, 'log_end_epoch', observing_log)
    final = M.main(hp)
    assert len(logged) == len(meta['logged'])
    for mine, want in zip(logged, meta['logged']):
        assert (mine['epoch'], mine['on']) == (want['epoch'], want['on'])
        assert mine['metrics']['MSE'] == pytest.approx(want['metrics']['MSE'], abs=MSE_TOL)
    assert final['MSE'] == pytest.approx(meta['final']['MSE'], abs=MSE_TOL)
    assert final['HR@1'] == meta['final']['HR@1']
    best = torch.load(hp['model_path'], map_location='cpu')
    for k, v in _weights(z, 'best/').items():
        if k.startswith('attention_scorer_') and k.endswith('.3.bias'):
            continue       # shift-invariant softmax input: gradient is rounding noise (ill-conditioned)
        np.testing.assert_allclose(best[k].numpy(), v.numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
