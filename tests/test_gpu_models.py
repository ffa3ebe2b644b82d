"""Model-level parity on the GPU: the drop-in Model(hyper_params) classes, driven
exactly like the reference's host loop (main.py:26-60), against

  * the golden fixtures generated from the reference itself (tests/golden/), and
  * the CPU oracle on fresh seeded inputs at larger shapes.

North-star tolerance: predicted ratings within 1e-4 MSE of the reference CPU
path (BASELINE.json); the asserts below are far tighter (1e-5 relative).
"""
import copy

import pytest
import torch

import oracle
from helpers import GOLDEN_CASES, TRAINABLE_CASES, Golden, synthetic_review_batch
from test_oracle_golden import ill_conditioned

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build_model(g, params=None, dropout=None):
    import reviews4rec_amd
    hp = dict(g.hp)
    if dropout is not None:
        hp['dropout'] = dropout
    P = params if params is not None else g.params()
    key = 'target.word2vec.weight' if hp['model_type'].startswith('transnet') else 'word2vec.weight'
    if key in P:
        hp['word_vectors'] = P[key].numpy()
    model = reviews4rec_amd.get_model_class(hp['model_type'])(hp)
    missing = model.load_state_dict(P, strict=True)       # key-for-key the reference's state_dict
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(DEV), hp


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_eval_forward_matches_reference_golden(case):
    g = Golden(case)
    model, _ = build_model(g)
    model.eval()
    with torch.no_grad():
        for k in (0, 1):
            data, _ = g.batch(k, DEV)
            out = model(data)
            if case.startswith('transnet'):
                torch.testing.assert_close(out[0].cpu(), g.arr('eval%d/src' % k), rtol=1e-5, atol=1e-5)
                torch.testing.assert_close(out[1].cpu(), g.arr('eval%d/tgt' % k), rtol=1e-5, atol=1e-5)
                torch.testing.assert_close(out[2].cpu(), g.arr('eval%d/transform' % k), rtol=1e-5, atol=1e-5)
            else:
                ref = g.arr('eval%d' % k)
                torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
                assert float(((out.cpu() - ref) ** 2).mean()) < 1e-4       # the north-star bound
        out = model(g.neg_batch(DEV))
        if case.startswith('transnet'):
            out = out[0]
        assert tuple(out.shape) == (3, 6)
        torch.testing.assert_close(out.cpu(), g.arr('neg_eval'), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('case', TRAINABLE_CASES)
def test_training_trajectory_matches_reference_golden(case):
    """zero_grad -> forward -> SE -> mean -> backward -> Adam, three steps, dropout 0."""
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    crit = MSELoss(hp)
    opt = Adam(model.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])
    names = {id(p): k for k, p in model.named_parameters()}
    for step in range(3):
        data, y = g.batch(step % 2, DEV)
        model.zero_grad()
        opt.zero_grad()
        out = model(data)
        se = crit(out, y, return_mean=False)
        torch.testing.assert_close(se.detach().cpu(), g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
        torch.mean(se).backward()
        if step == 0:
            ref_g = g.group('g0')
            got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
            assert set(got) == set(ref_g)                  # unused params keep grad None
            for k, v in ref_g.items():
                if ill_conditioned(k):
                    continue
                torch.testing.assert_close(got[k].cpu(), v, rtol=1e-4, atol=1e-6, msg=lambda m: k + ': ' + m)
        opt.step()
        if step in (0, 2):
            sd = model.state_dict()
            for k, v in g.params('w%d' % (step + 1)).items():
                if ill_conditioned(k):
                    continue
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    for p in model.parameters():
        k = names[id(p)]
        if id(p) in opt.state and not ill_conditioned(k):
            torch.testing.assert_close(opt.state[id(p)]['exp_avg'].cpu(), g.group('m3')[k], rtol=1e-4, atol=1e-7,
                                       msg=lambda m: k + ': ' + m)
            torch.testing.assert_close(opt.state[id(p)]['exp_avg_sq'].cpu(), g.group('v3')[k], rtol=1e-4,
                                       atol=1e-9, msg=lambda m: k + ': ' + m)


@pytest.mark.parametrize('case', ['mf_full', 'deepconnpp_e20', 'narre_e16', 'transnetpp_e16', 'neumf_mlp', 'neumf_full'])
def test_train_mode_dropout_with_injected_masks(case):
    """The device draws the masks (Philox); the same multipliers injected into the CPU
    oracle must reproduce the train-mode outputs (streams themselves are
    not comparable with the reference's unseeded global RNG)."""
    from reviews4rec_amd import ops
    g = Golden(case)
    model, hp = build_model(g, dropout=0.5)
    model.train()
    ops.DropoutState.manual_seed(99)
    ops.DropoutState.record = {}
    data, _ = g.batch(0, DEV)
    out = model(data)
    masks = {k: v.cpu() for k, v in ops.DropoutState.record.items()}
    ops.DropoutState.record = None
    assert masks, 'no dropout site fired'
    ref = oracle.model_forward(g.params(), g.batch(0)[0], hp, train=True, masks=masks)
    if case.startswith('transnet'):
        for a, b in zip(out, ref):
            torch.testing.assert_close(a.detach().cpu(), b, rtol=1e-5, atol=1e-5)
    else:
        torch.testing.assert_close(out.detach().cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('case', ['transnet_e16', 'transnetpp_e16'])
def test_transnet_three_optimiser_step_matches_reference_golden(case):
    """SURVEY 8 row a-12 on the HIP path: reviews4rec_amd.main.train drives the 3 backward passes /
    3 fused-Adam steps of main.py:35-53; the fused Adam writes through raw pointers, which is the
    torch-0.4 write-through behaviour the fixture was generated with."""
    from reviews4rec_amd import main as M
    from reviews4rec_amd.loss import MSELoss
    g = Golden(case)
    model, hp = build_model(g)
    optimizer = M.make_optimizer(hp, model)
    sse = 0.0
    for step in range(3):
        class OneBatch:
            def iter(self, eval=False):
                yield g.batch(step % 2, DEV)
        metrics = M.train(model, MSELoss(hp), optimizer, OneBatch(), hp)
        ref_se = g.arr('tn_se%d' % step)
        assert metrics['MSE'] == pytest.approx(round(float(ref_se.mean()), 4), abs=2e-4)
        aux = g.arr('tn_aux%d' % step)
        assert metrics['MSE_target'] == pytest.approx(float(aux[0]), abs=2e-4)
        assert metrics['MSE_transform'] == pytest.approx(float(aux[1]), abs=2e-4)
        if step in (0, 2):
            sd = model.state_dict()
            for k, v in g.params('tn_w%d' % (step + 1)).items():
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=1e-5, msg=lambda m: k + ': ' + m)


def test_deepconn_full_size_batch_against_oracle_and_properties():
    """BASELINE config 3 shape: B=128, T=1000, E=300, 100 filters.  The oracle checks a
    sample of rows (seconds on CPU); the whole batch is checked through
    size-independent properties: row permutation equivariance and duplicate rows."""
    import reviews4rec_amd
    B, T, E, V, U, I = 128, 1000, 300, 5000, 1000, 500
    hp = dict(model_type='deepconn', latent_size=10, word_embed_size=E, input_length=T, dropout=0.0,
              total_users=U, total_items=I, lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=3)
    hp['word_vectors'] = P['word2vec.weight'].numpy()
    model = reviews4rec_amd.get_model_class('deepconn')(hp)
    model.load_state_dict(P)
    model = model.to(DEV).eval()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=11)
    data[3][5] = data[3][4]                                 # duplicate rows -> identical predictions
    data[4][5] = data[4][4]
    with torch.no_grad():
        out = model([d.to(DEV) for d in data]).cpu()
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
        out_p = model([d[perm].to(DEV) for d in data]).cpu()
    assert torch.equal(out_p, out[perm])                    # bit-exact: tiles never span documents
    assert out[4] == out[5]
    rows = [0, 4, 63, 127]
    ref = oracle.model_forward(P, [d[rows] for d in data], hp, train=False)
    torch.testing.assert_close(out[rows], ref, rtol=1e-5, atol=1e-5)
    assert float(((out[rows] - ref) ** 2).mean()) < 1e-4


@pytest.mark.parametrize('mt', ['NARRE', 'transnet++'])
def test_review_models_at_baseline_shapes_against_oracle_and_properties(mt):
    """BASELINE configs 4 and 5 at their batch shapes -- NARRE: B=128 x 10 reviews x 100 words per side, E=64,
    Kindle's 68,223 users / 61,934 items; TransNet++: B=128, three 1,000-word documents per rating, E=64, a
    100,000-word vocabulary -- through the fused native engines' eval forward.  The oracle checks a sample of rows
    (seconds on CPU); the whole batch through size-independent properties: row-permutation equivariance (bit-exact:
    no tile, segment or workgroup spans ratings in a way that depends on their order) and duplicated ratings."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    if mt == 'NARRE':
        B, T, E, V, U, I, R, W = 128, 100, 64, 50002, 68223, 61934, 10, 100
    else:
        B, T, E, V, U, I, R, W = 128, 1000, 64, 100000, 50000, 20000, None, None
    hp = dict(model_type=mt, latent_size=10, word_embed_size=E, input_length=T, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, narre_num_reviews=10, narre_num_words=100, batch_size=B)
    P = oracle.init_params(hp, vocab_size=V, seed=21)
    table = P['target.word2vec.weight' if mt.startswith('transnet') else 'word2vec.weight']
    model = reviews4rec_amd.get_model_class(mt)(dict(hp, word_vectors=table.numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).eval()
    eng = M.make_engine(dict(hp, engine='native'), model)
    data, y = synthetic_review_batch(B, T, V, U, I, seed=31, R=R, W=W)
    for slot in range(7):                                   # rating 5 := rating 4
        data[slot][5] = data[slot][4]
    out = eng.predict([d.to(DEV) for d in data], None)[0].cpu().clone()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    out_p = eng.predict([d[perm].to(DEV) for d in data], None)[0].cpu().clone()
    assert torch.equal(out_p, out[perm])
    assert out[4] == out[5]
    rows = [0, 4, 77, 127]
    ref = oracle.model_forward(P, [d[rows] for d in data], hp, train=False)
    ref = ref[0] if isinstance(ref, (list, tuple)) else ref
    torch.testing.assert_close(out[rows], ref, rtol=1e-4, atol=1e-4)
    assert float(((out[rows] - ref) ** 2).mean()) < 1e-4    # SURVEY 8d's parity bound


def test_deepconn_one_step_at_baseline_shape_against_oracle():
    """One full training step (B=16 rows of the config-3 shape) vs the CPU oracle."""
    import reviews4rec_amd
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    B, T, E, V, U, I = 16, 1000, 300, 3000, 100, 50
    hp = dict(model_type='deepconn', latent_size=10, word_embed_size=E, input_length=T, dropout=0.0,
              total_users=U, total_items=I, lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=5)
    hpm = dict(hp, word_vectors=P['word2vec.weight'].numpy())
    model = reviews4rec_amd.get_model_class('deepconn')(hpm)
    model.load_state_dict(P)
    model = model.to(DEV).train()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=12)
    opt = Adam(model.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])
    out = model([d.to(DEV) for d in data])
    se = MSELoss(hp)(out, y.to(DEV), return_mean=False)
    torch.mean(se).backward()
    opt.step()
    ref_P = copy.deepcopy(P)
    sse, grads = oracle.train_step(ref_P, data, y, hp, oracle.AdamState())
    torch.testing.assert_close(se.detach().sum().cpu(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    got = {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}
    for k, v in grads.items():
        if v is None:
            assert k not in got
            continue
        torch.testing.assert_close(got[k], v, rtol=2e-4, atol=1e-6, msg=lambda m: k + ': ' + m)
    # Adam's first step is lr * g / (|g| + eps): where |g| ~ eps = 1e-8 a 1e-9 difference in g
    # moves the weight by a visible fraction of lr, so weights are compared where the gradient
    # is well above eps (the gradients themselves were compared above).
    sd = model.state_dict()
    for k, v in ref_P.items():
        mine = sd[k].cpu()
        if k in grads and grads[k] is not None:
            solid = grads[k].abs() > 1e-6
            torch.testing.assert_close(mine[solid], v[solid], rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
            assert (mine - v).abs().max() < 2.5e-3          # never more than ~one lr step apart
        else:
            torch.testing.assert_close(mine, v, rtol=0, atol=0, msg=lambda m: k + ': ' + m)


def test_no_silent_cpu_fallback():
    """CPU tensors must be refused, not computed some other way."""
    import reviews4rec_amd
    g = Golden('mf_dot')
    hp = dict(g.hp)
    model = reviews4rec_amd.get_model_class('MF_dot')(hp)   # left on the CPU on purpose
    with pytest.raises(RuntimeError, match='ROCm device'):
        model(g.batch(0)[0])


def test_batcher_double_buffered_h2d_feeds_the_native_step():
    """data_fast.DataLoader on the GPU: pinned arrays, batch k+1 copied on a copy stream while
    batch k trains; every batch must arrive intact and in order (ragged tail included)."""
    import numpy as np
    from reviews4rec_amd import data_fast
    hp = dict(batch_size=128, model_type='deepconn', data_dir='data/Tiny/5_core/')
    data, y = synthetic_review_batch(300, 200, 500, 30, 20, seed=3)
    data_np, y_np = [d.numpy() for d in data], y.numpy()
    loader = data_fast.DataLoader.from_arrays(hp, data_np, y_np)
    assert loader.device.type == 'cuda'
    for epoch in range(2):
        at, sizes = 0, []
        for batch, yy in loader.iter():
            n = yy.shape[0]
            assert all(t.is_cuda and t.dtype == torch.int64 for t in batch) and yy.dtype == torch.float32
            burn = torch.empty(1 << 22, device=DEV).normal_()          # keep the compute stream busy
            for t, src in zip(batch, data_np):
                assert np.array_equal(t.cpu().numpy(), src[at:at + n])
            assert np.allclose(yy.cpu().numpy(), y_np[at:at + n])
            at += n
            sizes.append(n)
            del burn
        assert sizes == [128, 128, 44]


def test_neumf_init_and_three_stage_schedule(tmp_path):
    """NeuMF.init on the device == the reference's (fixture), and main.main_NeuMF runs its
    GMF -> MLP -> NeuMF schedule end to end on the HIP path (main.py:289-340)."""
    from reviews4rec_amd import main as M
    from reviews4rec_amd.pytorch_models.NeuMF import GMF, MLP, NeuMF
    g = Golden('neumf_full')
    hp = dict(g.hp)
    gmf, mlp, full = GMF(hp).to(DEV), MLP(hp).to(DEV), NeuMF(hp).to(DEV)
    gmf.load_state_dict(g.group('init_gmf'), strict=True)
    mlp.load_state_dict(g.group('init_mlp'), strict=True)
    full.init(gmf, mlp)
    for k, v in g.params().items():
        assert torch.equal(full.state_dict()[k].cpu(), v), k

    class Reader:
        def __len__(self):
            return 2

        def iter(self, eval=False):
            for k in (0, 1):
                yield g.batch(k, DEV)

    hp.update(epochs=2, dataset='golden', log_file=str(tmp_path / 'neumf.log'), model_path=str(tmp_path / 'neumf.pt'))
    metrics, ucm, icm = M.main_NeuMF(hp, (Reader(), Reader(), Reader()))
    assert 0.0 < metrics['MSE'] < 25.0 and sum(len(v) for v in ucm.values()) == 13 + 10
    for tag in ('_gmf', '_mlp', ''):                          # one best-on-validation checkpoint per stage
        assert (tmp_path / ('neumf.pt' + tag)).exists()
    log = open(tmp_path / 'neumf.log').read()
    assert log.count('end of epoch 2') == 3


@pytest.mark.parametrize('mt', ['bias_only', 'MF_dot', 'MF', 'deepconn', 'deepconn++', 'NARRE', 'transnet', 'transnet++'])
def test_main_pytorch_end_to_end_every_model_family(tmp_path, mt):
    """reviews4rec_amd.main.main_pytorch (main.py:342-399) on synthetic splits with the default engine
    choice ('auto': the native step where the family has one): three epochs, validation each epoch,
    best-checkpoint reload, test metrics -- MSE finite and better than predicting the global mean badly."""
    import numpy as np
    from reviews4rec_amd import data_fast, main as M
    torch.manual_seed(1234)          # xavier_init draws from the global stream: do not depend on the tests run before
    U, I, V, T, R, W, E, L = 40, 30, 300, 60, 4, 20, 16, 8
    hp = dict(model_type=mt, latent_size=L, word_embed_size=E, input_length=T, dropout=0.2, total_users=U, total_items=I,
              lr=0.01, weight_decay=1e-6, batch_size=32, epochs=3, dataset='synthetic', narre_num_reviews=R,
              narre_num_words=W, log_file=str(tmp_path / 'log.txt'), model_path=str(tmp_path / 'model.pt'), seed=5)
    rng = np.random.default_rng(3)
    hp['word_vectors'] = rng.uniform(-0.1, 0.1, size=(V, E)).astype(np.float32)

    def split(n, seed):
        narre = mt == 'NARRE'
        data, y = synthetic_review_batch(n, T, V, U, I, seed=seed, R=R if narre else None, W=W if narre else None)
        if narre:
            g = torch.Generator().manual_seed(seed)
            data[1] = torch.randint(0, U + 2, (n, R), generator=g)
            data[2] = torch.randint(0, I + 2, (n, R), generator=g)
        y = (3.0 + 0.5 * ((data[5] % 3).float() - 1.0) + 0.3 * ((data[6] % 2).float())).to(torch.float32)   # learnable signal
        return data_fast.DataLoader.from_arrays(hp, [d.numpy() for d in data], y.numpy())

    readers = (split(150, 1), split(70, 2), split(70, 3))
    metrics, by_user, by_item = M.main_pytorch(hp, readers, review_based_model=mt not in ('bias_only', 'MF_dot', 'MF'))
    # (TransNet's source network only learns the rating through source_fm: 15 steps leave its MSE near the
    # squared mean rating; its target network is the one that fits quickly)
    bound = 20.0 if mt.startswith('transnet') else 2.0
    assert np.isfinite(metrics['MSE']) and metrics['MSE'] < bound, metrics
    if mt.startswith('transnet'):
        assert metrics['MSE_right'] < 2.5, metrics
    log = open(hp['log_file']).read()
    assert 'end of epoch   3' in log or 'end of epoch 3' in log
