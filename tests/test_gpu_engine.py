"""The fused native DeepCoNN step (r4r_deepconn_step + flat Adam) against the
reference-generated golden trajectories, the autograd module path and the CPU
oracle (dropout masks drawn on the device and injected into the oracle)."""
import copy
import os

import numpy as np

import pytest
import torch

import oracle
from helpers import Golden, synthetic_review_batch
from test_gpu_models import build_model

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def make_engine(g, dropout=None, **kw):
    from reviews4rec_amd.engine import DeepCoNNEngine
    model, hp = build_model(g, dropout=dropout)
    return DeepCoNNEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], **kw), model, hp


ALGOS = [1, 2]      # R4R_CONV_DIRECT, R4R_CONV_PROJECT (include/r4r.h)


@pytest.mark.parametrize('algo', ALGOS)
@pytest.mark.parametrize('case', ['deepconn_e20', 'deepconn_e64'])
def test_engine_eval_matches_reference_golden(case, algo):
    g = Golden(case)
    eng, model, _ = make_engine(g, conv_algo=algo)
    model.eval()
    for k in (0, 1):
        data, y = g.batch(k, DEV)
        pred, se = eng.predict(data, y)
        ref = g.arr('eval%d' % k)
        torch.testing.assert_close(pred.cpu(), ref, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(se.cpu(), (ref - y.cpu()) ** 2, rtol=1e-4, atol=1e-5)
    pred, _ = eng.predict(g.neg_batch(DEV))
    assert tuple(pred.shape) == (3, 6)
    torch.testing.assert_close(pred.cpu(), g.arr('neg_eval'), rtol=1e-5, atol=1e-5)
    # the module path sees the same (re-homed) weights
    with torch.no_grad():
        torch.testing.assert_close(model(g.batch(0, DEV)[0]).cpu(), g.arr('eval0'), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('algo', ALGOS)
@pytest.mark.parametrize('case', ['deepconn_e20', 'deepconn_e64'])
def test_engine_training_trajectory_matches_reference_golden(case, algo):
    g = Golden(case)
    eng, model, hp = make_engine(g, conv_algo=algo)
    model.train()
    total = 0.0
    for step in range(3):
        data, y = g.batch(step % 2, DEV)
        se = eng.train_step(data, y)
        torch.testing.assert_close(se.cpu(), g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
        total += float(g.arr('se%d' % step).sum())
        if step == 0:
            got = eng.grads()
            ref_g = g.group('g0')
            assert set(got) == set(ref_g)                  # exactly the parameters the reference trains
            for k, v in ref_g.items():
                torch.testing.assert_close(got[k].cpu(), v, rtol=1e-4, atol=1e-6, msg=lambda m: k + ': ' + m)
        if step in (0, 2):
            sd = model.state_dict()
            for k, v in g.params('w%d' % (step + 1)).items():
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    m, v = eng.moments()
    for k, ref in g.group('m3').items():
        torch.testing.assert_close(m[k].cpu(), ref, rtol=1e-4, atol=1e-7, msg=lambda mm: k + ': ' + mm)
    for k, ref in g.group('v3').items():
        torch.testing.assert_close(v[k].cpu(), ref, rtol=1e-4, atol=1e-9, msg=lambda mm: k + ': ' + mm)
    torch.testing.assert_close(eng.sse.cpu()[0], torch.tensor(total), rtol=1e-5, atol=1e-4)   # running metric
    # untouched-by-this-mode parameters never moved
    sd = model.state_dict()
    for k in ('final.0.weight', 'final.3.bias', 'user_bias', 'item_bias'):
        assert torch.equal(sd[k].cpu(), g.params()[k])


def test_engine_dropout_masks_injected_into_oracle():
    g = Golden('deepconn_e20')
    eng, model, hp = make_engine(g, dropout=0.5, seed=7)
    model.train()
    data, y = g.batch(0, DEV)
    P = copy.deepcopy(g.params())
    se = eng.train_step(data, y)
    B, T, L = y.shape[0], data[3].shape[1], hp['latent_size']
    mult = eng.dropout_multipliers(B, T).cpu()
    vals = set(torch.unique(mult).tolist())
    assert vals <= {0.0, 2.0} and len(vals) == 2
    masks = {'user_conv.dropout': mult[:, :L], 'item_conv.dropout': mult[:, L:]}
    sse, grads = oracle.train_step(P, g.batch(0)[0], g.batch(0)[1], hp, oracle.AdamState(), masks=masks)
    torch.testing.assert_close(se.sum().cpu(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        if grads.get(k) is not None:
            solid = grads[k].abs() > 1e-6
            torch.testing.assert_close(sd[k].cpu()[solid], v[solid], rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    # a second step draws different masks (the Philox offset advanced)
    eng.train_step(data, y)
    assert not torch.equal(eng.dropout_multipliers(B, T).cpu(), mult)


@pytest.mark.parametrize('algo', ALGOS)
def test_engine_matches_module_path_at_baseline_shape(algo):
    """B=32 rows of the config-3 shape (T=1000, E=300): fused step == op-by-op autograd path."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import DeepCoNNEngine
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    B, T, E, V, U, I = 32, 1000, 300, 4000, 100, 50
    hp = dict(model_type='deepconn', latent_size=10, word_embed_size=E, input_length=T, dropout=0.0,
              total_users=U, total_items=I, lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=9)
    data, y = synthetic_review_batch(B, T, V, U, I, seed=21, device=DEV)

    def fresh():
        m = reviews4rec_amd.get_model_class('deepconn')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
        m.load_state_dict(P)
        return m.to(DEV).train()

    ref = fresh()
    opt = Adam(ref.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])
    eng_model = fresh()
    eng = DeepCoNNEngine(eng_model, lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=algo)
    for _ in range(2):
        ref.zero_grad()
        se_ref = MSELoss(hp)(ref(data), y, return_mean=False)
        torch.mean(se_ref).backward()
        opt.step()
        se = eng.train_step(data, y)
        torch.testing.assert_close(se, se_ref.detach(), rtol=1e-5, atol=1e-6)
    ref_g = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    for k, gv in eng.grads().items():
        torch.testing.assert_close(gv, ref_g[k], rtol=1e-4, atol=1e-7, msg=lambda m: k + ': ' + m)
    a, b = ref.state_dict(), eng_model.state_dict()
    for k in a:
        if k in ref_g:
            # two Adam steps: lr * m / (sqrt(v) + eps) amplifies 1e-9-level gradient differences
            # on elements whose gradient is ~1e-6; the gradients themselves are compared above
            # (atol: a gradient of 1e-5 known to the 1e-7 the comparison above allows is lr * 1 % = 2e-5 after one
            # step; one element of 51,530 sat at 2.5e-5 once the head summed its FC products in 8 parts)
            solid = ref_g[k].abs() > 1e-5
            torch.testing.assert_close(b[k][solid], a[k][solid], rtol=1e-5, atol=5e-5, msg=lambda m: k + ': ' + m)
            assert (b[k] - a[k]).abs().max() < 1e-3


@pytest.mark.parametrize('algo', ALGOS)
def test_engine_ragged_and_single_row_batches(algo):
    g = Golden('deepconn_e20')
    eng, model, _ = make_engine(g, conv_algo=algo)
    model.eval()
    data, y = g.batch(0, DEV)
    full, _ = eng.predict(data, y)
    for n in (1, 3):
        part, _ = eng.predict([d[:n] for d in data], y[:n])
        assert torch.equal(part, full[:n])


def test_projection_and_direct_conv_agree_on_argmax_and_pooled():
    """Full config-3 batch (B=128, T=1000, E=300, Zipf tokens, zero-padded tails): the two
    conv algorithms must give the same predictions to rounding and the same gradients."""
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import DeepCoNNEngine
    hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', dropout=0.0, vocab=20000)
    hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
    gen = synthetic.Generator(hp, seed=5)
    data, y = gen.batch(128)
    data = [torch.from_numpy(d).to(DEV) for d in data]
    y = torch.from_numpy(y).to(DEV)
    outs = {}
    for algo in ALGOS:
        torch.manual_seed(0)
        m = reviews4rec_amd.get_model_class('deepconn')(hp)
        from reviews4rec_amd.utils import xavier_init
        xavier_init(m)
        m = m.to(DEV).train()
        eng = DeepCoNNEngine(m, conv_algo=algo)
        se = eng.train_step(data, y).clone()
        outs[algo] = (se, {k: v.clone() for k, v in eng.grads().items()})
    torch.testing.assert_close(outs[1][0], outs[2][0], rtol=1e-5, atol=1e-6)
    for k in outs[1][1]:
        torch.testing.assert_close(outs[1][1][k], outs[2][1][k], rtol=1e-4, atol=1e-7, msg=lambda mm: k + ': ' + mm)


def test_full_size_batch_gradient_is_the_sum_of_its_shards_gradients():
    """BASELINE config 3 shape (B=128, T=1000, E=300): the fused step's gradient of the whole batch equals the sum of
    the gradients of its two halves computed with the same 1 / B_global loss scale -- the property data parallelism
    rests on, at full size (the small-shape DP tests pin it against the reference's trajectories)."""
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import DeepCoNNEngine
    from reviews4rec_amd.utils import xavier_init
    hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', dropout=0.0, vocab=20000)
    hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
    gen = synthetic.Generator(hp, seed=9)
    data, y = gen.batch(128)
    data = [torch.from_numpy(d).to(DEV) for d in data]
    y = torch.from_numpy(y).to(DEV)
    torch.manual_seed(0)
    m = reviews4rec_amd.get_model_class('deepconn')(hp)
    xavier_init(m)
    eng = DeepCoNNEngine(m.to(DEV).train(), conv_algo=2)

    def grad_of(lo, hi):
        d = [t[lo:hi].contiguous() for t in data]
        eng._launch(d, y[lo:hi].contiguous(), grad=True, training=True, inv_denom=1.0 / 128)   # gradients only: no Adam
        return eng.flat_g.clone()
    whole = grad_of(0, 128)
    parts = grad_of(0, 64) + grad_of(64, 128)
    assert float(whole.abs().max()) > 0
    torch.testing.assert_close(parts, whole, rtol=2e-4, atol=2e-7 * float(whole.abs().max()) + 1e-9)


@pytest.mark.parametrize('shape', [(64, 400, 128, 3000), (32, 200, 32, 300000)], ids=['v3k', 'v300k'])
@pytest.mark.parametrize('how', ['fused', 'side_stream'])
def test_token_prefetch_is_bit_identical_and_survives_mispredicted_batches(how, shape):
    """Token compaction of batch k+1 prepared during step k -- riding on step k's backward /
    reduce launches ('fused') or on a side stream -- must not change a single bit; a prepared
    batch that is never trained on (wrong guess, an eval in between) is discarded cleanly.
    (v300k: above 262,144 words the fused compaction takes eight int4 groups per thread, tokens_device.h.)"""
    import reviews4rec_amd
    from reviews4rec_amd.engine import DeepCoNNEngine
    (B, T, E, V), U, I = shape, 100, 50
    hp = dict(model_type='deepconn', latent_size=10, word_embed_size=E, input_length=T, dropout=0.0,
              total_users=U, total_items=I, lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=4)
    batches = [synthetic_review_batch(B, T, V, U, I, seed=30 + i, device=DEV) for i in range(5)]

    def fresh():
        m = reviews4rec_amd.get_model_class('deepconn')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
        m.load_state_dict(P)
        return DeepCoNNEngine(m.to(DEV).train(), lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=2)

    plain, pre = fresh(), fresh()
    order = [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 2, 2]
    for k, b in enumerate(order):
        data, y = batches[b]
        se_a = plain.train_step(data, y).clone()
        if k == 4:          # wrong guess: batch 3 is announced, batch 0 comes next
            nxt = batches[3][0]
        elif k + 1 < len(order):
            nxt = batches[order[k + 1]][0]
        else:
            nxt = None
        if how == 'fused':
            se_b = pre.train_step(data, y, next_data=nxt).clone()
        else:
            se_b = pre.train_step(data, y).clone()
            if nxt is not None:
                pre.prefetch_tokens(nxt)
        assert torch.equal(se_a, se_b), k
        if k == 7:          # an evaluation on another batch while a prepared state is pending
            pa, _ = plain.predict(batches[4][0], batches[4][1])
            pb, _ = pre.predict(batches[4][0], batches[4][1])
            assert torch.equal(pa, pb)
    torch.cuda.synchronize()
    assert torch.equal(plain.flat_p, pre.flat_p)
    assert torch.equal(plain.flat_m, pre.flat_m) and torch.equal(plain.flat_v, pre.flat_v)


def test_host_loop_with_native_engine_and_lookahead_matches_reference_metric():
    """main.train driving the native engine with the one-batch lookahead (the upcoming batch's
    tokens are prepared during the current step; the ragged last batch is not): the epoch metric
    round(sum SE / N, 4) (main.py:66) equals the reference's over the golden batches."""
    from reviews4rec_amd import main as M
    from reviews4rec_amd.loss import MSELoss
    g = Golden('deepconn_e20')
    eng, model, hp = make_engine(g, conv_algo=2)
    model.train()

    class Reader:
        def iter(self, eval=False):
            for k in (0, 1):
                yield g.batch(k, DEV)

    metrics = M.train(model, MSELoss(hp), None, Reader(), hp, engine=eng)
    n = g.arr('y0').shape[0] + g.arr('y1').shape[0]
    expected = round(float(g.arr('se0').sum() + g.arr('se1').sum()) / n, 4)
    assert metrics['MSE'] == pytest.approx(expected, abs=2e-4)
    # a second epoch over batches of equal shape exercises the prepared-token path end to end
    data, y = g.batch(0, DEV)

    class Same:
        def iter(self, eval=False):
            for _ in range(3):
                yield [d.clone() for d in data], y

    ref, _, _ = make_engine(g, conv_algo=2)
    ref.model.train()
    M.train(model, MSELoss(hp), None, Same(), hp, engine=eng)
    for k in (0, 1):
        ref.train_step(*g.batch(k, DEV))
    for _ in range(3):
        ref.train_step(data, y)
    torch.cuda.synchronize()
    assert torch.equal(ref.flat_p, eng.flat_p)


@pytest.mark.parametrize('case,engine_kind', [('deepconn_e20', 'native'), ('deepconn_e20', 'module'),
                                              ('mf_dot', 'native'), ('mf_dot', 'module'),
                                              ('narre_e16', 'auto'), ('deepconnpp_e20', 'graph'),
                                              ('transnetpp_e16', 'auto')])
def test_train_complete_resumes_exactly_from_its_epoch_checkpoint(tmp_path, engine_kind, case):
    """main.train_complete with hyper_params['checkpoint_path']: a run stopped after epoch 2 and
    started again lands on the weights of an uninterrupted 4-epoch run -- Adam moments, step
    counts and the dropout stream position travel with the checkpoint (dropout 0.5 here: a lost
    stream position would show up as a gross difference, not a rounding one)."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M, ops
    g = Golden(case)

    class Reader:
        def __len__(self):
            return 2

        def iter(self, eval=False):
            for k in (0, 1):
                yield g.batch(k, DEV)

    def run(tag, epochs, ckpt):
        ops.DropoutState.device_counter = None          # a fresh process would start without one
        ops.DropoutState.manual_seed(1234)
        model, hp = build_model(g, dropout=0.5)
        hp.update(engine=engine_kind, epochs=epochs, dataset='golden', log_file=str(tmp_path / (tag + '.log')),
                  model_path=str(tmp_path / (tag.rstrip('12') + '.pt')), seed=99)    # a resumed run keeps its model_path
        if ckpt:
            hp['checkpoint_path'] = str(tmp_path / 'resume.ckpt')
        Model = reviews4rec_amd.get_model_class(hp['model_type'])
        if engine_kind == 'native':                           # the fused step of this model family is in use
            from reviews4rec_amd.engine import DeepCoNNEngine, MFEngine
            assert isinstance(M.make_engine(hp, model), MFEngine if case == 'mf_dot' else DeepCoNNEngine)
        M.train_complete(hp, Model, Reader(), Reader(), {}, {}, model, review=case != 'mf_dot')
        ops.DropoutState.device_counter = None
        return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    whole = run('whole', 4, ckpt=False)
    run('part1', 2, ckpt=True)                       # stops after epoch 2, leaves resume.ckpt
    resumed = run('part2', 4, ckpt=True)             # fresh model object: everything comes from the file
    log2 = open(tmp_path / 'part2.log').read()
    assert 'Resuming after epoch 2' in log2 and 'end of epoch 3' in log2 and 'end of epoch 1' not in log2
    assert 'Resuming' not in open(tmp_path / 'whole.log').read()
    for k in whole:
        if engine_kind == 'native':                      # the fused steps sum in a fixed order: bit-identical
            assert torch.equal(whole[k], resumed[k]), k
        else:                                            # the op-by-op path's dense ID-table gradient is an
            torch.testing.assert_close(resumed[k], whole[k], rtol=1e-5, atol=1e-6,   # atomic scatter-add
                                       msg=lambda m: k + ': ' + m)


# ------------------------------------------------------------------------ MF_dot / bias_only native step
@pytest.mark.parametrize('case', ['mf_dot', 'mf_bias_only'])
def test_mf_engine_trajectory_matches_reference_golden(case):
    """r4r_mf_step (two launches, no dense table gradient) along the reference-generated 3-step
    trajectory: per-example SE, gradients of step 0, weights after 1 and 3 steps, Adam moments."""
    from reviews4rec_amd.engine import MFEngine
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    eng = MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    total = 0.0
    for step in range(3):
        data, y = g.batch(step % 2, DEV)
        se = eng.train_step(data, y).clone()
        torch.testing.assert_close(se.cpu(), g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
        total += float(g.arr('se%d' % step).sum())
        if step == 0:
            got, ref_g = eng.dense_grads(data), g.group('g0')
            assert set(got) == set(ref_g)
            for k, v in ref_g.items():
                torch.testing.assert_close(got[k].cpu(), v, rtol=1e-4, atol=1e-7, msg=lambda m: k + ': ' + m)
        if step in (0, 2):
            sd = model.state_dict()
            for k, v in g.params('w%d' % (step + 1)).items():
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    m, v = eng.moments()
    for k, ref in g.group('m3').items():
        torch.testing.assert_close(m[k].cpu(), ref, rtol=1e-4, atol=1e-7, msg=lambda mm: k + ': ' + mm)
    for k, ref in g.group('v3').items():
        torch.testing.assert_close(v[k].cpu(), ref, rtol=1e-4, atol=1e-10, msg=lambda mm: k + ': ' + mm)
    torch.testing.assert_close(eng.sse.cpu()[0], torch.tensor(total), rtol=1e-5, atol=1e-4)
    # eval through the engine == the reference's eval outputs at the initial weights
    model2, _ = build_model(g)
    eng2 = MFEngine(model2.eval())
    for k in (0, 1):
        data, y = g.batch(k, DEV)
        pred, se = eng2.predict(data, y)
        torch.testing.assert_close(pred.cpu(), g.arr('eval%d' % k), rtol=1e-5, atol=1e-5)
    pred, _ = eng2.predict(g.neg_batch(DEV))
    torch.testing.assert_close(pred.cpu(), g.arr('neg_eval'), rtol=1e-5, atol=1e-5)


def test_mf_engine_equals_module_path_with_duplicates_and_dropout():
    """Cardinalities where the sweep matters (20 k users, 5 k items, d = 64), a batch with many
    repeated ids, dropout 0.5: the device-drawn masks injected into the CPU oracle reproduce the
    step, and three engine steps equal three oracle steps."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import MFEngine
    U, I, D, B = 20000, 5000, 64, 128
    hp = dict(model_type='MF_dot', latent_size=D, dropout=0.5, total_users=U, total_items=I, lr=0.002,
              weight_decay=1e-6)
    P = oracle.init_params(hp, seed=3)
    model = reviews4rec_amd.get_model_class('MF_dot')(hp)
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    state = oracle.AdamState()
    rng = torch.Generator().manual_seed(5)
    for step in range(3):
        uid = torch.randint(0, 40, (B,), generator=rng)               # heavy repetition
        iid = torch.randint(0, I, (B,), generator=rng)
        iid[:16] = iid[0]
        y = torch.randint(1, 6, (B,), generator=rng).float()
        data = [None, None, None, None, None, uid, iid]
        se = eng.train_step([None] * 5 + [uid.to(DEV), iid.to(DEV)], y.to(DEV)).cpu().clone()
        mult = eng.dropout_multipliers(B).cpu()
        masks = {'dropout.user': mult[:, :D], 'dropout.item': mult[:, D:]}
        assert 0.3 < float((mult == 0).float().mean()) < 0.7
        sse, _ = oracle.train_step(P, data, y, hp, state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


@pytest.mark.parametrize('D,B', [(64, 5000), (32, 3000), (10, 2500), (64, 16384), (4, 2100), (8, 1000), (128, 1500),
                                 (256, 2100), (100, 3000), (64, 65536), (10, 40000)])     # (past 16,384: hyper_params.py:60 has no bound)
def test_mf_engine_large_batch_with_popular_rows(D, B):
    """Batches of thousands (SURVEY 8d quotes MF at B = 8,192): an item named by ~14 % of the
    ratings, a user by ~5 %, rows named once, twice, and ids whose first / last rating sit at the
    ends of the batch.  Wide (D = 64, 32) and generic (D = 10) forms of the entry waves against
    the CPU oracle for two steps, dropout masks injected; and the step is deterministic."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import MFEngine
    U, I = 30000, 9000
    hp = dict(model_type='MF_dot', latent_size=D, dropout=0.5, total_users=U, total_items=I, lr=0.002,
              weight_decay=1e-6)
    P0 = oracle.init_params(hp, seed=11)
    rng = torch.Generator().manual_seed(17)
    batches = []
    for step in range(2):
        uid = torch.randint(0, U, (B,), generator=rng)
        iid = torch.randint(0, I, (B,), generator=rng)
        iid[torch.rand(B, generator=rng) < 0.14] = 7
        uid[torch.rand(B, generator=rng) < 0.05] = 123
        uid[0] = uid[B - 1] = 4242                                        # first and last rating of the batch
        iid[1] = iid[B - 2] = 4243
        y = torch.randint(1, 6, (B,), generator=rng).float()
        batches.append((uid, iid, y))

    def run():
        model = reviews4rec_amd.get_model_class('MF_dot')(hp)
        model.load_state_dict(P0)
        model = model.to(DEV).train()
        eng = MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
        out = []
        for uid, iid, y in batches:
            se = eng.train_step([None] * 5 + [uid.to(DEV), iid.to(DEV)], y.to(DEV)).cpu().clone()
            out.append((se, eng.dropout_multipliers(B).cpu()))
        return {k: v.cpu().clone() for k, v in model.state_dict().items()}, out

    sd, out = run()
    P, state = {k: v.clone() for k, v in P0.items()}, oracle.AdamState()
    for (uid, iid, y), (se, mult) in zip(batches, out):
        masks = {'dropout.user': mult[:, :D], 'dropout.item': mult[:, D:]}
        sse, _ = oracle.train_step(P, [None] * 5 + [uid, iid], y, hp, state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
    for k, v in P.items():
        torch.testing.assert_close(sd[k], v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    sd2, _ = run()
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k


# --------------------------------------------------------------------------------- NARRE native step
def test_narre_engine_eval_matches_reference_golden():
    from reviews4rec_amd.engine import NarreEngine
    g = Golden('narre_e16')
    model, hp = build_model(g)
    eng = NarreEngine(model.eval())
    for k in (0, 1):
        data, y = g.batch(k, DEV)
        pred, se = eng.predict(data, y)
        torch.testing.assert_close(pred.cpu(), g.arr('eval%d' % k), rtol=1e-5, atol=1e-5)
    pred, _ = eng.predict(g.neg_batch(DEV))
    torch.testing.assert_close(pred.cpu(), g.arr('neg_eval'), rtol=1e-5, atol=1e-5)


def test_narre_engine_training_trajectory_matches_reference_golden():
    """r4r_narre_step along the reference-generated 3-step trajectory: per-example SE, every
    gradient of step 0 (ID tables / biases rebuilt from their compact rows), weights after 1 and
    3 steps, Adam moments, the running SE."""
    from test_oracle_golden import ill_conditioned
    from reviews4rec_amd.engine import NarreEngine
    g = Golden('narre_e16')
    model, hp = build_model(g)
    model.train()
    eng = NarreEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    total = 0.0
    for step in range(3):
        data, y = g.batch(step % 2, DEV)
        nxt = g.batch((step + 1) % 2, DEV)[0]                # shapes differ (ragged batch): the guess is declined
        se = eng.train_step(data, y, next_data=nxt).clone()
        torch.testing.assert_close(se.cpu(), g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
        total += float(g.arr('se%d' % step).sum())
        if step == 0:
            got, ref_g = eng.grads(data), g.group('g0')
            assert set(got) == set(ref_g)
            for k, v in ref_g.items():
                torch.testing.assert_close(got[k].cpu(), v, rtol=1e-4, atol=1e-7, msg=lambda m: k + ': ' + m)
        if step in (0, 2):
            sd = model.state_dict()
            for k, v in g.params('w%d' % (step + 1)).items():
                if ill_conditioned(k):
                    continue
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    m, v = eng.moments()
    for k, ref in g.group('m3').items():
        torch.testing.assert_close(m[k].cpu(), ref, rtol=1e-4, atol=1e-7, msg=lambda mm: k + ': ' + mm)
    for k, ref in g.group('v3').items():
        if not ill_conditioned(k):
            torch.testing.assert_close(v[k].cpu(), ref, rtol=1e-4, atol=1e-10, msg=lambda mm: k + ': ' + mm)
    torch.testing.assert_close(eng.sse.cpu()[0], torch.tensor(total), rtol=1e-5, atol=1e-4)


def test_narre_engine_dropout_masks_injected_into_oracle():
    """Dropout 0.5 on all seven sites: the multipliers the device drew, injected into the CPU
    oracle, reproduce the step's SE and its updated weights."""
    from reviews4rec_amd.engine import NarreEngine
    g = Golden('narre_e16')
    model, hp = build_model(g, dropout=0.5)
    model.train()
    eng = NarreEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    P = {k: v.clone() for k, v in g.params().items()}
    state = oracle.AdamState()
    L = hp['latent_size']
    for step in range(2):
        data, y = g.batch(0, DEV)
        se = eng.train_step(data, y).cpu().clone()
        B, R = data[5].numel(), data[3].shape[-2]
        mult = eng.dropout_multipliers(data).cpu()
        assert 0.3 < float((mult == 0).float().mean()) < 0.7
        RL = R * L
        masks = {'user_conv.dropout': mult[:, 0:RL].reshape(B * R, L),
                 'item_conv.dropout': mult[:, RL:2 * RL].reshape(B * R, L),
                 'attention_scorer_user.2': mult[:, 2 * RL:3 * RL].reshape(B, R, L),
                 'attention_scorer_item.2': mult[:, 3 * RL:4 * RL].reshape(B, R, L),
                 'dropout.user': mult[:, 4 * RL:4 * RL + L], 'dropout.item': mult[:, 4 * RL + L:4 * RL + 2 * L],
                 'final.0': mult[:, 4 * RL + 2 * L:]}
        cpu_data, cpu_y = g.batch(0)
        sse, _ = oracle.train_step(P, cpu_data, cpu_y, dict(hp), state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    from test_oracle_golden import ill_conditioned
    for k, v in P.items():
        if not ill_conditioned(k):
            # two Adam steps turn 1e-9-level gradient differences on near-zero gradients into
            # 1e-5-level weight differences (lr * m / sqrt(v)): rounding-level agreement on all but a
            # handful of elements, and nothing beyond a fraction of one lr-sized step anywhere
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3, k
            assert float(diff.max()) < 5e-4, k


def test_narre_engine_wide_latent_uses_the_general_instantiation():
    """latent_size 24 / 20 reviews (> 16: the <= 32 kernels) against the CPU oracle: two steps,
    SE and updated weights."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import NarreEngine
    from test_oracle_golden import ill_conditioned
    B, R, W, E, V, U, I, L = 6, 20, 14, 32, 300, 50, 40, 24
    hp = dict(model_type='NARRE', latent_size=L, word_embed_size=E, dropout=0.0, total_users=U, total_items=I,
              lr=0.002, weight_decay=1e-6, narre_num_reviews=R, narre_num_words=W)
    P = oracle.init_params(hp, vocab_size=V, seed=8)
    model = reviews4rec_amd.get_model_class('NARRE')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = NarreEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    state = oracle.AdamState()
    data, y = synthetic_review_batch(B, W, V, U, I, seed=2, R=R, W=W)
    data[1] = torch.randint(0, U + 2, (B, R))
    data[2] = torch.randint(0, I + 2, (B, R))
    for step in range(2):
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        sse, _ = oracle.train_step(P, data, y, hp, state)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        if not ill_conditioned(k):
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k


def test_narre_engine_popular_id_rows_in_the_fused_step():
    """The fused step's ID-table role (entry waves in pairs, narre_rows_block) with rows that hundreds of the
    batch's 1,056 compact entries hit -- several hits per lane, further rounds after the pair's common round
    trip -- and users / items rated more than once: two steps against the CPU oracle (dense Adam)."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import NarreEngine
    from test_oracle_golden import ill_conditioned
    B, R, W, E, V, U, I, L = 96, 10, 12, 32, 300, 60, 50, 10
    hp = dict(model_type='NARRE', latent_size=L, word_embed_size=E, dropout=0.0, total_users=U, total_items=I,
              lr=0.002, weight_decay=1e-6, narre_num_reviews=R, narre_num_words=W)
    P = oracle.init_params(hp, vocab_size=V, seed=12)
    model = reviews4rec_amd.get_model_class('NARRE')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = NarreEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    state = oracle.AdamState()
    gen = torch.Generator().manual_seed(5)
    data, y = synthetic_review_batch(B, W, V, U, I, seed=4, R=R, W=W)
    who = torch.randint(0, U + 2, (B, R), generator=gen)
    what = torch.randint(0, I + 2, (B, R), generator=gen)
    who[torch.rand((B, R), generator=gen) < 0.45] = 7        # ~430 of the 960 neighbour entries on one user row
    what[torch.rand((B, R), generator=gen) < 0.30] = 3
    data[1], data[2] = who, what
    data[5] = torch.randint(0, 12, (B,), generator=gen)      # a dozen users rate everything: self rows repeat
    data[6] = torch.randint(0, I, (B,), generator=gen)
    for step in range(2):
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        sse, _ = oracle.train_step(P, data, y, hp, state)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        if not ill_conditioned(k):
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k


def test_narre_engine_token_prefetch_is_bit_identical_with_wrong_guesses():
    """NARRE: the next batch's token state prepared on the current step's launches (project path)
    changes no bit; a wrong guess and an eval in between are handled."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import NarreEngine
    B, R, W, E, V, U, I, L = 16, 10, 60, 32, 900, 60, 40, 10
    hp = dict(model_type='NARRE', latent_size=L, word_embed_size=E, dropout=0.0, total_users=U, total_items=I,
              lr=0.002, weight_decay=1e-6, narre_num_reviews=R, narre_num_words=W)
    P = oracle.init_params(hp, vocab_size=V, seed=6)

    def batch(seed):
        data, y = synthetic_review_batch(B, W, V, U, I, seed=seed, R=R, W=W, device=DEV)
        g = torch.Generator().manual_seed(seed)
        data[1] = torch.randint(0, U + 2, (B, R), generator=g).to(DEV)
        data[2] = torch.randint(0, I + 2, (B, R), generator=g).to(DEV)
        return data, y
    batches = [batch(40 + k) for k in range(4)]

    def fresh():
        m = reviews4rec_amd.get_model_class('NARRE')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
        m.load_state_dict(P)
        return NarreEngine(m.to(DEV).train(), lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=2)
    plain, pre = fresh(), fresh()
    order = [0, 1, 2, 3, 0, 1, 2, 3, 1, 1]
    for k, b in enumerate(order):
        data, y = batches[b]
        se_a = plain.train_step(data, y).clone()
        nxt = batches[2][0] if k == 3 else (batches[order[k + 1]][0] if k + 1 < len(order) else None)   # k == 3: wrong
        se_b = pre.train_step(data, y, next_data=nxt).clone()
        assert torch.equal(se_a, se_b), k
        if k == 6:
            pa, _ = plain.predict(*batches[0])
            pb, _ = pre.predict(*batches[0])
            assert torch.equal(pa, pb)
    torch.cuda.synchronize()
    assert torch.equal(plain.flat_p, pre.flat_p)
    for a, c in zip(plain.rows, pre.rows):
        assert torch.equal(a, c)


@pytest.mark.parametrize('kind', ['deepconn', 'NARRE', 'MF_dot'])
def test_native_engines_long_run_stays_consistent_and_learns(kind):
    """300 steps over a cycle of 5 batches with dropout and next-batch announcements: the running
    state (token double-buffering, row tags, Philox offsets, Adam step counts) stays consistent --
    an engine fed the announcements ends bit-identical to one that is not -- nothing goes non-finite,
    and the training MSE of the memorisable cycle drops well below its starting value."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import DeepCoNNEngine, MFEngine, NarreEngine
    B, V, U, I, L = 32, 700, 50, 40, 8
    if kind == 'NARRE':
        R, W, E = 10, 30, 32
        hp = dict(model_type='NARRE', latent_size=L, word_embed_size=E, dropout=0.2, total_users=U, total_items=I,
                  lr=0.01, weight_decay=1e-6, narre_num_reviews=R, narre_num_words=W)
    elif kind == 'deepconn':
        T, E = 120, 128
        hp = dict(model_type='deepconn', latent_size=L, word_embed_size=E, input_length=T, dropout=0.2,
                  total_users=U, total_items=I, lr=0.01, weight_decay=1e-6)
    else:
        hp = dict(model_type='MF_dot', latent_size=16, dropout=0.2, total_users=U, total_items=I, lr=0.01,
                  weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=12)

    def batch(seed):
        if kind == 'NARRE':
            data, y = synthetic_review_batch(B, hp['narre_num_words'], V, U, I, seed=seed, R=hp['narre_num_reviews'],
                                             W=hp['narre_num_words'], device=DEV)
            g = torch.Generator().manual_seed(seed)
            data[1] = torch.randint(0, U + 2, (B, hp['narre_num_reviews']), generator=g).to(DEV)
            data[2] = torch.randint(0, I + 2, (B, hp['narre_num_reviews']), generator=g).to(DEV)
            return data, y
        return synthetic_review_batch(B, hp.get('input_length', 8), V, U, I, seed=seed, device=DEV)
    batches = [batch(70 + k) for k in range(5)]

    def fresh():
        extra = {'word_vectors': P['word2vec.weight'].numpy()} if 'word2vec.weight' in P else {}
        m = reviews4rec_amd.get_model_class(hp['model_type'])(dict(hp, **extra))
        m.load_state_dict(P)
        m = m.to(DEV).train()
        if kind == 'deepconn':
            return DeepCoNNEngine(m, lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=2, seed=5)
        if kind == 'NARRE':
            return NarreEngine(m, lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=2, seed=5)
        return MFEngine(m, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=5)
    plain, told = fresh(), fresh()
    first = last = None
    for k in range(300):
        data, y = batches[k % 5]
        se_a = plain.train_step(data, y)
        se_b = told.train_step(data, y, next_data=batches[(k + 1) % 5][0])
        if k % 50 == 49 or k < 5:
            assert torch.equal(se_a, se_b), k
            assert bool(torch.isfinite(se_a).all()), k
        if k < 5:
            first = (first or 0.0) + float(se_a.mean()) / 5
        if k >= 295:
            last = (last or 0.0) + float(se_a.mean()) / 5
    assert last < 0.5 * first, (first, last)
    sa, sb = plain.model.state_dict(), told.model.state_dict()
    for k_ in sa:
        assert torch.equal(sa[k_], sb[k_]), k_


# ------------------------------------------------------------------------------ DeepCoNN++ native step
def test_deepconnpp_engine_matches_reference_golden():
    """r4r_deepconnpp_step: eval outputs, then the reference-generated 3-step trajectory (SE, every
    gradient of step 0, weights after 1 and 3 steps, Adam moments); `fm`, unused in this mode, never moves."""
    from reviews4rec_amd.engine import DeepCoNNPPEngine
    g = Golden('deepconnpp_e20')
    model, hp = build_model(g)
    eng = DeepCoNNPPEngine(model.eval())
    for k in (0, 1):
        data, y = g.batch(k, DEV)
        pred, _ = eng.predict(data, y)
        torch.testing.assert_close(pred.cpu(), g.arr('eval%d' % k), rtol=1e-5, atol=1e-5)
    pred, _ = eng.predict(g.neg_batch(DEV))
    torch.testing.assert_close(pred.cpu(), g.arr('neg_eval'), rtol=1e-5, atol=1e-5)
    for algo in ALGOS:
        model, hp = build_model(g)
        model.train()
        eng = DeepCoNNPPEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=algo)
        total = 0.0
        for step in range(3):
            data, y = g.batch(step % 2, DEV)
            se = eng.train_step(data, y, next_data=g.batch((step + 1) % 2, DEV)[0]).clone()
            torch.testing.assert_close(se.cpu(), g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
            total += float(g.arr('se%d' % step).sum())
            if step == 0:
                got, ref_g = eng.grads(data), g.group('g0')
                assert set(got) == set(ref_g)
                for k, v in ref_g.items():
                    torch.testing.assert_close(got[k].cpu(), v, rtol=1e-4, atol=1e-7, msg=lambda m: k + ': ' + m)
            if step in (0, 2):
                sd = model.state_dict()
                for k, v in g.params('w%d' % (step + 1)).items():
                    torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
        m, v = eng.moments()
        for k, ref in g.group('m3').items():
            torch.testing.assert_close(m[k].cpu(), ref, rtol=1e-4, atol=1e-7, msg=lambda mm: k + ': ' + mm)
        for k, ref in g.group('v3').items():
            torch.testing.assert_close(v[k].cpu(), ref, rtol=1e-4, atol=1e-10, msg=lambda mm: k + ': ' + mm)
        torch.testing.assert_close(eng.sse.cpu()[0], torch.tensor(total), rtol=1e-5, atol=1e-4)


def test_deepconnpp_engine_dropout_masks_injected_into_oracle():
    from reviews4rec_amd.engine import DeepCoNNPPEngine
    g = Golden('deepconnpp_e20')
    model, hp = build_model(g, dropout=0.5)
    model.train()
    eng = DeepCoNNPPEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    P = {k: v.clone() for k, v in g.params().items()}
    state = oracle.AdamState()
    L = hp['latent_size']
    for step in range(2):
        data, y = g.batch(0, DEV)
        se = eng.train_step(data, y).cpu().clone()
        mult = eng.dropout_multipliers(data).cpu()
        assert 0.25 < float((mult == 0).float().mean()) < 0.75
        masks = {'user_conv.dropout': mult[:, :L], 'item_conv.dropout': mult[:, L:2 * L], 'final.2': mult[:, 2 * L:]}
        cpu_data, cpu_y = g.batch(0)
        sse, _ = oracle.train_step(P, cpu_data, cpu_y, dict(hp), state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        diff = (sd[k].cpu() - v).abs()
        assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k


# --------------------------------------------------------------------------------- TransNet native step
@pytest.mark.parametrize('case', ['transnet_e16', 'transnetpp_e16'])
@pytest.mark.parametrize('algo', ALGOS)
def test_transnet_engine_matches_reference_three_optimiser_trajectory(case, algo):
    """r4r_transnet_step (one backward, three disjoint parameter groups, one flat Adam) along the
    reference-generated trajectory of the three-optimiser step (main.py:35-53 under torch-0.4
    write-through semantics): per-example source SE, target / transform losses, weights after 1
    and 3 steps; then eval-mode predictions."""
    from reviews4rec_amd.engine import TransNetEngine
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    eng = TransNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=algo)
    for step in range(3):
        data, y = g.batch(step % 2, DEV)
        before = eng.sse.clone()
        se = eng.train_step(data, y).clone()
        torch.testing.assert_close(se.cpu(), g.arr('tn_se%d' % step), rtol=1e-4, atol=1e-5)
        aux = g.arr('tn_aux%d' % step)
        got = (eng.sse - before).cpu()
        torch.testing.assert_close(got[1:], aux, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(got[0], g.arr('tn_se%d' % step).sum(), rtol=1e-4, atol=1e-4)
        if step in (0, 2):
            sd = model.state_dict()
            for k, v in g.params('tn_w%d' % (step + 1)).items():
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


@pytest.mark.parametrize('case', ['transnet_e16', 'transnetpp_e16'])
def test_transnet_engine_dropout_masks_and_eval(case):
    """Dropout 0.5 at all seven sites: the device-drawn masks injected into the CPU oracle's
    three-optimiser step reproduce two engine steps; eval-mode predictions equal the module's."""
    from reviews4rec_amd.engine import TransNetEngine
    g = Golden(case)
    model, hp = build_model(g, dropout=0.5)
    P = {k: v.clone() for k, v in g.params().items()}
    model.train()
    eng = TransNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    L = int(hp['latent_size'])
    states = dict(source=oracle.AdamState(), source_fm=oracle.AdamState(), target=oracle.AdamState())
    for step in range(2):
        data, y = g.batch(step % 2, DEV)
        se = eng.train_step(data, y).cpu().clone()
        mult = eng.dropout_multipliers(data).cpu()
        assert 0.3 < float((mult[:, :5 * L] == 0).float().mean()) < 0.7
        masks = {'source.user_conv.dropout': mult[:, :L], 'source.item_conv.dropout': mult[:, L:2 * L],
                 'target.conv.dropout': mult[:, 2 * L:3 * L], 'source.dropout': mult[:, 3 * L:4 * L],
                 'target.dropout': mult[:, 4 * L:5 * L], 'dropout.user': mult[:, 5 * L:5 * L + 5],
                 'dropout.item': mult[:, 5 * L + 5:]}
        cdata, cy = g.batch(step % 2)
        ref_se, lt, ltr = oracle.transnet_train_step(P, cdata, cy, hp, states, masks=masks)
        torch.testing.assert_close(se, ref_se, rtol=1e-4, atol=1e-5)
        aux = eng.aux(data).cpu()
        torch.testing.assert_close(aux[:, 1].mean(), torch.tensor(lt), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(aux[:, 2].mean(), torch.tensor(ltr), rtol=1e-4, atol=1e-5)
    sd = model.state_dict()
    for k, v in P.items():
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    model.eval()
    data, y = g.batch(0, DEV)
    pred, se = eng.predict(data, y)
    with torch.no_grad():
        ref = model(data)
    torch.testing.assert_close(pred, ref[0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(eng.aux(data)[:, 0], ref[1].reshape(-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('case', ['transnet_e16', 'transnetpp_e16'])
def test_transnet_engine_through_the_host_loop(case):
    """reviews4rec_amd.main.train with the native TransNet step in place of the optimiser list:
    the three metrics of main.py:62-66 along the reference trajectory, and make_engine picks it."""
    from reviews4rec_amd import main as M
    from reviews4rec_amd.engine import TransNetEngine
    from reviews4rec_amd.loss import MSELoss
    g = Golden(case)
    model, hp = build_model(g)
    eng = M.make_engine(dict(hp, engine='auto'), model)
    assert isinstance(eng, TransNetEngine)
    for step in range(3):
        class OneBatch:
            def iter(self, eval=False):
                yield g.batch(step % 2, DEV)
        metrics = M.train(model, MSELoss(hp), None, OneBatch(), hp, engine=eng)
        assert metrics['MSE'] == pytest.approx(round(float(g.arr('tn_se%d' % step).mean()), 4), abs=2e-4)
        aux = g.arr('tn_aux%d' % step)
        assert metrics['MSE_target'] == pytest.approx(float(aux[0]), abs=2e-4)
        assert metrics['MSE_transform'] == pytest.approx(float(aux[1]), abs=2e-4)
    sd = model.state_dict()
    for k, v in g.params('tn_w3').items():
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


@pytest.mark.parametrize('T,L', [(200, 24), (1100, 10)], ids=['wide', 'tall'])
@pytest.mark.parametrize('mt', ['transnet', 'transnet++'])
def test_transnet_engine_wide_latent_and_larger_shapes(mt, T, L):
    """latent_size 24 (> 16: the <= 32 head instantiation), E = 64, T = 200 (project-then-gather by
    choice), duplicated ids -- and tall documents (T = 1100: nine 128-position segments, the head's pool
    finish takes a second round of tiles) -- against the CPU oracle's literal three-optimiser step: two steps."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import TransNetEngine
    from test_oracle_golden import ill_conditioned
    B, E, V, U, I = 12, 64, 400, 30, 20
    hp = dict(model_type=mt, latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U, total_items=I,
              lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=4)
    model = reviews4rec_amd.get_model_class(mt)(dict(hp, word_vectors=P['target.word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = TransNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=2)
    states = dict(source=oracle.AdamState(), source_fm=oracle.AdamState(), target=oracle.AdamState())
    data, y = synthetic_review_batch(B, T, V, U, I, seed=3)
    data[5][:4] = data[5][0]                                 # the same user four times, the same item three times
    data[6][5:8] = data[6][5]
    for step in range(2):
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        ref_se, lt, ltr = oracle.transnet_train_step(P, data, y, hp, states)
        torch.testing.assert_close(se, ref_se, rtol=1e-4, atol=1e-4)
        aux = eng.aux([d.to(DEV) for d in data]).cpu()
        torch.testing.assert_close(aux[:, 1].mean(), torch.tensor(lt), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(aux[:, 2].mean(), torch.tensor(ltr), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        if not ill_conditioned(k):
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k


def test_transnet_engine_token_prefetch_is_bit_identical_with_wrong_guesses():
    """TransNet++: the next batch's token state (three towers) prepared on the current step's
    launches changes no bit; a wrong guess and an eval in between are handled."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import TransNetEngine
    B, T, E, V, U, I, L = 16, 300, 32, 900, 60, 40, 10
    hp = dict(model_type='transnet++', latent_size=L, word_embed_size=E, input_length=T, dropout=0.5, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=6)
    batches = [synthetic_review_batch(B, T, V, U, I, seed=50 + k, device=DEV) for k in range(4)]

    def run(prefetch):
        model = reviews4rec_amd.get_model_class('transnet++')(dict(hp, word_vectors=P['target.word2vec.weight'].numpy()))
        model.load_state_dict(P)
        eng = TransNetEngine(model.to(DEV).train(), lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=2, seed=7)
        order = [0, 1, 2, 3, 0, 2]
        for n, k in enumerate(order):
            data, y = batches[k]
            guess = batches[order[n + 1] if n + 1 < len(order) and n != 2 else 1][0]    # step 2 announces the wrong batch
            eng.train_step(data, y, next_data=guess if prefetch else None)
            if n == 3:
                model.eval()
                eng.predict(batches[1][0], batches[1][1])
                model.train()
        return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('case', ['deepconn_e20', 'mf_dot', 'narre_e16', 'deepconnpp_e20', 'transnetpp_e16'])
def test_native_engines_accept_an_empty_and_a_one_row_batch(case):
    """Edge sizes through every native step: a batch of zero ratings (a rank's empty shard, an
    exhausted reader) is a no-op that changes no weight, a batch of one rating trains."""
    from reviews4rec_amd import main as M
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    eng = M.make_engine(dict(hp, engine='native'), model)
    assert eng is not None
    data, y = g.batch(0, DEV)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    empty = [None if d is None else d[:0] for d in data]
    se = eng.train_step(empty, y[:0])
    assert se.numel() == 0
    pred, _ = eng.predict(empty, y[:0])
    assert pred.numel() == 0
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    one = [None if d is None else d[:1] for d in data]
    se = eng.train_step(one, y[:1])
    assert se.shape == (1,) and bool(torch.isfinite(se).all())
    changed = [k for k, v in model.state_dict().items() if not torch.equal(v, before[k])]
    assert changed, 'a one-rating step must move the weights'


@pytest.mark.parametrize('mt,L', [('deepconn', 24), ('deepconn', 32), ('deepconn++', 24), ('deepconn++', 32)])
def test_deepconn_engines_wide_latent_use_the_general_instantiation(mt, L):
    """latent_size 24 and 32 (> 16: the <= 32 head instantiations; 32 is the hard limit) with tall
    documents (T = 1100: nine 128-position segments, past the head's one-round-trip pool-finish)
    against the CPU oracle: two steps, SE and updated weights."""
    from reviews4rec_amd import main as M
    from test_oracle_golden import ill_conditioned
    B, T, E, V, U, I = 10, 1100, 32, 500, 30, 20
    hp = dict(model_type=mt, latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U, total_items=I,
              lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, vocab_size=V, seed=9)
    import reviews4rec_amd
    model = reviews4rec_amd.get_model_class(mt)(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = M.make_engine(dict(hp, engine='native'), model)
    state = oracle.AdamState()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=12)
    for step in range(2):
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        sse, _ = oracle.train_step(P, data, y, hp, state)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        if not ill_conditioned(k):
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k


@pytest.mark.parametrize('case', ['deepconnpp_e20', 'transnet_e16', 'transnetpp_e16'])
def test_native_engines_eval_incl_negatives_shaped_batches(case):
    """predict() of the DeepCoNN++ / TransNet engines on the golden batches and on the negatives-shaped
    [B, 6, ...] batch of eval.py:64-92 (the engine folds the extra dim into the batch)."""
    from reviews4rec_amd import main as M
    g = Golden(case)
    model, hp = build_model(g)
    model.eval()
    eng = M.make_engine(dict(hp, engine='native'), model)
    tn = case.startswith('transnet')
    for k in (0, 1):
        data, y = g.batch(k, DEV)
        pred, se = eng.predict(data, y)
        torch.testing.assert_close(pred.cpu(), g.arr('eval%d/src' % k if tn else 'eval%d' % k), rtol=1e-5, atol=1e-5)
        if tn:                                               # target prediction and transform loss: the aux outputs
            aux = eng.aux(data).cpu()
            torch.testing.assert_close(aux[:, 0], g.arr('eval%d/tgt' % k).reshape(-1), rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(aux[:, 2].mean(), g.arr('eval%d/transform' % k).reshape(()), rtol=1e-5, atol=1e-5)
    pred, _ = eng.predict(g.neg_batch(DEV))
    ref = g.arr('neg_eval')
    assert tuple(pred.shape) == tuple(ref.shape) == (3, 6)
    torch.testing.assert_close(pred.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('B', [700, 5000])
def test_mf_engine_bias_only_large_batch(B):
    """bias_only (no tables: D = 0) at batches of hundreds / thousands with a popular item and user:
    the entry waves' bias-only form against the CPU oracle, two steps."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import MFEngine
    U, I = 3000, 900
    hp = dict(model_type='bias_only', latent_size=8, dropout=0.0, total_users=U, total_items=I, lr=0.002, weight_decay=1e-6)
    P = oracle.init_params(hp, seed=13)
    model = reviews4rec_amd.get_model_class('bias_only')(hp)
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    state = oracle.AdamState()
    rng = torch.Generator().manual_seed(19)
    for step in range(2):
        uid = torch.randint(0, U, (B,), generator=rng)
        iid = torch.randint(0, I, (B,), generator=rng)
        iid[torch.rand(B, generator=rng) < 0.2] = 5
        uid[torch.rand(B, generator=rng) < 0.1] = 77
        y = torch.randint(1, 6, (B,), generator=rng).float()
        se = eng.train_step([None] * 5 + [uid.to(DEV), iid.to(DEV)], y.to(DEV)).cpu().clone()
        sse, _ = oracle.train_step(P, [None] * 5 + [uid, iid], y, hp, state)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
    sd = model.state_dict()
    for k, v in P.items():
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def test_narre_rows_apply_thousands_of_gathered_entries():
    """r4r_narre_rows_apply on its own (the data-parallel ID-table update): 6,000 gathered entries per table
    (past the 4,096 the fused single-process role holds: the multi-word hit masks), popular rows, padding
    entries (-1), against a dense Adam step in plain torch; two steps."""
    import ctypes
    from reviews4rec_amd import _lib
    from reviews4rec_amd._lib import ptr
    lib = _lib.lib()
    U, I, L, n = 3000, 2000, 10, 6007                       # (not a multiple of the 16 entries an entry workgroup owns)
    B, R, T, E, V = 8, 4, 30, 16, 100                       # (only sizes the workspace that holds the row tags)
    gen = torch.Generator().manual_seed(23)
    tabs = [torch.randn(U, L, generator=gen), torch.randn(I, L, generator=gen), torch.randn(U, generator=gen),
            torch.randn(I, generator=gen)]
    P = [t.clone().to(DEV) for t in tabs]
    M = [torch.zeros_like(t) for t in P]
    Vv = [torch.zeros_like(t) for t in P]
    ws = torch.zeros(lib.r4r_narre_ws_bytes(B, R, T, E, L, V, U, I), dtype=torch.uint8, device=DEV)
    refP = [t.clone() for t in tabs]
    state = oracle.AdamState()
    lr, wd = 0.002, 1e-6
    p4 = lambda ts: (ctypes.c_uint64 * 4)(*[t.data_ptr() for t in ts])   # noqa: E731
    for step in (1, 2):
        gid = [torch.randint(0, U, (n,), generator=gen), torch.randint(0, I, (n,), generator=gen)]
        gid[0][torch.rand(n, generator=gen) < 0.1] = 17          # ~600 entries on one row
        gid[1][torch.rand(n, generator=gen) < 0.05] = 3
        pad = torch.rand(n, generator=gen) < 0.03
        gid[0][pad] = -1
        gid[1][pad] = -1
        grow = [torch.randn(n, L, generator=gen) * 0.1, torch.randn(n, L, generator=gen) * 0.1]
        g_entry = torch.where(torch.rand(n, generator=gen) < 0.2, torch.randn(n, generator=gen), torch.zeros(n))
        d = [x.to(DEV) for x in gid + grow + [g_entry]]
        rc = lib.r4r_narre_rows_apply(ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), n, p4(P), p4(M), p4(Vv), U, I,
                                      ptr(ws), ws.numel(), B, R, T, E, L, V, lr, 0.9, 0.999, 1e-8, wd, step,
                                      _lib.current_stream())
        _lib.check(rc, 'r4r_narre_rows_apply')
        ok = ~pad
        grads = {'0': torch.zeros(U, L).index_add_(0, gid[0][ok], grow[0][ok]),
                 '1': torch.zeros(I, L).index_add_(0, gid[1][ok], grow[1][ok]),
                 '2': torch.zeros(U).index_add_(0, gid[0][ok], g_entry[ok]),
                 '3': torch.zeros(I).index_add_(0, gid[1][ok], g_entry[ok])}
        params = {str(k): refP[k] for k in range(4)}
        oracle.adam_step(params, grads, state, lr, wd)
        refP = [params[str(k)] for k in range(4)]
    for k in range(4):
        torch.testing.assert_close(P[k].cpu(), refP[k], rtol=1e-5, atol=5e-6, msg=lambda m: 'table %d: %s' % (k, m))


@pytest.mark.parametrize('dist,expect', [('uniform', 'direct'), ('zipf', 'project')])
def test_conv_rule_measures_the_distinct_tokens_and_flips(dist, expect):
    """conv_algo = 0: the engine probes the batch's distinct-token count (left on the device by the gather
    kernel) at a fixed step and lets r4r_conv_pick choose.  Uniformly drawn full-length documents over a
    300k-word vocabulary (E = 64) hold ~208k distinct rows in 256k positions: the direct conv is faster and
    the rule must flip to it; Zipf documents with zero-padded tails stay on projection.  The count it read is
    exactly numpy's, and training through the flip lands where a projection-only run lands."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import DeepCoNNEngine
    from reviews4rec_amd import synthetic
    if os.environ.get('R4R_CONV_ALGO'):
        pytest.skip('R4R_CONV_ALGO pins the algorithm')
    V, E, B, T = 300000, 64, 128, 1000
    hp = dict(model_type='deepconn', latent_size=10, word_embed_size=E, input_length=T, dropout=0.0, total_users=500,
              total_items=300, lr=0.002, weight_decay=1e-6, vocab=V)
    table = synthetic.word_table(V, E)
    gen = synthetic.Generator(dict(hp), seed=11, doc_fill='full' if dist == 'uniform' else 'lognormal', token_dist=dist)
    batches = []
    for _ in range(3):
        d, y = gen.batch(B)
        batches.append(([torch.from_numpy(x).cuda() for x in d], torch.from_numpy(y).cuda(), d))

    def run(conv_algo):
        torch.manual_seed(3)
        model = reviews4rec_amd.get_model_class('deepconn')(dict(hp, word_vectors=table))
        from reviews4rec_amd.utils import xavier_init
        xavier_init(model)
        eng = DeepCoNNEngine(model.cuda().train(), lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=conv_algo)
        rows_seen = None
        for k in range(DeepCoNNEngine.PROBE_AT + 4):
            data, y, _ = batches[k % 3]
            eng.train_step(data, y, next_data=batches[(k + 1) % 3][0])
            if k == DeepCoNNEngine.PROBE_AT:
                rows_seen = (eng.conv_rows, k % 3)
        pred, _ = eng.predict(batches[0][0])
        return eng, rows_seen, pred.clone()

    auto, rows_seen, pred_auto = run(0)
    assert auto.conv_choice == expect
    d = batches[rows_seen[1]][2]
    assert rows_seen[0] == len(np.unique(d[3])) + len(np.unique(d[4]))
    pinned, _, pred_proj = run(2)
    assert pinned.conv_choice is None
    torch.testing.assert_close(pred_auto, pred_proj, rtol=2e-4, atol=2e-4)


# --------------------------------------------------------------------------------- MF / NeuMF native step
IDNET_CASES = ['mf_full', 'neumf_gmf', 'neumf_mlp', 'neumf_full']


@pytest.mark.parametrize('case', IDNET_CASES)
def test_idnet_engine_matches_reference_golden(case):
    """r4r_idnet_step (model_type 'MF' and NeuMF's GMF / MLP / NeuMF): eval outputs incl. the negatives
    shape, then the reference-generated 3-step trajectory -- SE, every gradient of step 0 (ID tables and
    bias vectors rebuilt from the compact rows), weights after 1 and 3 steps, Adam moments."""
    from reviews4rec_amd.engine import IdNetEngine
    g = Golden(case)
    model, hp = build_model(g)
    eng = IdNetEngine(model.eval())
    for k in (0, 1):
        data, y = g.batch(k, DEV)
        pred, se = eng.predict(data, y)
        torch.testing.assert_close(pred.cpu(), g.arr('eval%d' % k), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(se.cpu(), (g.arr('eval%d' % k) - y.cpu()) ** 2, rtol=1e-4, atol=1e-5)
    pred, _ = eng.predict(g.neg_batch(DEV))
    assert tuple(pred.shape) == tuple(g.arr('neg_eval').shape)
    torch.testing.assert_close(pred.cpu(), g.arr('neg_eval'), rtol=1e-5, atol=1e-5)

    model, hp = build_model(g)
    model.train()
    eng = IdNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    total = 0.0
    for step in range(3):
        data, y = g.batch(step % 2, DEV)
        se = eng.train_step(data, y).clone()
        torch.testing.assert_close(se.cpu(), g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
        total += float(g.arr('se%d' % step).sum())
        if step == 0:
            got, ref_g = eng.grads(data), g.group('g0')
            assert set(got) == set(ref_g)
            for k, v in ref_g.items():
                torch.testing.assert_close(got[k].cpu(), v, rtol=1e-4, atol=1e-7, msg=lambda m: k + ': ' + m)
        if step in (0, 2):
            sd = model.state_dict()
            for k, v in g.params('w%d' % (step + 1)).items():
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    m, v = eng.moments()
    for k, ref in g.group('m3').items():
        torch.testing.assert_close(m[k].cpu(), ref, rtol=1e-4, atol=1e-7, msg=lambda mm: k + ': ' + mm)
    for k, ref in g.group('v3').items():
        torch.testing.assert_close(v[k].cpu(), ref, rtol=1e-4, atol=1e-10, msg=lambda mm: k + ': ' + mm)
    torch.testing.assert_close(eng.sse.cpu()[0], torch.tensor(total), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('case', IDNET_CASES)
def test_idnet_engine_dropout_masks_injected_into_oracle(case):
    """Train-mode parity: the masks the device drew (Philox) are injected into the CPU oracle at every dropout
    site (the gathered rows of each table pair, the projection's input)."""
    from reviews4rec_amd.engine import IdNetEngine
    g = Golden(case)
    model, hp = build_model(g, dropout=0.5)
    model.train()
    eng = IdNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    P = {k: v.clone() for k, v in g.params().items()}
    state = oracle.AdamState()
    L = hp['latent_size']
    for step in range(2):
        data, y = g.batch(0, DEV)
        B = y.numel()
        se = eng.train_step(data, y).cpu().clone()
        mult = eng.dropout_multipliers(B).cpu()
        assert 0.25 < float((mult == 0).float().mean()) < 0.75
        if eng.kind == 'NeuMF':
            masks = {'dropout.gmf_user': mult[:, :L], 'dropout.gmf_item': mult[:, L:2 * L],
                     'dropout.mlp_user': mult[:, 2 * L:3 * L], 'dropout.mlp_item': mult[:, 3 * L:4 * L],
                     'project.0': mult[:, 4 * L:]}
        else:
            masks = {'dropout.user': mult[:, :L], 'dropout.item': mult[:, L:2 * L]}
            if eng.kind != 'GMF':
                masks['projection.0' if eng.kind == 'MF' else 'project.0'] = mult[:, 2 * L:]
        cpu_data, cpu_y = g.batch(0)
        sse, _ = oracle.train_step(P, cpu_data, cpu_y, dict(hp), state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        diff = (sd[k].cpu() - v).abs()
        assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k


@pytest.mark.parametrize('kind,L,B', [('MF', 32, 3000), ('NeuMF', 10, 2500), ('MLP', 24, 700), ('GMF', 5, 16384), ('GMF', 5, 32768),
                                      ('MF', 16, 20000), ('MF', 64, 1200), ('NeuMF', 48, 900), ('MLP', 64, 500)])
def test_idnet_engine_large_batches_with_popular_rows(kind, L, B):
    """Wide / odd latent sizes and batches where one item collects 14 % of the ratings and one user 5 %: the
    step against the oracle (dropout masks injected), run-to-run bit equality."""
    import reviews4rec_amd
    from reviews4rec_amd.engine import IdNetEngine
    U, I = 4000, 900
    hp = dict(model_type='MF' if kind == 'MF' else 'NeuMF', latent_size=L, dropout=0.3, total_users=U, total_items=I,
              lr=0.002, weight_decay=1e-6, word_embed_size=16, input_length=10)
    if kind != 'MF':
        hp['neumf_stage'] = kind
    P = oracle.init_params(hp, vocab_size=None, seed=3)
    gen = torch.Generator().manual_seed(B + L)
    uid = torch.randint(0, U, (B,), generator=gen)
    iid = torch.randint(0, I, (B,), generator=gen)
    iid[torch.rand(B, generator=gen) < 0.14] = 7
    uid[torch.rand(B, generator=gen) < 0.05] = 11
    uid[0], iid[-1] = U, I                                    # the last row of each table
    y = torch.randint(1, 6, (B,), generator=gen).float()
    data = [None, None, None, None, None, uid, iid]

    def run():
        model = reviews4rec_amd.get_model_class(hp['model_type'])(hp)
        model.load_state_dict(P)
        model = model.to(DEV).train()
        eng = IdNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=9)
        dev = [None] * 5 + [uid.to(DEV), iid.to(DEV)]
        ses, mults = [], []
        for _ in range(2):
            ses.append(eng.train_step(dev, y.to(DEV)).clone())
            mults.append(eng.dropout_multipliers(B).cpu())
        return model, eng, ses, mults

    model, eng, ses, mults = run()
    model2, eng2, ses2, _ = run()
    sd, sd2 = model.state_dict(), model2.state_dict()
    assert all(torch.equal(sd[k], sd2[k]) for k in sd) and torch.equal(ses[1], ses2[1])      # deterministic
    ref = {k: v.clone() for k, v in P.items()}
    state = oracle.AdamState()
    for step in range(2):
        mult = mults[step]
        if kind == 'NeuMF':
            masks = {'dropout.gmf_user': mult[:, :L], 'dropout.gmf_item': mult[:, L:2 * L],
                     'dropout.mlp_user': mult[:, 2 * L:3 * L], 'dropout.mlp_item': mult[:, 3 * L:4 * L],
                     'project.0': mult[:, 4 * L:]}
        else:
            masks = {'dropout.user': mult[:, :L], 'dropout.item': mult[:, L:2 * L]}
            if kind != 'GMF':
                masks['projection.0' if kind == 'MF' else 'project.0'] = mult[:, 2 * L:]
        sse, _ = oracle.train_step(ref, data, y, dict(hp), state, masks=masks)
        torch.testing.assert_close(ses[step].sum().cpu(), torch.tensor(sse), rtol=2e-4, atol=1e-2)
    for k, v in ref.items():
        diff = (sd[k].cpu() - v).abs()
        # Adam's first steps are lr * g / (|g| + eps): weights whose gradient is ~1e-8 amplify rounding;
        # bound the outliers by one lr step
        assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 5e-3 and float(diff.max()) < 2.1 * hp['lr'], k


@pytest.mark.parametrize('case', ['deepconn_e20', 'deepconn_e64'])
def test_engine_trajectory_under_the_fp16_split_gemm(case, monkeypatch):
    """R4R_GEMM_MATH=f16x2 (opt-in): the DeepCoNN step with the fp16-split projection GEMM follows the
    reference-generated trajectory at the SAME tolerances as the fp32 GEMM -- SE, weights after 1 and 3 steps."""
    from reviews4rec_amd import _lib
    from reviews4rec_amd.engine import DeepCoNNEngine
    monkeypatch.setenv('R4R_GEMM_MATH', 'f16x2')
    g = Golden(case)
    try:
        model, hp = build_model(g)
        model.train()
        eng = DeepCoNNEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], conv_algo=2)
        assert eng.gemm_math == 'f16x2'
        for step in range(3):
            data, y = g.batch(step % 2, DEV)
            se = eng.train_step(data, y).clone()
            torch.testing.assert_close(se.cpu(), g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
            if step in (0, 2):
                sd = model.state_dict()
                for k, v in g.params('w%d' % (step + 1)).items():
                    torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)
    finally:
        _lib.lib().r4r_gemm_math(0, 0.0, 0.0)


@pytest.mark.parametrize('n,W,bias', [(50021, 10, True), (20000, 64, True), (3001, 5, False), (70000, 33, True)])
def test_rows_apply_large_against_a_dense_adam_step(n, W, bias):
    """r4r_rows_apply_large (csrc/rows_large.hip: any number of compact entries -- the path behind the fused steps'
    entry-count caps): tens of thousands of entries over a 5,000-row table, one row named by a third of them (NARRE's
    padding sentinel, data.py:275-276), padding entries (-1), rows wider than one column pass -- against a dense Adam
    step in plain torch on the NAMED rows (the others must keep their bits: they are the caller's sweep); two steps;
    and the same call from the same state twice gives the same bits (no floating-point atomics, no
    scheduling-dependent order: replicas stay identical)."""
    import ctypes
    from reviews4rec_amd import _lib
    from reviews4rec_amd._lib import ptr
    lib = _lib.lib()
    rows = 5000
    gen = torch.Generator().manual_seed(n + W)
    tab, bvec = torch.randn(rows, W, generator=gen), torch.randn(rows, generator=gen)
    P, Pb = tab.clone().to(DEV), bvec.clone().to(DEV)
    M, V, Mb, Vb = torch.zeros_like(P), torch.zeros_like(P), torch.zeros_like(Pb), torch.zeros_like(Pb)
    scratch = torch.empty(lib.r4r_rows_large_ws_bytes(n), dtype=torch.uint8, device=DEV)
    refP, refB = tab.clone(), bvec.clone()
    refM, refV, refMb, refVb = torch.zeros(rows, W), torch.zeros(rows, W), torch.zeros(rows), torch.zeros(rows)
    lr, wd, b1, b2, eps = 0.002, 1e-6, 0.9, 0.999, 1e-8
    null = ctypes.c_void_p(None)

    def call(ids, g, gb, step, p, m, v, pb, mb, vb):
        rc = lib.r4r_rows_apply_large(ptr(ids), ptr(g), ptr(gb) if bias else null, n, W, ptr(p), ptr(m), ptr(v),
                                      ptr(pb) if bias else null, ptr(mb) if bias else null, ptr(vb) if bias else null,
                                      rows, ptr(scratch), scratch.numel(), lr, b1, b2, eps, wd, step, _lib.current_stream())
        _lib.check(rc, 'r4r_rows_apply_large')

    for step in (1, 2):
        ids = torch.randint(0, rows, (n,), generator=gen)
        ids[torch.rand(n, generator=gen) < 0.3] = rows - 1        # the sentinel row: a third of all entries
        ids[torch.rand(n, generator=gen) < 0.05] = 17
        pad = torch.rand(n, generator=gen) < 0.03
        ids[pad] = -1
        g = torch.randn(n, W, generator=gen) * 0.1
        gb = torch.where(torch.rand(n, generator=gen) < 0.2, torch.randn(n, generator=gen), torch.zeros(n))
        d_ids, d_g, d_gb = ids.to(DEV), g.to(DEV), gb.to(DEV)
        if step == 2:                                        # determinism: the same call on a copy of the state
            twin = [t.clone() for t in (P, M, V, Pb, Mb, Vb)]
            call(d_ids, d_g, d_gb, step, *twin)
        call(d_ids, d_g, d_gb, step, P, M, V, Pb, Mb, Vb)
        if step == 2:
            for mine, other in zip((P, M, V, Pb, Mb, Vb), twin):
                assert torch.equal(mine, other)
        ok = ~pad
        named = ids[ok].unique()
        G = torch.zeros(rows, W, dtype=torch.float64).index_add_(0, ids[ok], g[ok].double()).float()
        Gb = torch.zeros(rows, dtype=torch.float64).index_add_(0, ids[ok], gb[ok].double()).float()

        def adam(p, m, v, grad):                             # torch.optim.Adam's update of the named rows
            grad = grad + wd * p
            m.mul_(b1).add_(grad, alpha=1 - b1)
            v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
            p.addcdiv_(m / (1 - b1 ** step), v.sqrt() / (1 - b2 ** step) ** 0.5 + eps, value=-lr)

        for (p, m, v, grad) in ((refP, refM, refV, G),) + (((refB, refMb, refVb, Gb),) if bias else ()):
            pn, mn, vn = p[named].clone(), m[named].clone(), v[named].clone()
            adam(pn, mn, vn, grad[named])
            p[named], m[named], v[named] = pn, mn, vn
    torch.testing.assert_close(P.cpu(), refP, rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(M.cpu(), refM, rtol=1e-4, atol=1e-6)
    if bias:
        torch.testing.assert_close(Pb.cpu(), refB, rtol=1e-5, atol=5e-6)
    else:
        assert torch.equal(Pb.cpu(), bvec)
    untouched = torch.ones(rows, dtype=torch.bool)
    untouched[named] = False                                  # (rows neither step named keep their bits)


@pytest.mark.parametrize('B', [512, 2048])
def test_narre_engine_beyond_the_fused_entry_caps_against_the_oracle(B):
    """NARRE steps with more ID entries per table than the fused launch's entry waves hold (4,096): B = 512 x (1 + 10)
    = 5,632 runs as gradients -> flat Adam -> r4r_narre_rows_apply; B = 2,048 -> 22,528 entries, past that launch's
    16,384 too: r4r_narre_rows_apply_large.  One process, two training steps, dropout 0, every parameter against
    the oracle (hyper_params.py:60 puts no bound on batch_size)."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from test_oracle_golden import ill_conditioned
    T, E, V, U, I, L, R, W = 20, 16, 400, 3000, 2500, 8, 10, 20
    hp = dict(model_type='NARRE', latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, narre_num_reviews=R, narre_num_words=W, batch_size=B)
    assert M.native_step_limits(hp) is None
    P = oracle.init_params(hp, vocab_size=V, seed=57)
    model = reviews4rec_amd.get_model_class('NARRE')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = M.make_engine(dict(hp, engine='native'), model)
    assert eng is not None
    state = oracle.AdamState()
    for step in range(2):
        data, y = synthetic_review_batch(B, W, V, U, I, seed=70 + step, R=R, W=W)
        gen = torch.Generator().manual_seed(step)
        data[1] = torch.randint(0, U + 2, (B, R), generator=gen)
        data[2] = torch.randint(0, I + 2, (B, R), generator=gen)
        data[1][torch.rand(B, R, generator=gen) < 0.3] = U + 1       # the padding sentinel (data.py:275-276)
        data[2][torch.rand(B, R, generator=gen) < 0.3] = I + 1
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        sse, grads = oracle.train_step(P, data, y, hp, state)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
    sd = model.state_dict()
    for k, v in P.items():
        if ill_conditioned(k):
            continue
        diff = (sd[k].cpu() - v).abs()
        assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3, k
        assert float(diff.max()) < 2.5e-3, k


def _engine_of(kind):
    from reviews4rec_amd import engine as E
    case, cls = {'mf': ('mf_dot', E.MFEngine), 'idnet': ('neumf_full', E.IdNetEngine),
                 'narre': ('narre_e16', E.NarreEngine), 'transnetpp': ('transnetpp_e16', E.TransNetEngine)}[kind]
    g = Golden(case)
    model, hp = build_model(g)
    return g, model.train(), cls(model, lr=hp['lr'], weight_decay=hp['weight_decay'])


@pytest.mark.parametrize('kind', ['mf', 'idnet', 'narre', 'transnetpp'])
def test_rejected_checkpoint_and_failed_step_leave_the_engine_as_it_was(kind):
    """(1) a training step that RAISES before anything is launched must not consume its step number -- the
    blocked sweeps count pending gradient-zero updates from it; (2) a checkpoint that does not fit is rejected BEFORE
    anything is written -- weights, moments, step count and the sweep schedule stay, and training goes on as if nothing
    had happened; (3) a checkpoint written by name survives a round trip bit for bit."""
    import copy
    g, model, eng = _engine_of(kind)
    defer = dict(defer_sweep=True) if getattr(eng, 'TEMPORAL_SWEEP', False) else {}
    for step in range(3):
        data, y = g.batch(step % 2, DEV)
        eng.train_step(data, y, **defer)
    good = copy.deepcopy(eng.state_dict())
    assert isinstance(good['exp_avg'], dict) and good['step'] == 3            # moments BY NAME, every engine
    before = (eng.step_count, getattr(eng, '_tb_base', None), getattr(eng, '_tb_period', None))
    # (1) a step that raises on the host: a rating tensor of the wrong length
    data, y = g.batch(0, DEV)
    with pytest.raises(Exception):
        eng.train_step(data, None, **defer)                                    # y = None: fails before any launch
    assert (eng.step_count, getattr(eng, '_tb_base', None), getattr(eng, '_tb_period', None)) == before
    # (2) a checkpoint with one moment of the wrong size, and one with a missing name
    bad = copy.deepcopy(good)
    k0 = sorted(bad['exp_avg'])[0]
    bad['exp_avg'][k0] = torch.zeros(bad['exp_avg'][k0].numel() + 1)
    with pytest.raises(ValueError):
        eng.load_state_dict(bad)
    bad = copy.deepcopy(good)
    del bad['exp_avg_sq'][sorted(bad['exp_avg_sq'])[-1]]
    with pytest.raises(KeyError):
        eng.load_state_dict(bad)
    assert (eng.step_count, getattr(eng, '_tb_base', None), getattr(eng, '_tb_period', None)) == before
    now = eng.state_dict()
    for k, v in good['exp_avg'].items():
        assert torch.equal(now['exp_avg'][k], v) and torch.equal(now['exp_avg_sq'][k], good['exp_avg_sq'][k]), k
    # training goes on: a reference engine that never saw the bad checkpoints takes the same next step
    g2, model2, ref = _engine_of(kind)
    for step in range(3):
        data, y = g2.batch(step % 2, DEV)
        ref.train_step(data, y, **defer)
    data, y = g.batch(1, DEV)
    se, se_ref = eng.train_step(data, y, **defer).clone(), ref.train_step(data, y, **defer).clone()
    assert torch.equal(se, se_ref)
    sd, sd_ref = model.state_dict(), model2.state_dict()
    for k in sd:
        assert torch.equal(sd[k], sd_ref[k]), k
    # (3) the round trip
    eng.load_state_dict(copy.deepcopy(good))
    again = eng.state_dict()
    assert again['step'] == 3
    for k, v in good['exp_avg'].items():
        assert torch.equal(again['exp_avg'][k], v), k


@pytest.mark.parametrize('kind', ['mf', 'idnet', 'transnetpp'])
def test_loading_a_checkpoint_over_pending_sweep_updates_leaves_the_checkpoint_s_tables(kind):
    """main.py:306-308's order -- model.load_state_dict, then engine.load_state_dict -- with updates of the
    temporally blocked sweep still pending in the engine being overwritten: they belong to the state that is replaced
    (old moments, old step numbers) and must be dropped with it, not applied to the freshly loaded rows."""
    import copy
    g, model, eng = _engine_of(kind)
    defer = dict(defer_sweep=True)
    for step in range(3):
        eng.train_step(*g.batch(step % 2, DEV), **defer)
    ck_model = {k: v.detach().clone() for k, v in model.state_dict().items()}     # (flushes through the hook)
    ck_opt = copy.deepcopy(eng.state_dict())
    for step in range(5):
        eng.train_step(*g.batch(step % 2, DEV), **defer)
    assert eng._tb_period > 1 and eng._tb_base < eng.step_count                   # updates are pending right now
    model.load_state_dict(ck_model)
    eng.load_state_dict(ck_opt)
    assert eng.step_count == 3 and eng._tb_base == 3
    torch.cuda.synchronize()
    now = model.state_dict()
    for k, v in ck_model.items():
        assert torch.equal(now[k], v), k
    again = eng.state_dict()
    for k, v in ck_opt['exp_avg'].items():
        assert torch.equal(again['exp_avg'][k], v) and torch.equal(again['exp_avg_sq'][k], ck_opt['exp_avg_sq'][k]), k
    # ... and the resumed engine continues like one that stopped there
    g2, model2, ref = _engine_of(kind)
    for step in range(3):
        ref.train_step(*g2.batch(step % 2, DEV), **defer)
    a = eng.train_step(*g.batch(1, DEV), **defer).clone()
    b = ref.train_step(*g2.batch(1, DEV), **defer).clone()
    assert torch.equal(a, b)
    sd, sd_ref = model.state_dict(), model2.state_dict()
    for k in sd:
        assert torch.equal(sd[k], sd_ref[k]), k
