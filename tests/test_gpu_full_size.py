"""Parity at the sizes bench.py measures (BASELINE.json configs 2 and 5): the tagged dense-Adam sweeps of
r4r_mf_step and r4r_transnet_step over the FULL ID tables -- 192,403 x 64 / 63,001 x 64 (16.6 M parameters)
and 10 M x 5 / 1 M x 5 (55 M parameters) next to a 1 M-word table -- against the CPU oracle's dense Adam
(MF.py:52-58, TransNet.py:74-77,107-110, main.py:94-96: torch.optim.Adam updates every row, touched or not).
Every element of every table is compared; the oracle takes seconds at these sizes."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('B', [128, 8192])
def test_mf_step_at_cfg2_cardinalities(B):
    """cfg2 (MF_dot, Electronics: 192,403 users / 63,001 items, D = 64) at SURVEY 8d's two batch sizes, Zipf ids
    from the bench's generator, dropout masks drawn on the device and injected into the oracle, two steps:
    per-step SSE, then all 16.6 M parameters."""
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import MFEngine
    hp = synthetic.hyper_params_for('cfg2_mfdot_electronics', dropout=0.5)
    D = hp['latent_size']
    P = oracle.init_params(hp, seed=23)
    model = reviews4rec_amd.get_model_class('MF_dot')(hp)
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    gen = synthetic.Generator(hp, seed=3)
    state = oracle.AdamState()
    for step in range(2):
        data, y = gen.batch(B)
        uid, iid, y = torch.from_numpy(data[5]), torch.from_numpy(data[6]), torch.from_numpy(y)
        uid[0] = uid[B - 1] = hp['total_users'] - 1                   # the table's last row, at both ends of the batch
        iid[1] = iid[B - 2] = hp['total_items'] - 1
        se = eng.train_step([None] * 5 + [uid.to(DEV), iid.to(DEV)], y.to(DEV)).cpu().clone()
        mult = eng.dropout_multipliers(B).cpu()
        masks = {'dropout.user': mult[:, :D], 'dropout.item': mult[:, D:]}
        sse, _ = oracle.train_step(P, [None] * 5 + [uid, iid], y, hp, state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
    sd = model.state_dict()
    assert sd['user_embedding.weight'].shape == (hp['total_users'] + 1, D)          # MF.py:21
    for k, v in P.items():
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def test_scheduled_sweep_at_cfg2_cardinalities_against_the_oracle():
    """The scheduled temporally blocked sweep DIRECTLY against the oracle's dense Adam (not through the plain engine):
    cfg2's tables, B = 128, eleven steps with defer_sweep=True at period 4 -- chunks are visited on the schedule, the rows
    later batches name catch up in the forward and in their entry waves (a row is named again three and seven steps
    after its first touch) -- per-step SSE with the device's dropout masks injected, then all 16.6 M parameters."""
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import MFEngine
    hp = dict(synthetic.hyper_params_for('cfg2_mfdot_electronics', dropout=0.5), sweep_period=4)
    D, B = hp['latent_size'], 128
    P = oracle.init_params(hp, seed=29)
    model = reviews4rec_amd.get_model_class('MF_dot')(hp)
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    if eng.sweep_period != 4:
        pytest.skip('R4R_SWEEP_PERIOD overrides the period this test is written for')
    gen = synthetic.Generator(hp, seed=7)
    state = oracle.AdamState()
    for step in range(11):
        data, y = gen.batch(B)
        uid, iid, y = torch.from_numpy(data[5]), torch.from_numpy(data[6]), torch.from_numpy(y)
        if step in (0, 3, 7):
            uid[5], iid[9] = 12345, hp['total_items'] - 1              # the same rows again, several steps apart
        se = eng.train_step([None] * 5 + [uid.to(DEV), iid.to(DEV)], y.to(DEV), defer_sweep=True).cpu().clone()
        assert step == 0 or eng._tb_base < eng.step_count               # on the schedule: updates are pending
        mult = eng.dropout_multipliers(B).cpu()
        masks = {'dropout.user': mult[:, :D], 'dropout.item': mult[:, D:]}
        sse, _ = oracle.train_step(P, [None] * 5 + [uid, iid], y, hp, state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
    sd = model.state_dict()                                             # (flushes through the hook)
    for k, v in P.items():
        torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def test_transnet_step_at_cfg5_cardinalities():
    """cfg5 (TransNet++, 10 M users / 1 M items / 1 M words, E = 64, T = 1000, B = 128): one training step of the
    native engine against the oracle's literal three-optimiser step.  Per-rating source SE, the two auxiliary
    losses, every dense parameter, and EVERY element of both ID tables (the chunk-tagged sweep: touched rows
    by the entry waves, the other 10,999,9xx rows by weight decay alone)."""
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import TransNetEngine
    from test_oracle_golden import ill_conditioned
    B = 128
    hp = synthetic.hyper_params_for('cfg5_transnetpp_synthetic', dropout=0.0)
    V, U, I = hp['vocab'], hp['total_users'], hp['total_items']
    P = oracle.init_params(hp, vocab_size=V, seed=29)
    model = reviews4rec_amd.get_model_class('transnet++')(dict(hp, word_vectors=P['target.word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = TransNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    data, y = synthetic.Generator(hp, seed=7).batch(B)
    data = [torch.from_numpy(d) for d in data]
    y = torch.from_numpy(y)
    data[5][:3] = data[5][0]                                 # a user named three times, an item twice
    data[6][4:6] = data[6][4]
    data[5][B - 1] = U - 1                                    # the tables' last rows
    data[6][B - 2] = I - 1
    se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
    aux = eng.aux([d.to(DEV) for d in data]).cpu()
    states = dict(source=oracle.AdamState(), source_fm=oracle.AdamState(), target=oracle.AdamState())
    ref_se, lt, ltr = oracle.transnet_train_step(P, data, y, hp, states)
    torch.testing.assert_close(se, ref_se, rtol=1e-4, atol=1e-4)
    assert float(((se - ref_se) ** 2).mean()) < 1e-4
    torch.testing.assert_close(aux[:, 1].mean(), torch.tensor(lt), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(aux[:, 2].mean(), torch.tensor(ltr), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    assert sd['user_embedding.weight'].shape == (U + 2, 5) and sd['item_embedding.weight'].shape == (I + 2, 5)
    touched = {'user_embedding.weight': data[5].unique(), 'item_embedding.weight': data[6].unique()}
    for k, v in P.items():
        mine = sd[k].cpu()
        if k in touched:
            # rows no rating names: weight decay alone, the same arithmetic on both sides
            rest = torch.ones(v.shape[0], dtype=torch.bool)
            rest[touched[k]] = False
            torch.testing.assert_close(mine[rest], v[rest], rtol=1e-5, atol=1e-7, msg=lambda m: k + ' (untouched rows): ' + m)
            torch.testing.assert_close(mine[touched[k]], v[touched[k]], rtol=1e-4, atol=2e-5,
                                       msg=lambda m: k + ' (touched rows): ' + m)
        elif not ill_conditioned(k):
            diff = (mine - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k


@pytest.mark.parametrize('mt,E', [('deepconn', 50), ('deepconn++', 50), ('NARRE', 50), ('transnet', 50), ('transnet++', 50),
                                  ('deepconn++', 702), ('NARRE', 702), ('transnet++', 702)])
def test_word_embed_size_not_a_multiple_of_four(mt, E):
    """The reference takes any word_embed_size (hyper_params.py:64; the conv window is [3, E],
    common_pytorch_models.py:15): E = 50 (GloVe-50) on the native engines -- zero-padded table rows and conv
    weights inside the engine, the Parameters stay [100, 1, 3, 50] views -- two training steps and an eval forward
    against the CPU oracle at E = 50; the state_dict keeps the reference's shapes and the pad columns stay 0.
    E = 702 (padded to 704): beyond the 680 the weight-gradient window used to take in one pass."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from helpers import synthetic_review_batch
    from test_oracle_golden import ill_conditioned
    B, T, V, U, I, L = 12, 60, 300, 40, 30, 8
    R, W = (10, 20) if mt == 'NARRE' else (None, None)
    hp = dict(model_type=mt, latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, narre_num_reviews=10, narre_num_words=20, batch_size=B)
    P = oracle.init_params(hp, vocab_size=V, seed=41)
    tkey = 'target.word2vec.weight' if mt.startswith('transnet') else 'word2vec.weight'
    model = reviews4rec_amd.get_model_class(mt)(dict(hp, word_vectors=P[tkey].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    assert M.native_step_limits(hp) is None
    eng = M.make_engine(dict(hp, engine='native'), model)
    assert eng is not None and eng.E == (52 if E == 50 else 704) and eng.E_model == E
    is_tn = mt.startswith('transnet')
    states = dict(source=oracle.AdamState(), source_fm=oracle.AdamState(), target=oracle.AdamState()) if is_tn \
        else oracle.AdamState()
    for step in range(2):
        data, y = synthetic_review_batch(B, T, V, U, I, seed=50 + step, R=R, W=W)
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        if is_tn:
            ref_se, _, _ = oracle.transnet_train_step(P, data, y, hp, states)
            torch.testing.assert_close(se, ref_se, rtol=1e-4, atol=1e-4)
        else:
            sse, _ = oracle.train_step(P, data, y, hp, states)
            torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    sd = model.state_dict()
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        if not ill_conditioned(k):
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 5e-4, k
    for k, p in model.named_parameters():
        if k.endswith('convs.0.weight'):
            assert tuple(p.shape) == (100, 1, 3, E) and p.stride()[-2] == eng.E     # a view of the padded slot
            full = p.data.as_strided((100, 1, 3, eng.E), p.stride(), p.storage_offset())
            assert float(full[..., E:].abs().max()) == 0.0                           # pad columns: exactly 0
    model.eval()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=60, R=R, W=W)
    out = eng.predict([d.to(DEV) for d in data], None)[0].cpu()
    ref = oracle.model_forward(P, data, hp, train=False)
    ref = ref[0] if isinstance(ref, (list, tuple)) else ref
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('mt', ['deepconn', 'NARRE'])
def test_word_embed_size_not_a_multiple_of_four_module_path(mt):
    """The same configuration on the op-by-op module path (ops._padded_table): eval forward and one
    autograd step's gradients against the oracle."""
    import reviews4rec_amd
    from reviews4rec_amd.loss import MSELoss
    from helpers import synthetic_review_batch
    from test_oracle_golden import ill_conditioned
    B, T, E, V, U, I, L = 9, 60, 50, 300, 40, 30, 8
    R, W = (10, 20) if mt == 'NARRE' else (None, None)
    hp = dict(model_type=mt, latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, narre_num_reviews=10, narre_num_words=20, batch_size=B)
    P = oracle.init_params(hp, vocab_size=V, seed=43)
    model = reviews4rec_amd.get_model_class(mt)(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=70, R=R, W=W)
    out = model([d.to(DEV) for d in data])
    se = MSELoss(hp)(out, y.to(DEV), return_mean=False)
    torch.mean(se).backward()
    sse, grads = oracle.train_step(dict(P), data, y, hp, oracle.AdamState())
    torch.testing.assert_close(se.detach().sum().cpu(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
    got = {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}
    for k, v in grads.items():
        if v is not None and not ill_conditioned(k):
            torch.testing.assert_close(got[k], v, rtol=2e-4, atol=1e-6, msg=lambda m: k + ': ' + m)


@pytest.mark.parametrize('mt,L', [('deepconn', 160), ('MF', 80), ('transnet++', 80), ('deepconn++', 80)])
def test_latent_size_beyond_the_native_steps(mt, L):
    """latent_size has no bound in the reference (hyper_params.py:63).  The fused native steps are built for
    latent_size <= 32 (DeepCoNN's: <= 64, test_deepconn_native_step_at_latent_sizes_up_to_128); beyond that ``engine='auto'`` falls back -- with the reason -- to the op-by-op HIP path,
    whose factorization machine now takes up to 512 inputs (DeepCoNN's FM reads 2 x latent_size): one training
    step's loss and gradients and an eval forward against the CPU oracle at latent_size 48 / 80 / 160."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from reviews4rec_amd.loss import MSELoss
    from helpers import synthetic_review_batch
    B, T, E, V, U, I = 10, 50, 16, 200, 40, 30
    hp = dict(model_type=mt, latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, batch_size=B)
    assert M.native_step_limits(hp) is not None and M.module_path_limits(hp) is None
    P = oracle.init_params(hp, vocab_size=V, seed=47)
    tkey = 'target.word2vec.weight' if mt.startswith('transnet') else 'word2vec.weight'
    extra = dict(word_vectors=P[tkey].numpy()) if tkey in P else {}
    model = reviews4rec_amd.get_model_class(mt)(dict(hp, **extra))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    assert M.make_engine(dict(hp, engine='auto', log_file=None), model) is None       # no native step: module path
    data, y = synthetic_review_batch(B, T, V, U, I, seed=80)
    out = model([d.to(DEV) for d in data])
    pred = out[0] if isinstance(out, (list, tuple)) else out
    ref = oracle.model_forward(P, data, hp, train=True)
    ref = ref[0] if isinstance(ref, (list, tuple)) else ref
    torch.testing.assert_close(pred.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    if not mt.startswith('transnet'):
        se = MSELoss(hp)(pred, y.to(DEV), return_mean=False)
        torch.mean(se).backward()
        sse, grads = oracle.train_step(dict(P), data, y, hp, oracle.AdamState())
        torch.testing.assert_close(se.detach().sum().cpu(), torch.tensor(sse), rtol=1e-4, atol=1e-4)
        got = {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}
        for k, v in grads.items():
            if v is not None:
                torch.testing.assert_close(got[k], v, rtol=2e-4, atol=1e-6, msg=lambda m: k + ': ' + m)


def test_headline_batches_on_both_sides_of_the_resident_plans_edge():
    """cfg3's own batches (128 ratings x 2 documents of 1,000 words, E = 300) have 1,735 .. 1,914 row tiles of
    distinct words over the two towers: some fit the A-resident GEMM's plan with a row tile shared by >= 3
    workgroups (<= 7 1/3 row tiles per workgroup), some need the half-tile shares (two sharers, units 7 .. 9 on the
    two-column waves of SIMDs 0 - 2).  Each of the bench's first four batches, one training step from the same
    state under r4r_gemm_form 1 (balanced tile form) and 3 (A-resident): same bits in every parameter."""
    import reviews4rec_amd
    from reviews4rec_amd import _lib, synthetic
    from reviews4rec_amd.engine import DeepCoNNEngine
    from reviews4rec_amd.utils import xavier_init
    hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', dropout=0.0)
    hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
    gen = synthetic.Generator(hp, seed=synthetic.SEED)
    batches = [gen.batch(128) for _ in range(4)]
    tiles = [sum((len(np.unique(d[k])) + 15) // 16 for k in (3, 4)) for d, _ in batches]
    assert min(tiles) <= 1850 and max(tiles) > 1880, tiles          # both kinds of plan are exercised
    torch.manual_seed(0)
    m = reviews4rec_amd.get_model_class('deepconn')(hp)
    xavier_init(m)
    m = m.to(DEV).train()
    start = {k: v.clone() for k, v in m.state_dict().items()}
    lib = _lib.lib()
    out = {}
    try:
        for form in (1, 3):
            lib.r4r_gemm_form(form)
            for i, (data, y) in enumerate(batches):
                m.load_state_dict(start)
                eng = DeepCoNNEngine(m, lr=hp['lr'], weight_decay=hp['weight_decay'])
                se = eng.train_step([None if d is None else torch.from_numpy(d).to(DEV) for d in data], torch.from_numpy(y).to(DEV))
                out[form, i] = (se.cpu().clone(), eng.flat_p.cpu().clone())
    finally:
        lib.r4r_gemm_form(-1)
    for i in range(len(batches)):
        assert torch.equal(out[1, i][0], out[3, i][0]), i
        assert torch.equal(out[1, i][1], out[3, i][1]), i


def _transnetpp_pair(hp, V, seed, n=2):
    import reviews4rec_amd
    from reviews4rec_amd.engine import TransNetEngine
    P = oracle.init_params(hp, vocab_size=V, seed=seed)
    out = []
    for _ in range(n):
        model = reviews4rec_amd.get_model_class('transnet++')(dict(hp, word_vectors=P['target.word2vec.weight'].numpy()))
        model.load_state_dict(P)
        model = model.to(DEV).train()
        out.append((model, TransNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=5)))
    return out


def _same_bits(a, b, what):
    sa, sb = a[0].state_dict(), b[0].state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), '%s: parameter %s differs' % (what, k)
    (ma, va), (mb, vb) = a[1].moments(), b[1].moments()
    for k in ma:
        assert torch.equal(ma[k], mb[k]) and torch.equal(va[k], vb[k]), '%s: moments of %s differ' % (what, k)


@pytest.mark.parametrize('period', [2, 4, 8])
def test_temporally_blocked_sweep_is_the_dense_sweep_bit_for_bit(period, monkeypatch):
    """TransNet++'s ID-vector Adam with untouched chunks visited every `period`-th step (their pending gradient-zero
    updates applied together, rows a rating names catching up on the way: include/r4r.h) against the plain sweep that
    visits every element every step: 13 training steps over tables of 26 + 8 chunks, dropout on, a ragged batch, a
    row named four times, a step outside the schedule, an evaluation in the middle -- every parameter and both Adam
    moments identical to the bit."""
    from reviews4rec_amd import synthetic
    monkeypatch.setenv('R4R_SWEEP_PERIOD', str(period))
    hp = dict(synthetic.hyper_params_for('cfg5_transnetpp_synthetic', dropout=0.5), total_users=21000, total_items=6000,
              input_length=60, vocab=3000)
    plain, blocked = _transnetpp_pair(hp, hp['vocab'], seed=3)
    assert blocked[1].sweep_period == period
    gen = synthetic.Generator(hp, seed=11)
    pool = []
    for k in range(6):
        data, y = gen.batch(32 if k != 4 else 19)
        pool.append(([torch.from_numpy(d).to(DEV) for d in data], torch.from_numpy(y).to(DEV)))
    pool[2][0][5][:4] = 20999                                # the last chunk of the user table, a row named four times
    order = [0, 1, 2, 3, 4, 5, 0, 2, 4, 1, 3, 5, 0]
    for s, k in enumerate(order):
        nxt = pool[order[s + 1]][0] if s + 1 < len(order) else None
        plain[1].train_step(*pool[k], next_data=nxt)
        blocked[1].train_step(*pool[k], next_data=nxt, defer_sweep=(s != 6))   # step 6 leaves the schedule: every chunk
        if s == 3:                                           # an evaluation between two steps flushes first
            pa, pb = plain[1].predict(pool[1][0])[0], blocked[1].predict(pool[1][0])[0]
            assert torch.equal(pa, pb)
        if s in (1, 8):                                      # on the schedule: updates are pending
            assert blocked[1]._tb_base < blocked[1].step_count
    _same_bits(plain, blocked, 'period %d' % period)         # (state_dict() / moments() bring the pending updates in)
    assert blocked[1]._tb_base == blocked[1].step_count
    blocked[1].check_announcements()


def test_temporally_blocked_sweep_at_cfg5_cardinalities():
    """The same equality on cfg5's own tables (10 M x 5 + 1 M x 5: 13,428 chunks), six steps at the default period,
    the flush as its own launch (the way bench.py ends its timed region)."""
    from reviews4rec_amd import synthetic
    hp = dict(synthetic.hyper_params_for('cfg5_transnetpp_synthetic', dropout=0.0), input_length=100, vocab=5000)
    plain, blocked = _transnetpp_pair(hp, hp['vocab'], seed=19)
    gen = synthetic.Generator(hp, seed=23)
    pool = []
    for _ in range(4):
        data, y = gen.batch(128)
        pool.append(([torch.from_numpy(d).to(DEV) for d in data], torch.from_numpy(y).to(DEV)))
    for s in range(11):
        plain[1].train_step(*pool[s % 4], next_data=pool[(s + 1) % 4][0])
        blocked[1].train_step(*pool[s % 4], next_data=pool[(s + 1) % 4][0], defer_sweep=True)
    assert blocked[1]._tb_base < blocked[1].step_count       # behind: most chunks are waiting for their turn
    assert not torch.equal(plain[0].user_embedding.weight, blocked[0].user_embedding.weight)
    # nn.Module.state_dict() IS the model at any point of an epoch (main.py:125): the engine's pre-hook brings the
    # pending updates in first -- no explicit flush() here (a submodule's state_dict() does the same)
    sd = blocked[0].user_embedding.state_dict()
    assert torch.equal(sd['weight'], plain[0].user_embedding.weight)
    assert blocked[1]._tb_base == blocked[1].step_count
    _same_bits(plain, blocked, 'cfg5 tables')
    blocked[1].check_announcements()


@pytest.mark.parametrize('period,B', [(2, 128), (4, 128), (8, 128), (4, 2500), (3, 128)])
def test_temporally_blocked_mf_sweep_is_the_dense_sweep_bit_for_bit(period, B, monkeypatch):
    """r4r_mf_step's table sweep with untouched chunks visited every `period`-th step (the scheduled form: nothing is
    announced, rows a rating names catch up on the way) against the sweep that visits every element every step, on
    cfg2's own tables (192,403 x 64 + 63,001 x 64: 3,991 chunks): twelve training steps, dropout on, a ragged batch,
    a row named three times, an evaluation in the middle, a step outside the schedule -- every parameter and both Adam
    moments identical to the bit (B = 2,500 runs the wide entry waves, mf_adam_kernel<8>)."""
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import MFEngine
    monkeypatch.setenv('R4R_SWEEP_PERIOD', str(period))
    hp = synthetic.hyper_params_for('cfg2_mfdot_electronics', dropout=0.5)
    P = oracle.init_params(hp, seed=31)
    pair = []
    for _ in range(2):
        model = reviews4rec_amd.get_model_class('MF_dot')(hp)
        model.load_state_dict(P)
        model = model.to(DEV).train()
        pair.append((model, MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=9)))
    plain, blocked = pair
    assert blocked[1].sweep_period == period
    gen = synthetic.Generator(hp, seed=13)
    pool = []
    for k in range(5):
        data, y = gen.batch(B if k != 3 else B - 37)
        pool.append(([None] * 5 + [torch.from_numpy(data[5]).to(DEV), torch.from_numpy(data[6]).to(DEV)], torch.from_numpy(y).to(DEV)))
    pool[1][0][5][:3] = hp['total_users'] - 1                # the user table's last row, three times
    order = [0, 1, 2, 3, 4, 0, 2, 4, 1, 3, 0, 1, 1, 1, 2, 0, 3, 4, 4, 2]
    for s, k in enumerate(order):
        plain[1].train_step(*pool[k])
        blocked[1].train_step(*pool[k], defer_sweep=(s != 6))   # step 6 leaves the schedule: it visits every chunk
        if s == 3:
            assert torch.equal(plain[1].predict(pool[1][0])[0], blocked[1].predict(pool[1][0])[0])
        if s in (8, 17):
            assert blocked[1]._tb_base < blocked[1].step_count
            if B == 128:                                     # behind: most chunks are waiting for their turn
                assert not torch.equal(plain[0].user_embedding.weight, blocked[0].user_embedding.weight)
    # nn.Module.state_dict() IS the model at any point of an epoch (main.py:125): the engine's pre-hook brings the
    # pending updates in first -- no explicit flush() here
    sa, sb = plain[0].state_dict(), blocked[0].state_dict()
    assert blocked[1]._tb_base == blocked[1].step_count
    blocked[1].check_announcements()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    (ma, va), (mb, vb) = plain[1].moments(), blocked[1].moments()
    for k in ma:
        assert torch.equal(ma[k], mb[k]) and torch.equal(va[k], vb[k]), k


@pytest.mark.parametrize('kind,L,period', [('MF', 32, 8), ('NeuMF', 32, 4), ('GMF', 10, 2), ('MLP', 24, 8)])
def test_temporally_blocked_idnet_sweeps_are_the_dense_sweeps_bit_for_bit(kind, L, period, monkeypatch):
    """r4r_idnet_step's table sweeps (one pair + the bias vectors in one launch; NeuMF's second pair in another, with
    its own per-row state) blocked against plain, on Electronics-sized tables: eleven steps, dropout on, a ragged
    batch, an evaluation, a step outside the schedule -- parameters and moments identical to the bit."""
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import IdNetEngine
    monkeypatch.setenv('R4R_SWEEP_PERIOD', str(period))
    U, I = 192403, 63001
    hp = dict(model_type='MF' if kind == 'MF' else 'NeuMF', latent_size=L, dropout=0.3, total_users=U, total_items=I,
              lr=0.002, weight_decay=1e-6, word_embed_size=16, input_length=10, batch_size=128)
    if kind != 'MF':
        hp['neumf_stage'] = kind
    P = oracle.init_params(hp, vocab_size=None, seed=5)
    pair = []
    for _ in range(2):
        model = reviews4rec_amd.get_model_class(hp['model_type'])(hp)
        model.load_state_dict(P)
        model = model.to(DEV).train()
        pair.append((model, IdNetEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=9)))
    plain, blocked = pair
    assert blocked[1].sweep_period == period
    gen = synthetic.Generator(dict(hp, vocab=0), seed=17)
    pool = []
    for k in range(5):
        data, y = gen.batch(128 if k != 2 else 77)
        pool.append(([None] * 5 + [torch.from_numpy(data[5]).to(DEV), torch.from_numpy(data[6]).to(DEV)], torch.from_numpy(y).to(DEV)))
    pool[0][0][5][:2] = U                                    # the user table's last row (MF.py:21: U + 1 rows)
    order = [0, 1, 2, 3, 4, 0, 3, 1, 4, 2, 0]
    for s, k in enumerate(order):
        plain[1].train_step(*pool[k])
        blocked[1].train_step(*pool[k], defer_sweep=(s != 5))   # step 5 leaves the schedule: it visits every chunk
        if s == 3:
            assert torch.equal(plain[1].predict(pool[1][0])[0], blocked[1].predict(pool[1][0])[0])
        if s == 8:
            assert blocked[1]._tb_base < blocked[1].step_count
    sa, sb = plain[0].state_dict(), blocked[0].state_dict()
    assert blocked[1]._tb_base == blocked[1].step_count
    blocked[1].check_announcements()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    (ma, va), (mb, vb) = plain[1].moments(), blocked[1].moments()
    for k in ma:
        assert torch.equal(ma[k], mb[k]) and torch.equal(va[k], vb[k]), k


@pytest.mark.parametrize('family', ['MF_dot', 'transnet++', 'NeuMF'])
def test_temporally_blocked_sweeps_under_a_random_schedule(family, monkeypatch):
    """150 steps of whatever a host loop might do -- steps on and off the schedule, ragged batches, evaluations,
    optimiser state_dict round trips, a changing visit period -- the blocked engine against the plain one after every
    tenth step and at the end: identical bits."""
    import random
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.engine import MFEngine, TransNetEngine, IdNetEngine
    rnd = random.Random(20200725)
    U, I = 60000, 9000
    if family == 'MF_dot':
        hp = dict(synthetic.hyper_params_for('cfg2_mfdot_electronics', dropout=0.4), total_users=U, total_items=I)
        mk = lambda m: MFEngine(m, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=3)
        P = oracle.init_params(hp, seed=7)
        tables = ['user_embedding.weight', 'item_embedding.weight']
    elif family == 'NeuMF':
        hp = dict(model_type='NeuMF', neumf_stage='NeuMF', latent_size=16, dropout=0.4, total_users=U, total_items=I, lr=0.002,
                  weight_decay=1e-6, word_embed_size=16, input_length=10, batch_size=64, vocab=0)
        mk = lambda m: IdNetEngine(m, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=3)
        P = oracle.init_params(hp, vocab_size=None, seed=7)
        tables = ['gmf_user_embedding.weight', 'mlp_item_embedding.weight']
    else:
        hp = dict(synthetic.hyper_params_for('cfg5_transnetpp_synthetic', dropout=0.4), total_users=U, total_items=I,
                  input_length=40, vocab=2000)
        mk = lambda m: TransNetEngine(m, lr=hp['lr'], weight_decay=hp['weight_decay'], seed=3)
        P = oracle.init_params(hp, vocab_size=hp['vocab'], seed=7)
        tables = ['user_embedding.weight', 'item_embedding.weight']
    pair = []
    for _ in range(2):
        kw = dict(hp, word_vectors=P['target.word2vec.weight'].numpy()) if family == 'transnet++' else hp
        model = reviews4rec_amd.get_model_class(hp['model_type'])(kw)
        model.load_state_dict(P)
        model = model.to(DEV).train()
        pair.append((model, mk(model)))
    plain, blocked = pair
    gen = synthetic.Generator(hp, seed=5)
    pool = []
    for k in range(9):
        data, y = gen.batch(rnd.choice([64, 64, 64, 41, 17]))
        pool.append(([None if (d.shape[-1] == 1 and family != 'transnet++' and j < 5) else torch.from_numpy(d).to(DEV)
                      for j, d in enumerate(data)], torch.from_numpy(y).to(DEV)))

    def same(what):
        blocked[1].flush()                                   # (whoever reads the Parameters directly brings them up to date first)
        sa, sb = plain[0].state_dict(), blocked[0].state_dict()
        assert all(torch.equal(sa[k], sb[k]) for k in sa), what
        (ma, va), (mb, vb) = plain[1].moments(), blocked[1].moments()
        assert all(torch.equal(ma[k], mb[k]) and torch.equal(va[k], vb[k]) for k in ma), what

    cur = 0
    behind = 0
    for s in range(150):
        nxt = rnd.randrange(9)
        act = rnd.random()
        on = act < 0.85                                      # (else: a step off the schedule, which visits every chunk)
        plain[1].train_step(*pool[cur])
        blocked[1].train_step(*pool[cur], next_data=pool[nxt][0], defer_sweep=on)
        if on:
            behind += int(not torch.equal(getattr_path(plain[0], tables[0]), getattr_path(blocked[0], tables[0])))
        cur = nxt
        r = rnd.random()
        if r < 0.06:
            assert torch.equal(plain[1].predict(pool[2][0])[0], blocked[1].predict(pool[2][0])[0])
        elif r < 0.10:
            for _, e in pair:
                e.load_state_dict(e.state_dict())            # resume from one's own state: nothing may be lost
        elif r < 0.14:
            blocked[1].sweep_period = rnd.choice([2, 3, 5, 8])
        if s % 10 == 9:
            same('after step %d' % s)
    blocked[1].check_announcements()
    same('at the end')
    assert behind > 20                                       # the blocked tables really were behind between steps


def getattr_path(obj, path):
    for part in path.split('.'):
        obj = getattr(obj, part)
    return obj


@pytest.mark.parametrize('L,dropout', [(48, 0.0), (64, 0.5), (33, 0.0), (96, 0.0), (128, 0.5)])
def test_deepconn_native_step_at_latent_sizes_up_to_128(L, dropout):
    """latent_size 33 .. 128 (hyper_params.py:63 has no bound) on DeepCoNN's fused native
    step -- the head's FM wave takes two (65 .. 128: four) of the 2 L inputs per lane -- instead of the 3x slower op-by-op path: two training steps
    (dropout multipliers drawn on the device, injected into the oracle) and an eval forward against the CPU oracle."""
    import copy
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from helpers import synthetic_review_batch
    B, T, E, V, U, I = 24, 60, 32, 300, 40, 30
    hp = dict(model_type='deepconn', latent_size=L, word_embed_size=E, input_length=T, dropout=dropout, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, batch_size=B)
    assert M.native_step_limits(hp) is None
    P = oracle.init_params(hp, vocab_size=V, seed=61)
    model = reviews4rec_amd.get_model_class('deepconn')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = M.make_engine(dict(hp, engine='auto', log_file=None), model)
    assert eng is not None and type(eng).__name__ == 'DeepCoNNEngine'
    state = oracle.AdamState()
    for step in range(2):
        data, y = synthetic_review_batch(B, T, V, U, I, seed=90 + step)
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        masks = None
        if dropout > 0:
            mult = eng.dropout_multipliers(B, T).cpu()
            assert 0.3 < float((mult == 0).float().mean()) < 0.7
            masks = {'user_conv.dropout': mult[:, :L], 'item_conv.dropout': mult[:, L:]}
        sse, grads = oracle.train_step(P, data, y, hp, state, masks=masks)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
        if step == 0:
            got = eng.grads()
            for k, v in grads.items():
                if v is not None:
                    torch.testing.assert_close(got[k].cpu(), v, rtol=2e-4, atol=1e-6, msg=lambda m: k + ': ' + m)
    sd = model.state_dict()
    for k, v in P.items():
        diff = (sd[k].cpu() - v).abs()
        assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3, k
        assert float(diff.max()) < 2.5e-3, k
    model.eval()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=99)
    pred = eng.predict([d.to(DEV) for d in data], None)[0].cpu()
    ref = oracle.model_forward(P, data, dict(hp, dropout=0.0), train=False)
    torch.testing.assert_close(pred, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('L,R', [(48, 10), (10, 40), (33, 33)])
def test_narre_native_step_at_latent_sizes_and_review_counts_up_to_64(L, R):
    """NARRE with latent_size / narre_num_reviews 33 .. 64 (hyper_params.py:63,78 have no bound) on
    the native engine: the head's 64 x 64 instantiation, and -- the fused ID-table role keeps a row in registers up to
    L = 32 -- the step as gradients -> flat Adam -> r4r_narre_rows_apply_large.  Two training steps and an eval forward
    against the CPU oracle."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from helpers import synthetic_review_batch
    from test_oracle_golden import ill_conditioned
    B, W, E, V, U, I = 12, 20, 16, 300, 60, 50
    hp = dict(model_type='NARRE', latent_size=L, word_embed_size=E, input_length=W, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, narre_num_reviews=R, narre_num_words=W, batch_size=B)
    assert M.native_step_limits(hp) is None
    P = oracle.init_params(hp, vocab_size=V, seed=67)
    model = reviews4rec_amd.get_model_class('NARRE')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = M.make_engine(dict(hp, engine='auto', log_file=None), model)
    assert eng is not None and type(eng).__name__ == 'NarreEngine'
    state = oracle.AdamState()
    for step in range(2):
        data, y = synthetic_review_batch(B, W, V, U, I, seed=110 + step, R=R, W=W)
        gen = torch.Generator().manual_seed(step)
        data[1] = torch.randint(0, U + 2, (B, R), generator=gen)
        data[2] = torch.randint(0, I + 2, (B, R), generator=gen)
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
    model.eval()
    pred = eng.predict([d.to(DEV) for d in data], None)[0].cpu()
    ref = oracle.model_forward(P, data, hp, train=False)
    torch.testing.assert_close(pred, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('path', ['native', 'module'])
def test_word_embed_size_768(path):
    """word_embed_size beyond 680 (hyper_params.py:64 has no bound; 768 = a BERT-width table).  The
    weight-gradient window is 3 E / 4 = 576 float4 columns wide: two passes of the split's documents (csrc/wgrad_device.h),
    the projection GEMM runs its K loop over 48 chunks.  DeepCoNN, one training step's loss and gradients, the weights
    after two steps and an eval forward against the CPU oracle, on the fused native step and on the op-by-op path."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    from helpers import synthetic_review_batch
    from test_oracle_golden import ill_conditioned
    B, T, E, V, U, I, L = 12, 40, 768, 300, 40, 30, 8
    hp = dict(model_type='deepconn', latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, batch_size=B)
    assert M.native_step_limits(hp) is None and M.module_path_limits(hp) is None
    P = oracle.init_params(hp, vocab_size=V, seed=71)
    model = reviews4rec_amd.get_model_class('deepconn')(dict(hp, word_vectors=P['word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    state = oracle.AdamState()
    if path == 'native':
        eng = M.make_engine(dict(hp, engine='native', log_file=None), model)
        assert type(eng).__name__ == 'DeepCoNNEngine'
    else:
        opt = Adam(model.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])
    for step in range(2):
        data, y = synthetic_review_batch(B, T, V, U, I, seed=120 + step)
        dev = [d.to(DEV) for d in data]
        if path == 'native':
            se = eng.train_step(dev, y.to(DEV)).cpu().clone()
            got = {k: v.cpu() for k, v in eng.grads().items()} if step == 0 else None
        else:
            opt.zero_grad()
            se = MSELoss(hp)(model(dev), y.to(DEV), return_mean=False)
            torch.mean(se).backward()
            got = {k: p.grad.cpu().clone() for k, p in model.named_parameters() if p.grad is not None} if step == 0 else None
            opt.step()
            se = se.detach().cpu()
        sse, grads = oracle.train_step(P, data, y, hp, state)
        torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
        if step == 0:
            for k, v in grads.items():
                if v is not None and not ill_conditioned(k):
                    torch.testing.assert_close(got[k], v, rtol=2e-4, atol=1e-6, msg=lambda m: k + ': ' + m)
    sd = model.state_dict()
    for k, v in P.items():
        if not ill_conditioned(k):
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 2.5e-3, k
    model.eval()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=130)
    dev = [d.to(DEV) for d in data]
    pred = (eng.predict(dev, None)[0] if path == 'native' else model(dev).detach()).cpu()
    torch.testing.assert_close(pred, oracle.model_forward(P, data, hp, train=False), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('mt,L', [('deepconn++', 64), ('transnet', 48), ('transnet++', 64), ('deepconn++', 33)])
def test_native_steps_of_the_other_text_families_at_latent_sizes_up_to_64(mt, L):
    """latent_size 33 .. 64 (hyper_params.py:63 has no bound) on the fused native steps of
    DeepCoNN++ and TransNet(++) -- 64-wide instantiations of their head kernels -- instead of the op-by-op path: two
    training steps (TransNet: the three-optimiser step) and an eval forward against the CPU oracle."""
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from helpers import synthetic_review_batch
    from test_oracle_golden import ill_conditioned
    B, T, E, V, U, I = 12, 60, 32, 300, 40, 30
    hp = dict(model_type=mt, latent_size=L, word_embed_size=E, input_length=T, dropout=0.0, total_users=U,
              total_items=I, lr=0.002, weight_decay=1e-6, batch_size=B)
    assert M.native_step_limits(hp) is None
    P = oracle.init_params(hp, vocab_size=V, seed=83)
    tkey = 'target.word2vec.weight' if mt.startswith('transnet') else 'word2vec.weight'
    model = reviews4rec_amd.get_model_class(mt)(dict(hp, word_vectors=P[tkey].numpy()))
    model.load_state_dict(P)
    model = model.to(DEV).train()
    eng = M.make_engine(dict(hp, engine='native', log_file=None), model)
    assert eng is not None and eng.L == L
    is_tn = mt.startswith('transnet')
    states = dict(source=oracle.AdamState(), source_fm=oracle.AdamState(), target=oracle.AdamState()) if is_tn \
        else oracle.AdamState()
    for step in range(2):
        data, y = synthetic_review_batch(B, T, V, U, I, seed=140 + step)
        se = eng.train_step([d.to(DEV) for d in data], y.to(DEV)).cpu().clone()
        if is_tn:
            ref_se, _, _ = oracle.transnet_train_step(P, data, y, hp, states)
            torch.testing.assert_close(se, ref_se, rtol=1e-4, atol=1e-4)
        else:
            sse, _ = oracle.train_step(P, data, y, hp, states)
            torch.testing.assert_close(se.sum(), torch.tensor(sse), rtol=1e-4, atol=1e-3)
    if hasattr(eng, 'flush'):
        eng.flush()
    sd = model.state_dict()
    for k, v in P.items():
        if not ill_conditioned(k):
            diff = (sd[k].cpu() - v).abs()
            assert float((diff > 2e-5 + 1e-4 * v.abs()).float().mean()) < 2e-3 and float(diff.max()) < 2.5e-3, k
    model.eval()
    data, y = synthetic_review_batch(B, T, V, U, I, seed=150)
    out = eng.predict([d.to(DEV) for d in data], None)[0].cpu()
    ref = oracle.model_forward(P, data, hp, train=False)
    ref = ref[0] if isinstance(ref, (list, tuple)) else ref
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
