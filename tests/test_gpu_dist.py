"""Data parallelism on the real HIP path: 2 processes share the one GPU of the test box (gloo
transport; RCCL itself needs one GPU per rank).  DP(2) over the two halves of a golden batch must
reproduce the reference's single-process step on the whole batch -- dense-layer gradients through
the flat all-reduce bucket (C1), ID tables / biases through the compact-list exchange (C2)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, 'tests')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '')
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    r4dist.init_from_env()
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    opt = Adam(model.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])
    data, y = g.batch(0, 'cuda')
    sd, sy = r4dist.shard_batch(data, y, rank, world)
    n_global = dp.global_count(sy.shape[0], sy.device)
    opt.zero_grad()
    se = MSELoss(hp)(model(sd), sy, return_mean=False)
    (se.sum() * dp.loss_scale(sy.shape[0], n_global)).backward()
    dp.allreduce_grads()
    opt.step()
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(out_dir, 'r%d.pt' % rank))
    torch.save(sorted(dp._sparse_set), os.path.join(out_dir, 'sparse%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('case', ['mf_dot', 'narre_e16', 'deepconnpp_e20'])
def test_dp2_hip_step_equals_reference_step_on_whole_batch(tmp_path, case):
    sys.path.insert(0, TESTS)
    from helpers import Golden
    from test_oracle_golden import ill_conditioned
    port = _free_port()
    mp.get_context('spawn')
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    g = Golden(case)
    r0 = torch.load(os.path.join(tmp_path, 'r0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'r1.pt'))
    assert torch.load(os.path.join(tmp_path, 'sparse0.pt'))      # some tables went through the compact exchange
    for k, v in g.params('w1').items():
        assert torch.equal(r0[k], r1[k]), k                       # replicas stay bit-identical
        if not ill_conditioned(k):
            torch.testing.assert_close(r0[k], v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def _engine_worker(rank, world, port, case, out_dir, exchange):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '')
    if exchange not in ('autotune', 'autotune_fail'):
        os.environ['R4R_DP_EXCHANGE'] = exchange
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist
    from reviews4rec_amd.engine import DeepCoNNEngine
    r4dist.init_from_env()
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    eng = DeepCoNNEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'], dp=dp, rank=rank, conv_algo=2)
    if exchange == 'autotune':
        before = eng.flat_p.clone()
        times = eng.autotune_exchange(trials=3)
        assert set(times) == {'allreduce', 'gather'} and eng.exchange in times
        assert torch.equal(before, eng.flat_p)                     # tuning runs on scratch buffers
    elif exchange == 'autotune_fail':
        # the 'gather' candidate cannot be set up on rank 1 ONLY: every rank must drop it together (VERDICT r4 next #2d) and go on
        # with the all-reduce -- not hang in a collective the failed rank never joins, not keep a form one rank cannot run
        import warnings
        real = eng._exchange_prepare

        def flaky(how):
            if how == 'gather' and rank == 1:
                raise RuntimeError('injected: this rank cannot set the gather form up')
            return real(how)
        eng._exchange_prepare = flaky
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter('always')
            times = eng.autotune_exchange(trials=3)
        eng._exchange_prepare = real
        assert set(times) == {'allreduce'} and eng.exchange == 'allreduce', (times, eng.exchange)
        assert any('dropped on every rank' in str(w.message) for w in seen)
    else:
        assert eng.exchange == exchange
    shards = [r4dist.shard_batch(*g.batch(k, 'cuda'), rank, world) for k in (0, 1)]
    ses = []
    for step in range(3):
        sd, sy = shards[step % 2]
        nxt = shards[(step + 1) % 2][0] if step < 2 else None      # shapes differ: the guess is declined
        n_global = dp.global_count(sy.shape[0], sy.device)
        ses.append(eng.train_step(sd, sy, n_global=n_global, next_data=nxt).cpu().clone())
    if exchange == 'peer':
        assert eng._peer is not None and eng._peer.world == world
        eng._peer.check()                                          # no wait timed out
        eng._peer.close()
    torch.save({'w': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'se': ses},
               os.path.join(out_dir, 'e%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('exchange', ['allreduce', 'gather', 'autotune', 'autotune_fail', 'peer'])
def test_dp2_native_engine_follows_the_reference_trajectory(tmp_path, exchange):
    """The fused DeepCoNN step under data parallelism -- gradients summed by one all-reduce and a
    separate Adam launch, or all_gathered and summed in rank order inside the Adam launch, or
    whichever of the two the engine measures to be faster, or ('peer') pushed by each rank's own kernel into
    buffers the two processes map from each other over CUDA IPC on the one GPU (dist.PeerExchange, csrc/peer.hip)
    and summed the same way: 2 ranks x half batches == the reference's 3 single-process steps."""
    sys.path.insert(0, TESTS)
    from helpers import Golden
    case = 'deepconn_e20'
    port = _free_port()
    mp.spawn(_engine_worker, args=(2, port, case, str(tmp_path), exchange), nprocs=2, join=True)
    g = Golden(case)
    r0 = torch.load(os.path.join(tmp_path, 'e0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'e1.pt'))
    for step in range(3):
        se = torch.cat([r0['se'][step], r1['se'][step]])
        torch.testing.assert_close(se, g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
    for k, v in g.params('w3').items():
        assert torch.equal(r0['w'][k], r1['w'][k]), k               # replicas stay bit-identical
        torch.testing.assert_close(r0['w'][k], v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def _mf_worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '')
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist, main as M
    from reviews4rec_amd.engine import MFEngine
    r4dist.init_from_env()
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    model.hyper_params['batch_size'] = 64                    # per-rank batch: every shard below fits it
    eng = M.make_engine(dict(hp, engine='auto', batch_size=64), model, dp=dp, rank=rank)
    assert isinstance(eng, MFEngine) and eng.dp is not None
    ses = []
    defer = os.environ.get('R4R_TEST_DEFER') == '1'          # the temporally blocked sweep over all ranks' announced next shards
    steps = int(os.environ.get('R4R_TEST_STEPS', '3'))
    shards = [r4dist.shard_batch(*g.batch(step % 2, 'cuda'), rank, world) for step in range(steps + 1)]
    for step in range(steps):
        data, y = g.batch(step % 2, 'cuda')
        sd, sy = shards[step]                                        # ragged: the ranks' shards differ in length
        # (with the global count known the shards are padded to hyper_params['batch_size'] and no sizes are
        # exchanged; without it the ranks agree on the sizes first: both forms)
        kw = dict(next_data=shards[step + 1][0] if step < steps - 1 else None, defer_sweep=True) if defer else {}
        ses.append(eng.train_step(sd, sy, n_global=int(y.shape[0]) if step != 1 else None, **kw).cpu().clone())
        if defer and step == 0:
            assert eng._tb_period == eng.sweep_period        # (the schedule is in force from the first step on)
    if defer:
        eng.flush()
    if os.environ.get('R4R_DP_EXCHANGE') == 'peer':          # the steps with a known global count went over peer-mapped memory
        assert eng._peer is not None and eng._peer_epoch == steps - 1
        eng.check_exchange()
    eng.close()
    torch.save({'w': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'se': ses},
               os.path.join(out_dir, 'm%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('case', ['mf_dot', 'mf_bias_only', 'mf_dot+blocked', 'mf_dot+registered', 'mf_dot+blocked+registered',
                                  'mf_dot+peer', 'mf_bias_only+peer', 'mf_dot+blocked+peer'])
def test_dp2_native_mf_step_equals_the_single_process_step(tmp_path, case, monkeypatch):
    """MF under data parallelism on the native step (r4r_mf_grad -> all_gather of the packed compact
    rows -> r4r_mf_apply): 2 ranks x ragged shards reproduce the reference's 3 single-process steps,
    replicas bit-identical -- and identical, bit for bit, to the single-process native step on the
    whole batch (same entries in the same order).  +registered: the update launch behind a registering launch (the form
    of more than 2,048 gathered ratings) instead of finding its rows by scanning the ids; +peer: no collective call,
    the blocks pushed into every rank's buffer over peer-mapped memory (r4r_mf_grad_push -> r4r_mf_apply_peer)."""
    sys.path.insert(0, TESTS)
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd.engine import MFEngine
    if case.endswith('+peer'):
        case = case[:-len('+peer')]
        monkeypatch.setenv('R4R_DP_EXCHANGE', 'peer')
    if case.endswith('+registered'):
        case = case[:-len('+registered')]
        monkeypatch.setenv('R4R_MF_DP_REGISTER', '1')
    if case.endswith('+blocked'):                            # ... with every rank announcing its next shard (r4r.h)
        case = case[:-len('+blocked')]
        monkeypatch.setenv('R4R_TEST_DEFER', '1')
    port = _free_port()
    mp.spawn(_mf_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    g = Golden(case)
    r0 = torch.load(os.path.join(tmp_path, 'm0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'm1.pt'))
    model, hp = build_model(g)
    model.train()
    eng = MFEngine(model, lr=hp['lr'], weight_decay=hp['weight_decay'])
    for step in range(3):
        data, y = g.batch(step % 2, 'cuda')
        eng.train_step(data, y)
        se = torch.cat([r0['se'][step], r1['se'][step]])
        torch.testing.assert_close(se, g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
    single = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for k, v in g.params('w3').items():
        assert torch.equal(r0['w'][k], r1['w'][k]), k               # replicas stay bit-identical
        assert torch.equal(r0['w'][k], single[k]), k                # == the single-process native step
        torch.testing.assert_close(r0['w'][k], v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


@pytest.mark.parametrize('blocked', [False, True])
def test_dp2_mf_peer_exchange_equals_the_collective_over_many_steps(tmp_path, monkeypatch, blocked):
    """Twenty steps of two ranks with the blocks pushed over peer-mapped memory (both parity buffers reused nine times,
    the scheduled sweep's chunks visited twice and more) end on the bits of the same steps over the collective."""
    monkeypatch.setenv('R4R_TEST_STEPS', '20')
    if blocked:
        monkeypatch.setenv('R4R_TEST_DEFER', '1')
    got = {}
    for how in ('collective', 'peer'):
        out = tmp_path / how
        out.mkdir()
        if how == 'peer':
            monkeypatch.setenv('R4R_DP_EXCHANGE', 'peer')
        mp.spawn(_mf_worker, args=(2, _free_port(), 'mf_dot', str(out)), nprocs=2, join=True)
        got[how] = [torch.load(os.path.join(out, 'm%d.pt' % r)) for r in range(2)]
    for r in range(2):
        for k, v in got['collective'][0]['w'].items():
            assert torch.equal(got['peer'][r]['w'][k], v), (r, k)
        for a, b in zip(got['peer'][r]['se'], got['collective'][r]['se']):
            assert torch.equal(a, b)


def _empty_shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '', R4R_DP_EXCHANGE='allreduce')
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist
    from reviews4rec_amd.engine import DeepCoNNEngine, MFEngine
    r4dist.init_from_env()
    out = {}
    for case, Eng, kw in (('deepconn_e20', DeepCoNNEngine, dict(conv_algo=2)), ('mf_dot', MFEngine, {})):
        g = Golden(case)
        model, hp = build_model(g)
        model.train()
        dp = r4dist.DataParallel(model)
        dp.broadcast_parameters()
        eng = Eng(model, lr=hp['lr'], weight_decay=hp['weight_decay'], dp=dp, rank=rank, **kw)
        data, y = g.batch(0, 'cuda')
        one = [None if d is None else d[:1] for d in data]           # a global batch of ONE rating: rank 1's shard is empty
        sd, sy = r4dist.shard_batch(one, y[:1], rank, world)
        se = eng.train_step(sd, sy, n_global=1)
        assert se.numel() == (1 if rank == 0 else 0)
        out[case] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save(out, os.path.join(out_dir, 'z%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_dp2_native_steps_with_an_empty_shard(tmp_path):
    """A global batch of one rating over two ranks: the rank without rows contributes a zero
    gradient / no entries, both ranks end on the weights of the single-process step on that rating."""
    sys.path.insert(0, TESTS)
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd.engine import DeepCoNNEngine, MFEngine
    port = _free_port()
    mp.spawn(_empty_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'z0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'z1.pt'))
    for case, Eng, kw in (('deepconn_e20', DeepCoNNEngine, dict(conv_algo=2)), ('mf_dot', MFEngine, {})):
        g = Golden(case)
        model, hp = build_model(g)
        model.train()
        eng = Eng(model, lr=hp['lr'], weight_decay=hp['weight_decay'], **kw)
        data, y = g.batch(0, 'cuda')
        eng.train_step([None if d is None else d[:1] for d in data], y[:1])
        for k, v in model.state_dict().items():
            assert torch.equal(r0[case][k], r1[case][k]), (case, k)
            torch.testing.assert_close(r0[case][k], v.cpu(), rtol=1e-6, atol=1e-7, msg=lambda m: case + ' ' + k + ': ' + m)


def _transnet_worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '')
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist, main as M
    from reviews4rec_amd.engine import TransNetEngine
    r4dist.init_from_env()
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    model.hyper_params['batch_size'] = 64                    # per-rank batch: every shard below fits it
    eng = M.make_engine(dict(hp, engine='auto', batch_size=64), model, dp=dp, rank=rank)
    assert isinstance(eng, TransNetEngine) and eng.dp is not None
    ses, aux = [], []
    for step in range(3):
        data, y = g.batch(step % 2, 'cuda')
        sd, sy = r4dist.shard_batch(data, y, rank, world)
        before = eng.sse.clone()
        ses.append(eng.train_step(sd, sy, n_global=int(y.shape[0]) if step != 1 else None).cpu().clone())
        aux.append((eng.sse - before)[1:].cpu().clone())
    torch.save({'w': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'se': ses, 'aux': aux},
               os.path.join(out_dir, 't%d.pt' % rank))
    torch.distributed.destroy_process_group()


def _blocked_dp_worker(rank, world, port, family, defer, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'), R4R_DP_SINGLE='',
                      R4R_SWEEP_PERIOD='4')
    import oracle
    import reviews4rec_amd
    from reviews4rec_amd import dist as r4dist, main as M, synthetic
    r4dist.init_from_env()
    if family == 'MF_dot':
        hp = dict(synthetic.hyper_params_for('cfg2_mfdot_electronics', dropout=0.3), total_users=30000, total_items=9000, batch_size=64)
        P = oracle.init_params(hp, seed=11)
        model = reviews4rec_amd.get_model_class('MF_dot')(hp)
    elif family == 'NeuMF':
        hp = dict(model_type='NeuMF', neumf_stage='NeuMF', latent_size=16, dropout=0.3, total_users=30000, total_items=9000,
                  lr=0.002, weight_decay=1e-6, word_embed_size=16, input_length=10, batch_size=64, vocab=0)
        P = oracle.init_params(hp, vocab_size=None, seed=11)
        model = reviews4rec_amd.get_model_class('NeuMF')(hp)
    else:
        hp = dict(synthetic.hyper_params_for('cfg5_transnetpp_synthetic', dropout=0.3), total_users=60000, total_items=9000,
                  input_length=40, vocab=2000, batch_size=64)
        P = oracle.init_params(hp, vocab_size=hp['vocab'], seed=11)
        model = reviews4rec_amd.get_model_class('transnet++')(dict(hp, word_vectors=P['target.word2vec.weight'].numpy()))
    model.load_state_dict(P)
    model = model.cuda().train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    eng = M.make_engine(dict(hp, engine='auto'), model, dp=dp, rank=rank)
    assert eng is not None and eng.dp is not None and eng.sweep_period == 4
    gen = synthetic.Generator(hp, seed=21)                   # (the same stream on every rank: each takes its shard)
    shards = []
    for k in range(6):
        data, y = gen.batch(96 if k != 3 else 51)            # global batches; the ranks' shards are ragged
        data = [None if (d.shape[-1] == 1 and family != 'transnet++' and j < 5) else torch.from_numpy(d).cuda() for j, d in enumerate(data)]
        shards.append((r4dist.shard_batch(data, torch.from_numpy(y).cuda(), rank, world), int(y.shape[0])))
    order = [0, 1, 2, 3, 4, 5, 1, 3, 0]
    for s, k in enumerate(order):
        (sd, sy), n_global = shards[k]
        nxt = shards[order[s + 1]][0][0] if s + 1 < len(order) else None
        if defer:                                            # (step 6 leaves the schedule: it visits every chunk)
            eng.train_step(sd, sy, n_global=n_global, next_data=nxt, defer_sweep=(s != 6))
        else:
            eng.train_step(sd, sy, n_global=n_global)
    eng.flush()
    m, v = eng.moments()
    torch.save({'w': {k: t.detach().cpu() for k, t in model.state_dict().items()},
                'm': {k: t.detach().cpu() for k, t in m.items()}, 'v': {k: t.detach().cpu() for k, t in v.items()},
                'used': bool(getattr(eng, '_tb_used', False))}, os.path.join(out_dir, 'b%d_%d.pt' % (int(defer), rank)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('family', ['MF_dot', 'transnet++', 'NeuMF'])
def test_dp2_blocked_sweep_equals_the_plain_sweep(tmp_path, family):
    """The temporally blocked sweep under data parallelism (r4r_mf_apply / r4r_transnet_rows_apply /
    r4r_idnet_rows_apply on the same schedule on every rank, the ranks' forwards catching up the rows they name): 2
    ranks x 9 steps of ragged shards, one step off the schedule -- parameters and both moments identical, bit for bit,
    across the ranks and to the same run with the plain sweep."""
    for defer in (0, 1):
        mp.spawn(_blocked_dp_worker, args=(2, _free_port(), family, defer, str(tmp_path)), nprocs=2, join=True)
    runs = {(d, r): torch.load(os.path.join(tmp_path, 'b%d_%d.pt' % (d, r))) for d in (0, 1) for r in (0, 1)}
    assert runs[(1, 0)]['used'] and not runs[(0, 0)]['used']
    ref = runs[(0, 0)]
    for key, run in runs.items():
        for part in ('w', 'm', 'v'):
            for k, t in ref[part].items():
                assert torch.equal(run[part][k], t), (key, part, k)


@pytest.mark.parametrize('case', ['transnet_e16', 'transnetpp_e16'])
def test_dp2_native_transnet_step_follows_the_reference_trajectory(tmp_path, case):
    """TransNet(++) under data parallelism on the native step: gradients only per rank, one all-reduce
    of the flat dense gradient + the flat Adam, and (TransNet++) the ranks' compact ID-vector rows
    gathered into the same tagged sweep: 2 ranks x ragged shards == the reference's three
    single-process three-optimiser steps; replicas bit-identical."""
    sys.path.insert(0, TESTS)
    from helpers import Golden
    port = _free_port()
    mp.spawn(_transnet_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    g = Golden(case)
    r0 = torch.load(os.path.join(tmp_path, 't0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 't1.pt'))
    for step in range(3):
        se = torch.cat([r0['se'][step], r1['se'][step]])
        torch.testing.assert_close(se, g.arr('tn_se%d' % step), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(r0['aux'][step] + r1['aux'][step], g.arr('tn_aux%d' % step), rtol=1e-4, atol=1e-5)
    for k, v in g.params('tn_w3').items():
        assert torch.equal(r0['w'][k], r1['w'][k]), k               # replicas stay bit-identical
        torch.testing.assert_close(r0['w'][k], v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def _dcpp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '')
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist, main as M
    from reviews4rec_amd.engine import DeepCoNNPPEngine
    r4dist.init_from_env()
    g = Golden('deepconnpp_e20')
    model, hp = build_model(g)
    model.train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    model.hyper_params['batch_size'] = 64                    # per-rank batch: every shard below fits it
    eng = M.make_engine(dict(hp, engine='auto', batch_size=64), model, dp=dp, rank=rank)
    assert isinstance(eng, DeepCoNNPPEngine) and eng.dp is not None
    ses = []
    for step in range(3):
        data, y = g.batch(step % 2, 'cuda')
        sd, sy = r4dist.shard_batch(data, y, rank, world)
        ses.append(eng.train_step(sd, sy, n_global=int(y.shape[0]) if step != 1 else None).cpu().clone())
    torch.save({'w': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'se': ses},
               os.path.join(out_dir, 'd%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_dp2_native_deepconnpp_step_follows_the_reference_trajectory(tmp_path):
    """DeepCoNN++ under data parallelism on the native step (gradients only per rank, flat all-reduce + Adam,
    the ranks' (uid, iid, d loss / d pred) gathered into the ID-bias sweep): 2 ranks x ragged shards == the
    reference's three single-process steps; replicas bit-identical."""
    sys.path.insert(0, TESTS)
    from helpers import Golden
    port = _free_port()
    mp.spawn(_dcpp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = Golden('deepconnpp_e20')
    r0 = torch.load(os.path.join(tmp_path, 'd0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'd1.pt'))
    for step in range(3):
        se = torch.cat([r0['se'][step], r1['se'][step]])
        torch.testing.assert_close(se, g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
    for k, v in g.params('w3').items():
        assert torch.equal(r0['w'][k], r1['w'][k]), k               # replicas stay bit-identical
        torch.testing.assert_close(r0['w'][k], v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def _narre_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '')
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist, main as M
    from reviews4rec_amd.engine import NarreEngine
    r4dist.init_from_env()
    g = Golden('narre_e16')
    model, hp = build_model(g)
    model.train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    model.hyper_params['batch_size'] = 64                    # per-rank batch: every shard below fits it
    eng = M.make_engine(dict(hp, engine='auto', batch_size=64), model, dp=dp, rank=rank)
    assert type(eng) is NarreEngine and eng.dp is not None
    ses = []
    for step in range(3):
        data, y = g.batch(step % 2, 'cuda')
        sd, sy = r4dist.shard_batch(data, y, rank, world)
        ses.append(eng.train_step(sd, sy, n_global=int(y.shape[0]) if step != 1 else None).cpu().clone())
    torch.save({'w': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'se': ses},
               os.path.join(out_dir, 'n%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_dp2_native_narre_step_follows_the_reference_trajectory(tmp_path):
    """NARRE under data parallelism on the native step (gradients only per rank, flat all-reduce + Adam,
    the ranks' compact ID entries gathered into r4r_narre_rows_apply): 2 ranks x ragged shards == the
    reference's three single-process steps; replicas bit-identical."""
    sys.path.insert(0, TESTS)
    from helpers import Golden
    from test_oracle_golden import ill_conditioned
    port = _free_port()
    mp.spawn(_narre_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = Golden('narre_e16')
    r0 = torch.load(os.path.join(tmp_path, 'n0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'n1.pt'))
    for step in range(3):
        se = torch.cat([r0['se'][step], r1['se'][step]])
        torch.testing.assert_close(se, g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
    for k, v in g.params('w3').items():
        assert torch.equal(r0['w'][k], r1['w'][k]), k               # replicas stay bit-identical
        if not ill_conditioned(k):
            torch.testing.assert_close(r0['w'][k], v, rtol=1e-4, atol=2e-5, msg=lambda m: k + ': ' + m)


def _idnet_worker(rank, world, port, case, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)        # a protocol mismatch between the ranks must not hang the suite
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND=os.environ.get('R4R_TEST_BACKEND', 'gloo'),
                      R4R_DP_SINGLE='1' if world == 1 else '')
    from helpers import Golden
    from test_gpu_models import build_model
    from reviews4rec_amd import dist as r4dist, main as M
    from reviews4rec_amd.engine import IdNetEngine
    r4dist.init_from_env()
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    model.hyper_params['batch_size'] = 64                    # per-rank batch: every shard below fits it
    eng = M.make_engine(dict(hp, engine='auto', batch_size=64), model, dp=dp, rank=rank)
    assert isinstance(eng, IdNetEngine) and eng.dp is not None
    ses = []
    for step in range(3):
        data, y = g.batch(step % 2, 'cuda')
        sd, sy = r4dist.shard_batch(data, y, rank, world)
        if step == 2 and rank == 1:                          # an empty shard on one rank
            sd, sy = [None if d is None else d[:0] for d in sd], sy[:0]
        elif step == 2:
            sd, sy = data, y
        ses.append(eng.train_step(sd, sy, n_global=int(y.shape[0]) if step != 1 else None).cpu().clone())
    torch.save({'w': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'se': ses},
               os.path.join(out_dir, 'i%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('case', ['mf_full', 'neumf_gmf', 'neumf_mlp', 'neumf_full'])
def test_dp2_native_idnet_step_follows_the_reference_trajectory(tmp_path, case):
    """MF / GMF / MLP / NeuMF under data parallelism on the native step (r4r_idnet_step gradients only, one
    all-reduce of the flat dense gradient + the flat Adam, the ranks' compact ID rows gathered into
    r4r_idnet_rows_apply): 2 ranks x ragged shards (one of them empty in the last step) == the reference's three
    single-process steps; replicas bit-identical."""
    sys.path.insert(0, TESTS)
    from helpers import Golden
    port = _free_port()
    mp.spawn(_idnet_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    g = Golden(case)
    r0 = torch.load(os.path.join(tmp_path, 'i0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'i1.pt'))
    for step in range(3):
        se = torch.cat([r0['se'][step], r1['se'][step]])
        torch.testing.assert_close(se, g.arr('se%d' % step), rtol=1e-4, atol=1e-5)
    for k, v in g.params('w3').items():
        assert torch.equal(r0['w'][k], r1['w'][k]), k               # replicas stay bit-identical
        torch.testing.assert_close(r0['w'][k], v, rtol=1e-5, atol=5e-6, msg=lambda m: k + ': ' + m)


def _peer_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND='gloo', R4R_DP_SINGLE='1' if world == 1 else '')
    import time
    from reviews4rec_amd import _lib, dist as r4dist
    r4dist.init_from_env()
    n = 4 * 12345
    px = r4dist.PeerExchange(n, 'cuda')
    lib = _lib.lib()
    sums = []
    for epoch in range(1, 8):
        gen = torch.Generator().manual_seed(100 * epoch + rank)
        flat = torch.randn(n, generator=gen).cuda()
        if (epoch + rank) % 3 == 0:
            torch.cuda.synchronize()
            time.sleep(0.05)                                       # the ranks drift apart: the flags must hold them
        gathered = px.exchange(flat, epoch)
        p, m, v, gs = (torch.zeros(n, device='cuda') for _ in range(4))
        _lib.check(lib.r4r_adam_gathered(p.data_ptr(), gathered, world, gs.data_ptr(), m.data_ptr(), v.data_ptr(), n,
                                         1e-3, 0.9, 0.999, 1e-8, 0.0, 1, _lib.current_stream()), 'r4r_adam_gathered')
        sums.append(gs.cpu())
    px.check()
    px.close()
    torch.save(sums, os.path.join(out_dir, 'p%d.pt' % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2, 3])
def test_peer_exchange_gathers_every_ranks_buffer(tmp_path, world):
    """dist.PeerExchange alone: `world` processes on the one GPU map each other's fine-grained segments
    (r4r_peer_segment_*), push seven epochs of seeded buffers with the ranks drifting apart, and the rank-ordered
    sum r4r_adam_gathered forms from the gathered slots is the sum of the seeded buffers, bit for bit, on every rank."""
    port = _free_port()
    mp.spawn(_peer_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, 'p%d.pt' % r)) for r in range(world)]
    for epoch in range(1, 8):
        want = None
        for r in range(world):
            t = torch.randn(4 * 12345, generator=torch.Generator().manual_seed(100 * epoch + r))
            want = t if want is None else want + t
        for r in range(world):
            assert torch.equal(got[r][epoch - 1], want), (epoch, r)


# ---- RCCL itself, on the one GPU of the test box: a ONE-rank job (R4R_DP_SINGLE=1 keeps the data-parallel
# path on at world size 1).  The collectives move nothing, but every call the N > 1 job makes -- process-group
# init over torch's 'nccl' backend, broadcast, the flat all-reduce, all_gather of the compact rows, the MIN / MAX /
# SUM reductions of the epoch counts, barrier -- goes through RCCL with the real tensors (device placement,
# dtypes, contiguity: what gloo forgives and RCCL does not), and the trajectory must still be the reference's.
RCCL_CASES = {
    'deepconn-allreduce': (_engine_worker, ('deepconn_e20', 'allreduce'), 'e', 'deepconn_e20', 'se', 'w3'),
    'deepconn-gather': (_engine_worker, ('deepconn_e20', 'gather'), 'e', 'deepconn_e20', 'se', 'w3'),
    'deepconn-autotune': (_engine_worker, ('deepconn_e20', 'autotune'), 'e', 'deepconn_e20', 'se', 'w3'),
    'deepconn-peer': (_engine_worker, ('deepconn_e20', 'peer'), 'e', 'deepconn_e20', 'se', 'w3'),
    'mf_dot': (_mf_worker, ('mf_dot',), 'm', 'mf_dot', 'se', 'w3'),
    'transnetpp': (_transnet_worker, ('transnetpp_e16',), 't', 'transnetpp_e16', 'tn_se', 'tn_w3'),
    'deepconnpp': (_dcpp_worker, (), 'd', 'deepconnpp_e20', 'se', 'w3'),
    'narre': (_narre_worker, (), 'n', 'narre_e16', 'se', 'w3'),
    'neumf': (_idnet_worker, ('neumf_full',), 'i', 'neumf_full', 'se', 'w3'),
}


_RCCL_OK = []


def _need_rccl():
    """Skip (not fail) where RCCL itself cannot come up -- a one-rank process group + one all-reduce in a
    subprocess, once per session: what is under test here is this package's use of RCCL, not the box's fabric."""
    if not _RCCL_OK:
        import subprocess
        code = ("import os, torch, torch.distributed as d; os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', "
                "RANK='0', WORLD_SIZE='1'); d.init_process_group('nccl', rank=0, world_size=1); "
                "t = torch.ones(8, device='cuda'); d.all_reduce(t); torch.cuda.synchronize(); d.destroy_process_group()" % _free_port())
        try:
            r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=180)
            _RCCL_OK.append((r.returncode == 0, r.stderr[-300:]))
        except subprocess.TimeoutExpired:
            _RCCL_OK.append((False, 'timed out'))
    if not _RCCL_OK[0][0]:
        pytest.skip('RCCL does not initialise on this box: ' + _RCCL_OK[0][1])


@pytest.mark.parametrize('which', sorted(RCCL_CASES))
def test_rccl_one_rank_job_runs_every_data_parallel_path(tmp_path, monkeypatch, which):
    _need_rccl()
    sys.path.insert(0, TESTS)
    from helpers import Golden
    from test_oracle_golden import ill_conditioned
    worker, extra, tag, case, se_key, w_key = RCCL_CASES[which]
    monkeypatch.setenv('R4R_TEST_BACKEND', 'nccl')
    port = _free_port()
    args = (1, port) + extra[:1] + (str(tmp_path),) + extra[1:]
    mp.spawn(worker, args=args, nprocs=1, join=True)
    g = Golden(case)
    r0 = torch.load(os.path.join(tmp_path, '%s0.pt' % tag))
    for step in range(3):
        torch.testing.assert_close(r0['se'][step], g.arr('%s%d' % (se_key, step)), rtol=1e-4, atol=1e-5)
    loose = which == 'narre'
    for k, v in g.params(w_key).items():
        if not ill_conditioned(k):
            torch.testing.assert_close(r0['w'][k], v, rtol=1e-4 if loose else 1e-5, atol=2e-5 if loose else 5e-6,
                                       msg=lambda m: k + ': ' + m)


def test_rccl_one_rank_bench_line(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, backend 'nccl' = RCCL), with one rank:
    the data-parallel step, the exchange autotune, the strong legs and the replica check all run over RCCL."""
    import json
    import subprocess
    _need_rccl()
    env = dict(os.environ, R4R_DP_SINGLE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('R4R_DIST_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '10',
           '--warmup', '3', '--no-cpu-baseline', '--strong-leg', '1024']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    cfg = line['config']
    assert cfg['dist_backend'] == 'nccl' and cfg['rccl_ranks'] == 1 and cfg['replicas_identical'] is True
    assert cfg['dp_exchange'] in ('allreduce', 'gather') and set(cfg['dp_exchange_ms']) == {'allreduce', 'gather'}
    assert line['value'] > 0 and line['strong']['global_batch'] == 1024


@pytest.mark.parametrize('workload,strong', [('cfg2_mfdot_electronics', 1024), ('cfg4_narre_kindle', 1024),
                                             ('cfg5_transnetpp_synthetic', 1024),
                                             # beyond the block forms' entry limits: the generic exchange of gathered entries
                                             ('cfg5_transnetpp_synthetic', 4096), ('cfg2_mfdot_electronics', 4096)])
def test_two_rank_bench_line_of_the_id_table_families(workload, strong):
    """bench.py --gpus 2 (two ranks sharing the one GPU over gloo) for the families whose data-parallel step pads
    every rank's shard to hyper_params['batch_size']: the weak line AND a strong leg whose per-rank batch (512) exceeds
    the weak one (128) -- the leg has to carry its own shard padding into the engine (it raised until round 4)."""
    import json
    import subprocess
    env = dict(os.environ, R4R_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
           '--warmup', '2', '--ramp', '4', '--no-cpu-baseline', '--workload', workload, '--strong-leg', str(strong)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['config']['replicas_identical'] is True
    legs = line.get('strong_legs') or [line.get('strong')]
    assert legs and legs[0]['global_batch'] == strong and legs[0].get('ratings_per_s', 0) > 0, legs


def test_bare_two_gpu_bench_command_launches_its_own_ranks():
    """Exactly `python bench.py --gpus 2 --steps 20 --warmup 5` -- no torchrun around it, no RANK / WORLD_SIZE in the
    environment: bench.py re-launches itself as a two-rank job (here both ranks share the one GPU over
    gloo), rank 0 prints the one JSON line with the weak value, the strong legs and the replica check."""
    import json
    import subprocess
    env = dict(os.environ, R4R_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'R4R_DP_SINGLE'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]               # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    cfg = line['config']
    assert line['n_gpus'] == 2 and line['steps'] == 20 and line['warmup_requested'] == 5 and line['value'] > 0
    assert line['scaling'] == 'weak' and cfg['ratings_per_step'] == 256 and cfg['parallelism'] == 'dp2'
    assert cfg['rccl_ranks'] == 2 and cfg['dist_backend'] == 'gloo' and cfg['replicas_identical'] is True
    assert cfg['dp_exchange'] in ('allreduce', 'gather')
    legs = [l for l in line['strong_legs'] if 'skipped' not in l]
    assert [l['global_batch'] for l in legs] == [1024, 8192, 32768] and all(l['ratings_per_s'] > 0 for l in legs)
    assert 'configs' not in line and 'cpu_baseline' not in line     # (N = 1 only)


def test_more_ranks_than_gpus_over_rccl_fails_with_one_line():
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'R4R_DIST_BACKEND', 'R4R_DP_SINGLE'):
        env.pop(k, None)
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and ('--gpus %d but %d GPU' % (n, n - 1)) in out.stderr, out.stderr[-1000:]


def test_peer_waits_are_bounded_when_a_rank_never_arrives():
    """The peer-mapped exchanges wait for the ranks' flags INSIDE kernels: a rank that never raises its flag must cost
    the timeout, not the GPU.  One process plays rank 0 of two (both 'ranks' map the same segment): r4r_peer_wait and
    r4r_mf_grad_push -> r4r_mf_apply_peer (every workgroup of the update launch polls) return after ~the timeout and
    name the missing rank in the timed_out word."""
    import ctypes
    import time
    from reviews4rec_amd import _lib
    from reviews4rec_amd._lib import ptr
    lib = _lib.lib()
    dev = torch.device('cuda')
    n_users, n_items, D, B = 300, 200, 16, 8
    nbytes = lib.r4r_mf_dp_block_bytes(B, D)
    FLAGS = 4096
    seg, handle = ctypes.c_void_p(), (ctypes.c_uint8 * 64)()
    _lib.check(lib.r4r_peer_segment_create(FLAGS + 2 * nbytes, ctypes.byref(seg), handle), 'r4r_peer_segment_create')
    try:
        local = torch.zeros(16, dtype=torch.int32, device=dev)              # [0] arrival counter, [1] timed_out
        # the bare wait: flag 1 of 2 never reaches epoch 1
        t0 = time.perf_counter()
        _lib.check(lib.r4r_peer_wait(seg.value, 2, 1, local.data_ptr() + 4, 0.2, _lib.current_stream()), 'r4r_peer_wait')
        torch.cuda.synchronize()
        assert 0.15 < time.perf_counter() - t0 < 5.0
        assert int(local[1].item()) in (1, 2)                               # 1 + the rank that was missing (0 and 1 both are)
        local.zero_()
        # the MF step: rank 0 pushes, "rank 1" never does
        torch.manual_seed(0)
        p = [torch.randn(n_users, D, device=dev) * 0.1, torch.randn(n_items, D, device=dev) * 0.1,
             torch.zeros(n_users, device=dev), torch.zeros(n_items, device=dev), torch.zeros(1, device=dev)]
        m, v = [torch.zeros_like(t) for t in p], [torch.zeros_like(t) for t in p]
        P5 = lambda ts: (ctypes.c_uint64 * 5)(*[t.data_ptr() for t in ts])   # noqa: E731
        uid = torch.randint(0, n_users, (B,), device=dev)
        iid = torch.randint(0, n_items, (B,), device=dev)
        y = torch.rand(B, device=dev) * 4 + 1
        pred, se, sse = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.zeros(1, device=dev)
        ws = torch.zeros(lib.r4r_mf_ws_bytes(2 * B, D, n_users, n_items), dtype=torch.uint8, device=dev)
        u64 = lambda vals: (ctypes.c_uint64 * 16)(*(vals + [0] * (16 - len(vals))))   # noqa: E731
        dst, flg = u64([seg.value + FLAGS, seg.value + FLAGS]), u64([seg.value, seg.value])
        _lib.check(lib.r4r_mf_grad_push(ptr(uid), ptr(iid), ptr(y), P5(p), None, None, n_users, n_items, D, ptr(pred), ptr(se),
                                        None, B, B, 0.0, 1, 1, 0, 1.0 / (2 * B), None, 1, 0, 0.002, 0.9, 0.999, 1e-8, 1e-6, 1,
                                        dst, flg, local.data_ptr(), 0, 2, 1, _lib.current_stream()), 'r4r_mf_grad_push')
        t0 = time.perf_counter()
        _lib.check(lib.r4r_mf_apply_peer(seg.value + FLAGS, 2, B, P5(p), P5(m), P5(v), n_users, n_items, D, ptr(ws), ws.numel(),
                                         1, 0, 1, ptr(se), B, ptr(sse), 0.002, 0.9, 0.999, 1e-8, 1e-6, 1,
                                         seg.value, 1, local.data_ptr() + 4, 0.2, _lib.current_stream()), 'r4r_mf_apply_peer')
        torch.cuda.synchronize()
        assert 0.15 < time.perf_counter() - t0 < 10.0
        assert int(local[1].item()) == 2                                    # rank 1 never raised its flag
    finally:
        torch.cuda.synchronize()
        _lib.check(lib.r4r_peer_segment_destroy(seg), 'r4r_peer_segment_destroy')
