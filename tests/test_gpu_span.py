"""Spans (include/r4r.h: r4r_*_span, csrc/span.hip; engine._Spans): an epoch whose full batches are enqueued K steps
per host call must leave EXACTLY what the same epoch leaves when main.train iterates the loader and calls train_step
per batch -- the same kernels with the same arguments, so the same bits: every parameter, every Adam moment, the step
count, the dropout stream position, the epoch metric.  Every native family, dropout on (the Philox masks are a function
of (seed, offset), so both runs draw the same ones), ragged last batch, two epochs (the ring restarts), the conv rule's
probe step in the middle of the first epoch (auto), span lengths that do and do not divide the batch groups.

The loop the spans replace: /root/reference/main.py:23-60 over data.py:250-372 / data_fast.py:99-109."""
import copy
import os
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def tiny_corpus(ratings=1500, users=300, items=120, vocab=400, seed=11):
    """A small dataset in the reference's pickled schema (what data.load_data unpickles)."""
    rng = np.random.default_rng(seed)
    seen, pairs = set(), []
    while len(pairs) < ratings:
        u, i = int(rng.zipf(1.3) % users), int(rng.zipf(1.3) % items)
        if (u, i) not in seen:
            seen.add((u, i))
            pairs.append([u, i, float(rng.integers(1, 6))])
    user_reviews = {u: [] for u in range(users)}
    item_reviews = {i: [] for i in range(items)}
    tiui = {}
    for u, i, r in pairs:
        rev = rng.integers(1, vocab, size=int(rng.integers(3, 40))).tolist()
        tiui.setdefault(u, {})[i] = [len(user_reviews[u]), len(item_reviews[i])]
        user_reviews[u].append(rev)
        item_reviews[i].append(rev)
    return dict(train=pairs, user_reviews=user_reviews, item_reviews=item_reviews, this_index_user_item=tiui)


def hyper(mt, **kw):
    hp = dict(model_type=mt, batch_size=32, input_length=64, narre_num_reviews=10, narre_num_words=12, total_users=300,
              total_items=120, latent_size=8, word_embed_size=16, dropout=0.5, lr=0.002, weight_decay=1e-6, vocab=400,
              total_words=400, engine='native', dataset='span_test')
    hp['word_vectors'] = (np.random.default_rng(3).random((400, 16), dtype=np.float32) - 0.5) * 0.2
    hp.update(kw)
    return hp


def build(mt, corpus, **kw):
    import reviews4rec_amd
    from reviews4rec_amd import main as M
    from reviews4rec_amd.data import DataLoader
    from reviews4rec_amd.utils import xavier_init
    hp = hyper(mt, **kw)
    reader = DataLoader(hp, corpus['train'], corpus['user_reviews'], corpus['item_reviews'], None,
                        this_index_user_item=corpus['this_index_user_item'], device=DEV)
    torch.manual_seed(5)
    model = reviews4rec_amd.get_model_class(mt)(hp)
    xavier_init(model)
    model = model.cuda()
    return hp, reader, model, M.make_engine(hp, model)


def run_epochs(mt, corpus, spans, epochs=2, span_steps=None, **kw):
    from reviews4rec_amd import main as M
    from reviews4rec_amd.loss import MSELoss
    hp, reader, model, engine = build(mt, corpus, **kw)
    hp['spans'] = spans
    if span_steps:
        engine.SPAN_STEPS = span_steps
    calls = {'span': 0, 'step': 0}
    if spans:
        inner_span, inner_step = engine._span, engine.train_step

        def counting_span(*a, **k):
            calls['span'] += 1
            return inner_span(*a, **k)

        def counting_step(*a, **k):
            calls['step'] += 1
            return inner_step(*a, **k)
        engine._span, engine.train_step = counting_span, counting_step
    metrics = [M.train(model, MSELoss(hp), None, reader, hp, engine=engine) for _ in range(epochs)]
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = engine.state_dict()
    torch.cuda.synchronize()
    return metrics, state, opt, calls, engine


def same(a, b, what):
    if torch.is_tensor(a):
        assert torch.equal(a, b), what
    elif isinstance(a, dict):
        assert set(a) == set(b), what
        for k in a:
            same(a[k], b[k], '%s/%s' % (what, k))
    else:
        assert a == b, what


FAMILIES = ['deepconn', 'deepconn++', 'NARRE', 'transnet', 'transnet++', 'MF_dot', 'bias_only', 'MF']


@pytest.mark.parametrize('mt', FAMILIES + ['NeuMF/GMF', 'NeuMF/MLP', 'NeuMF/NeuMF'])
def test_span_epochs_leave_the_bits_of_the_per_step_loop(mt):
    corpus = tiny_corpus()
    kw = {}
    if '/' in mt:                                            # the three stage models of main_NeuMF (NeuMF.py), one engine variant each
        mt, kw['neumf_stage'] = mt.split('/')
    want_m, want_w, want_o, _, _ = run_epochs(mt, corpus, spans=False, **kw)
    got_m, got_w, got_o, calls, engine = run_epochs(mt, corpus, spans=True, span_steps=7, **kw)
    assert calls['span'] >= 2 * (46 // 7), calls               # the epoch really went through the span entry
    assert calls['step'] <= 2 * 3, calls                       # ... but for the ragged tail and the rule's probes
    assert got_m == want_m
    same(got_w, want_w, 'weights')
    same(got_o, want_o, 'optimiser state')
    assert engine.step_count == 2 * 47


@pytest.mark.parametrize('mt,steps', [('deepconn', 64), ('NARRE', 33), ('MF_dot', 1), ('transnet++', 46)])
def test_span_length_does_not_matter(mt, steps):
    corpus = tiny_corpus(seed=12)
    _, want_w, want_o, _, _ = run_epochs(mt, corpus, spans=True, span_steps=5, epochs=1)
    _, got_w, got_o, _, _ = run_epochs(mt, corpus, spans=True, span_steps=steps, epochs=1)
    same(got_w, want_w, 'weights')
    same(got_o, want_o, 'optimiser state')


def test_span_batches_are_the_loader_s_batches():
    """The ring's batch b (r4r_span_build + r4r_span_batch) is what iter() yields as its b-th batch, slot for slot."""
    from reviews4rec_amd import _lib
    corpus = tiny_corpus(seed=13)
    for mt in ('deepconn', 'NARRE'):
        hp, reader, _, _ = build(mt, corpus)
        desc = reader.span_descriptor()
        lib = _lib.lib()
        doc = int(np.prod(desc.doc_shape))
        B = desc.batch_size
        for b, (data, y) in enumerate(reader.iter()):
            if b >= desc.full_batches:
                break
            _lib.check(lib.r4r_span_build(desc.words, b, ctypes.byref(desc.built), _lib.current_stream()), 'build')
            slots = (ctypes.c_uint64 * 8)()
            _lib.check(lib.r4r_span_batch(desc.words, b, slots), 'batch')
            torch.cuda.synchronize()
            base = desc.ring.data_ptr()
            for s, width in ((0, doc), (1, 10), (2, 10), (3, doc), (4, doc)):
                at = (slots[s] - base) // 8
                mine = desc.ring[at:at + B * width].view(data[s].shape)
                assert torch.equal(mine, data[s]), (mt, b, s)
            assert slots[5] == data[5].data_ptr() and slots[6] == data[6].data_ptr() and slots[7] == y.data_ptr()


def test_span_arguments_are_checked_before_anything_runs():
    from reviews4rec_amd import _lib
    corpus = tiny_corpus(seed=14)
    hp, reader, model, engine = build('deepconn', corpus)
    desc = reader.span_descriptor()
    before = copy.deepcopy(engine.state_dict())
    with pytest.raises(RuntimeError, match='span'):
        engine._span(desc, desc.full_batches - 1, 2, False)    # runs past the last full batch
    with pytest.raises(RuntimeError, match='span'):
        engine._span(desc, desc.full_batches - 2, 2, True)     # nothing left to announce
    same(engine.state_dict(), before, 'state after rejected spans')
    lib = _lib.lib()
    assert lib.r4r_span_batch(None, 0, (ctypes.c_uint64 * 8)()) != 0
    assert b'span' in lib.r4r_last_error()


# ---- data parallel: the headline family's span with the exchange issued from C (r4r_deepconn_span_dp), as a ONE-rank
# RCCL job (the test box has one GPU; RCCL wants one per rank): every call of the N > 1 path is made, with the real
# communicator, and an epoch through DP spans must leave the bits of the per-step DP loop.
def _dp_span_worker(rank, world, port, out_dir):
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), R4R_DIST_BACKEND='nccl', R4R_DP_SINGLE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import reviews4rec_amd
    from reviews4rec_amd import dist as r4dist, main as M
    from reviews4rec_amd.data import DataLoader
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.utils import xavier_init
    from test_gpu_span import hyper, tiny_corpus
    r4dist.init_from_env()
    corpus = tiny_corpus(seed=15)
    res = {}
    for exchange in ('allreduce', 'gather'):
        os.environ['R4R_DP_EXCHANGE'] = exchange
        got = {}
        for spans in (False, True):
            hp = hyper('deepconn', spans=spans)
            reader = DataLoader(hp, corpus['train'], corpus['user_reviews'], corpus['item_reviews'], None,
                                this_index_user_item=corpus['this_index_user_item'], device='cuda')
            torch.manual_seed(5)
            model = reviews4rec_amd.get_model_class('deepconn')(hp)
            xavier_init(model)
            model = model.cuda()
            dp = r4dist.DataParallel(model)
            dp.broadcast_parameters()
            engine = M.make_engine(hp, model, dp=dp, rank=rank)
            engine.SPAN_STEPS = 9
            calls = [0]
            inner = engine._span

            def counting(*a, _inner=inner, **k):
                calls[0] += 1
                return _inner(*a, **k)
            engine._span = counting
            metrics = [M.train(model, MSELoss(hp), None, reader, hp, engine=engine, dp=dp) for _ in range(2)]
            torch.cuda.synchronize()
            got[spans] = (metrics, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                          engine.flat_m.cpu().clone(), engine.flat_v.cpu().clone(), engine.step_count, engine.offset, calls[0])
            dp.close()
        a, b = got[False], got[True]
        res[exchange] = dict(same_metrics=a[0] == b[0], same_weights=all(torch.equal(a[1][k], b[1][k]) for k in a[1]),
                             same_moments=bool(torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])),
                             same_counts=(a[4], a[5]) == (b[4], b[5]), span_calls=b[6], per_step_span_calls=a[6])
    torch.save(res, os.path.join(out_dir, 'dp_span.pt'))
    torch.distributed.destroy_process_group()


def test_data_parallel_span_leaves_the_bits_of_the_per_step_exchange(tmp_path):
    import socket
    import torch.multiprocessing as mp
    from test_gpu_dist import _need_rccl
    _need_rccl()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_span_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(tmp_path, 'dp_span.pt'))
    for exchange in ('allreduce', 'gather'):
        r = res[exchange]
        assert r['span_calls'] >= 2 * (46 // 9) and r['per_step_span_calls'] == 0, r
        assert r['same_metrics'] and r['same_weights'] and r['same_moments'] and r['same_counts'], (exchange, r)
