"""Data parallelism without a cluster: 2 processes, gloo backend, CPU.  Asserts
DP(2) == the single-process step on the concatenated batch (dropout 0), with ragged
shards, parameters that never receive a gradient (DeepCoNN 'deepconn' mode), and
the flat-bucket path the fused engine uses."""
import copy
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, 'tests')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, n_rows, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from helpers import Golden, OracleModule
    from reviews4rec_amd import dist as r4dist
    from reviews4rec_amd.loss import MSELoss
    r4dist.init_from_env(backend='gloo')
    g = Golden(case)
    data, y = g.batch(0)
    data, y = [d[:n_rows] for d in data], y[:n_rows]
    model = OracleModule(g.hp, params=g.params())
    if rank == 1:                                           # replicas must start from rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp = r4dist.DataParallel(model)
    dp.broadcast_parameters()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=g.hp['lr'],
                           weight_decay=g.hp['weight_decay'])
    model.train()
    for _ in range(2):
        sd, sy = r4dist.shard_batch(data, y, rank, world)
        n_global = dp.global_count(sy.shape[0], sy.device)
        assert n_global == n_rows
        opt.zero_grad()
        if sy.shape[0] > 0:
            se = MSELoss(g.hp)(model(sd), sy, return_mean=False)
            (se.sum() * dp.loss_scale(sy.shape[0], n_global)).backward()
        dp.allreduce_grads()
        opt.step()
    torch.save({k: v.detach().clone() for k, v in model.as_dict().items()}, os.path.join(out_dir, 'r%d.pt' % rank))
    # flat-bucket path (what DeepCoNNEngine calls): sum of rank-dependent vectors
    flat = torch.full((7,), float(rank + 1))
    dp.allreduce_flat(flat)
    assert torch.equal(flat, torch.full((7,), 3.0))
    dist.destroy_process_group()


@pytest.mark.parametrize('case,n_rows', [('deepconn_e20', 5), ('mf_dot', 13), ('narre_e16', 4), ('deepconn_e20', 1)])
def test_dp2_equals_single_process(tmp_path, case, n_rows):
    sys.path.insert(0, TESTS)
    from helpers import Golden, OracleModule
    from reviews4rec_amd.loss import MSELoss
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, n_rows, str(tmp_path)), nprocs=2, join=True)
    g = Golden(case)
    data, y = g.batch(0)
    data, y = [d[:n_rows] for d in data], y[:n_rows]
    ref = OracleModule(g.hp, params=g.params())
    opt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=g.hp['lr'],
                           weight_decay=g.hp['weight_decay'])
    ref.train()
    for _ in range(2):
        opt.zero_grad()
        torch.mean(MSELoss(g.hp)(ref(data), y, return_mean=False)).backward()
        opt.step()
    r0 = torch.load(os.path.join(tmp_path, 'r0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'r1.pt'))
    from test_oracle_golden import ill_conditioned
    for k, v in ref.as_dict().items():
        assert torch.equal(r0[k], r1[k]), k                 # replicas stay bit-identical
        if ill_conditioned(k):                              # true gradient 0: Adam amplifies rounding noise
            continue
        torch.testing.assert_close(r0[k], v.detach(), rtol=1e-5, atol=2e-6, msg=lambda m: k + ': ' + m)


def test_shard_bounds_cover_the_batch():
    from reviews4rec_amd.dist import shard_bounds
    for n in (0, 1, 5, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _sparse_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from reviews4rec_amd import dist as r4dist, ops
    r4dist.init_from_env(backend='gloo')

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.zeros(11, 3))
            self.bias = torch.nn.Parameter(torch.zeros(11))
            self.dense = torch.nn.Parameter(torch.zeros(4))

    def rebuild(idx, g, R, D, out):                        # CPU stand-in for r4r_embed_scatter_add_ordered
        out.zero_()
        keep = idx >= 0
        out.index_add_(0, idx[keep], g[keep])
        return out

    model = Tiny()
    dp = r4dist.DataParallel(model, rebuild_fn=rebuild)
    gen = torch.Generator().manual_seed(100 + rank)
    n = 5 if rank == 0 else 3                               # ragged shards -> padded exchange
    for step in range(2):
        idx = torch.randint(0, 11, (n,), generator=gen)
        g = torch.randn((n, 3), generator=gen)
        gb = torch.randn((n, 1), generator=gen)
        # what ops.EmbedGather / BiasHead backward record in capture mode
        ops.SparseGradCapture.record(model.table.data_ptr(), idx, g)
        ops.SparseGradCapture.record(model.bias.data_ptr(), idx, gb)
        model.table.grad = torch.empty_like(model.table)     # the uninitialised placeholders
        model.bias.grad = torch.empty_like(model.bias)
        model.dense.grad = torch.full((4,), float(rank + 1))
        dp.allreduce_grads()
        torch.save(dict(table=model.table.grad.clone(), bias=model.bias.grad.clone(), dense=model.dense.grad.clone(),
                        idx=idx, g=g, gb=gb), os.path.join(out_dir, 's%d_r%d.pt' % (step, rank)))
    assert not ops.SparseGradCapture.contributions          # consumed
    ops.SparseGradCapture.active = False
    dist.destroy_process_group()


def test_sparse_table_gradients_are_exchanged_as_compact_lists(tmp_path):
    """ID-table / bias gradients travel as all-gathered (row-id, grad-row) lists; every
    rank rebuilds the same dense gradient = sum of all ranks' contributions; dense-layer gradients
    still go through the flat all-reduce bucket."""
    port = _free_port()
    mp.spawn(_sparse_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for step in range(2):
        r0 = torch.load(os.path.join(tmp_path, 's%d_r0.pt' % step))
        r1 = torch.load(os.path.join(tmp_path, 's%d_r1.pt' % step))
        want_t = torch.zeros(11, 3).index_add_(0, torch.cat([r0['idx'], r1['idx']]), torch.cat([r0['g'], r1['g']]))
        want_b = torch.zeros(11, 1).index_add_(0, torch.cat([r0['idx'], r1['idx']]), torch.cat([r0['gb'], r1['gb']]))
        for r in (r0, r1):
            torch.testing.assert_close(r['table'], want_t)
            torch.testing.assert_close(r['bias'], want_b.view(-1))
            assert torch.equal(r['dense'], torch.full((4,), 3.0))
        assert torch.equal(r0['table'], r1['table']) and torch.equal(r0['bias'], r1['bias'])


def _host_loop_worker(rank, world, port, out_dir):
    """train_complete under data parallelism on the CPU (oracle stand-in for the model): the stage model
    is re-bound to the exchange, global batch sizes come from ONE collective per epoch, only rank 0
    writes the best-model file and nobody reads it early, and the three-optimiser TransNet step refuses
    to run without its native engine."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from helpers import Golden, OracleModule
    from reviews4rec_amd import dist as r4dist, main as M
    from reviews4rec_amd.loss import MSELoss
    r4dist.init_from_env(backend='gloo')
    g = Golden('mf_dot')
    hp = dict(g.hp, epochs=2, dataset='Tiny', log_file=os.path.join(out_dir, 'log%d' % rank),
              model_path=os.path.join(out_dir, 'best.pt'), engine='module', batch_size=4)
    data, y = g.batch(0)                                      # 13 ratings -> global batches of 8, ragged tail

    class Reader:
        """This rank's contiguous shard of every global batch of 2 * batch_size ratings."""
        def __init__(self):
            self.batches = []
            for s in range(0, y.shape[0], 8):
                gd, gy = [None if d is None else d[s:s + 8] for d in data], y[s:s + 8]
                self.batches.append(r4dist.shard_batch(gd, gy, rank, world))
        def __len__(self):
            return len(self.batches)
        def batch_sizes(self):
            return [int(b[1].shape[0]) for b in self.batches]
        def iter(self, eval=False):
            return iter(self.batches)

    def rebuild(idx, gr, R, D, out):                          # CPU stand-in for r4r_embed_scatter_add_ordered
        out.zero_()
        keep = idx >= 0
        out.index_add_(0, idx[keep], gr[keep])
        return out

    seen = {}

    class Model(OracleModule):                                 # train_complete re-creates Model(hyper_params)
        def __init__(self, hyper_params):
            super().__init__(hyper_params, params=g.params())

    model = Model(hp)
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)                                    # rebind() must broadcast rank 0's weights
    stale = Model(hp)
    dp = r4dist.DataParallel(stale, rebuild_fn=rebuild, sparse_tables=False)
    dp._active, dp._bucket = [0], torch.zeros(1)               # cached decisions about ANOTHER model
    reader = Reader()
    counts = dp.epoch_counts(reader)
    assert counts == [8, 5], counts
    assert dp.gather_ints([rank, 10 * rank]) == [[0, 0], [1, 10]]
    # the reference's torch.optim.Adam stands in for the fused HIP Adam on this CPU rig
    real_make = M.make_optimizer
    M.make_optimizer = lambda hyper_params, m: torch.optim.Adam(
        [p for p in m.parameters() if p.requires_grad], lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'])
    M.is_cuda_available = False
    best = M.train_complete(hp, Model, reader, reader, {}, {}, model, review=False, dp=dp)
    M.make_optimizer = real_make
    assert dp.model is model and dp._active is not None and len(dp.params) == len(list(model.parameters()))
    torch.save({k: v.detach().clone() for k, v in model.as_dict().items()}, os.path.join(out_dir, 'hl%d.pt' % rank))
    torch.save({k: v.detach().clone() for k, v in best.as_dict().items()}, os.path.join(out_dir, 'best%d.pt' % rank))
    # ranks with different batch counts would hang the shorter rank's missing step: an error up front
    class Short(Reader):
        def batch_sizes(self):
            return super().batch_sizes()[:1 + rank]
    try:
        dp.epoch_counts(Short())
        raise AssertionError('unequal batch counts went unnoticed')
    except RuntimeError as e:
        assert 'batches this epoch' in str(e)
    # TransNet's op-by-op three-optimiser step has no gradient exchange: refused under DP
    try:
        M.train(model, MSELoss(hp), [None] * 4, reader, dict(hp, model_type='transnet'), engine=None, dp=dp)
        raise AssertionError('TransNet trained under DP without the native step')
    except RuntimeError as e:
        assert 'native step' in str(e)
    dist.destroy_process_group()


def test_host_loop_under_data_parallelism(tmp_path):
    sys.path.insert(0, TESTS)
    from helpers import Golden, OracleModule
    from reviews4rec_amd.loss import MSELoss
    port = _free_port()
    mp.spawn(_host_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = Golden('mf_dot')
    data, y = g.batch(0)
    ref = OracleModule(g.hp, params=g.params())
    opt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=g.hp['lr'],
                           weight_decay=g.hp['weight_decay'])
    ref.train()
    for epoch in range(2):
        for s in range(0, y.shape[0], 8):
            opt.zero_grad()
            torch.mean(MSELoss(g.hp)(ref([None if d is None else d[s:s + 8] for d in data]), y[s:s + 8],
                                     return_mean=False)).backward()
            opt.step()
    r0, r1 = torch.load(os.path.join(tmp_path, 'hl0.pt')), torch.load(os.path.join(tmp_path, 'hl1.pt'))
    b0, b1 = torch.load(os.path.join(tmp_path, 'best0.pt')), torch.load(os.path.join(tmp_path, 'best1.pt'))
    for k, v in ref.as_dict().items():
        assert torch.equal(r0[k], r1[k]), k                  # replicas identical although rank 1 started elsewhere
        assert torch.equal(b0[k], b1[k]), k                  # both ranks reloaded the same complete file
        torch.testing.assert_close(r0[k], v.detach(), rtol=1e-5, atol=2e-6, msg=lambda m: k + ': ' + m)
    assert not os.path.exists(os.path.join(tmp_path, 'best.pt.tmp'))
