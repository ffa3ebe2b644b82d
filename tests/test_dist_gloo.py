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
