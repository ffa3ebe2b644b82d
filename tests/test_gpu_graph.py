"""hipGraph-captured module-engine step (reviews4rec_amd.graph.GraphedStep): the replayed
steps must reproduce the reference-generated golden trajectories, draw fresh dropout masks on
every replay, and fall back to the eager path for a ragged batch."""
import pytest
import torch

from helpers import Golden
from test_gpu_models import build_model
from test_oracle_golden import ill_conditioned

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('case', ['deepconn_e20', 'deepconnpp_e20', 'narre_e16', 'mf_dot'])
def test_graphed_step_reproduces_reference_trajectory(case):
    from reviews4rec_amd import ops
    from reviews4rec_amd.graph import GraphedStep
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    ops.DropoutState.device_counter = None
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    opt = Adam(model.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])
    d0, y0 = g.batch(0, DEV)
    step = GraphedStep(model, MSELoss(hp), opt, d0, y0)
    total = 0.0
    for k in range(3):                                       # batch 1 is ragged: eager fallback
        data, y = g.batch(k % 2, DEV)
        se = step(data, y)
        torch.testing.assert_close(se.detach().cpu(), g.arr('se%d' % k), rtol=1e-4, atol=1e-5)
        total += float(g.arr('se%d' % k).sum())
        if k in (0, 2):
            sd = model.state_dict()
            for name, v in g.params('w%d' % (k + 1)).items():
                if not ill_conditioned(name):
                    torch.testing.assert_close(sd[name].cpu(), v, rtol=1e-5, atol=5e-6, msg=lambda m: name + ': ' + m)
    torch.testing.assert_close(step.sse.cpu(), torch.tensor(total), rtol=1e-5, atol=1e-4)
    ops.DropoutState.device_counter = None


def test_graph_replays_draw_fresh_dropout_masks():
    from reviews4rec_amd import ops
    from reviews4rec_amd.graph import GraphedStep
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    ops.DropoutState.device_counter = None
    ops.DropoutState.manual_seed(3)
    g = Golden('deepconnpp_e20')
    model, hp = build_model(g, dropout=0.5)
    model.train()
    opt = Adam(model.parameters(), lr=0.0, weight_decay=0.0)     # frozen weights: only the masks change
    d0, y0 = g.batch(0, DEV)
    step = GraphedStep(model, MSELoss(hp), opt, d0, y0)
    a = step(d0, y0).detach().clone()
    b = step(d0, y0).detach().clone()
    assert not torch.equal(a, b)                              # the device-side Philox offset advanced
    assert int(ops.DropoutState.device_counter.item()) > 0
    ops.DropoutState.device_counter = None


def test_host_loop_with_graph_engine_matches_reference_metric():
    """main.train with hyper_params['engine'] == 'graph': the epoch metric round(sum SE / N, 4)
    (main.py:66) over the two golden batches (the second is ragged -> eager fallback)."""
    from reviews4rec_amd import main as M, ops
    from reviews4rec_amd.loss import MSELoss
    ops.DropoutState.device_counter = None
    g = Golden('narre_e16')
    model, hp = build_model(g)
    hp['engine'] = 'graph'

    class Reader:
        def iter(self, eval=False):
            for k in (0, 1):
                yield g.batch(k, DEV)

    holder = M._GraphHolder()
    metrics = M.train(model, MSELoss(hp), M.make_optimizer(hp, model), Reader(), hp, graph=holder)
    n = g.arr('y0').shape[0] + g.arr('y1').shape[0]
    expected = round(float(g.arr('se0').sum() + g.arr('se1').sum()) / n, 4)
    assert holder.step is not None and metrics['MSE'] == pytest.approx(expected, abs=2e-4)
    ops.DropoutState.device_counter = None


@pytest.mark.parametrize('case', ['transnet_e16', 'transnetpp_e16'])
def test_graphed_transnet_three_optimiser_step(case):
    """The 3-optimiser TransNet step captured in one hipGraph reproduces the reference-generated
    trajectory (steps 0 and 2 replay the graph, step 1 is the ragged batch -> eager)."""
    from reviews4rec_amd import main as M, ops
    from reviews4rec_amd.graph import GraphedStep
    from reviews4rec_amd.loss import MSELoss
    ops.DropoutState.device_counter = None
    g = Golden(case)
    model, hp = build_model(g)
    model.train()
    d0, y0 = g.batch(0, DEV)
    step = GraphedStep(model, MSELoss(hp), M.make_optimizer(hp, model), d0, y0)
    for k in range(3):
        data, y = g.batch(k % 2, DEV)
        se = step(data, y)
        torch.testing.assert_close(se.detach().cpu(), g.arr('tn_se%d' % k), rtol=1e-4, atol=1e-5)
        if k in (0, 2):
            sd = model.state_dict()
            for name, v in g.params('tn_w%d' % (k + 1)).items():
                torch.testing.assert_close(sd[name].cpu(), v, rtol=1e-5, atol=1e-5, msg=lambda m: name + ': ' + m)
    ops.DropoutState.device_counter = None
