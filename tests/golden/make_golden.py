#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runnable only where /root/reference exists (the build container).  It imports
the reference's pytorch_models / loss / utils unmodified, builds tiny models,
applies utils.xavier_init exactly like main.py:375-377, and records

  w/<name>            state_dict after xavier_init
  b<k>/<slot>, y<k>   batches (the 7-slot list of data_fast.py:101-109)
  eval<k>             model.eval() outputs
  neg_*               a negatives-shaped [B, 6, ...] batch and its eval output
  se<k>               train-mode (dropout=0) per-example squared error BEFORE step k+1
  g0/<name>           gradients of mean(SE) on batch 0 at the initial weights
  w1/.., w3/..        weights after 1 and 3 torch.optim.Adam steps (main.py:94-96)
  m3/.., v3/..        Adam moments after 3 steps
  tn_se<k>, tn_aux<k>, tn_w1/.., tn_w3/..   TransNet only: the 3-optimiser step of main.py:35-53
                      run with write-through (torch-0.4 style) optimisers, see DataAdam
  init_gmf/.., init_mlp/..   neumf_full only: the GMF / MLP weights NeuMF.init (NeuMF.py:100-118)
                      was fed with; w/.. is its result

Only DATA is written (npz); no reference source travels.  Usage:
    cd /tmp && python /root/repo/tests/golden/make_golden.py
"""
import os
import pickle
import sys
import tempfile

sys.dont_write_bytecode = True
REF = '/root/reference'
sys.path.insert(0, REF)

import numpy as np
import torch

torch.set_num_threads(1)
OUT = os.path.dirname(os.path.abspath(__file__))


def base_hp(model_type, **kw):
    hp = {
        'dataset': 'Tiny', 'k_core': 5, 'percent_reviews_to_keep': 100,
        'weight_decay': 1e-6, 'lr': 0.002, 'epochs': 1, 'batch_size': 8,
        'latent_size': 10, 'word_embed_size': 20, 'input_length': 37,
        'dropout': 0.0, 'model_type': model_type,
        'narre_num_reviews': 10, 'narre_num_words': 12,
        'total_users': 30, 'total_items': 20,
    }
    hp.update(kw)
    return hp


def make_batch(rng, hp, B, V, negs=False):
    mt = hp['model_type']
    U, I = hp['total_users'], hp['total_items']
    lead = (B, 6) if negs else (B,)
    T = hp['input_length']
    uid = rng.integers(0, U, size=lead)
    iid = rng.integers(0, I, size=lead)
    if B > 2 and not negs:                       # duplicate ids inside one batch
        uid[1] = uid[0]
        iid[2] = iid[0]
    if mt == 'NARRE':
        R, W = hp['narre_num_reviews'], hp['narre_num_words']
        ur = rng.integers(0, V, size=lead + (R, W))
        ir = rng.integers(0, V, size=lead + (R, W))
        ur[..., -1, :] = 0                       # a padded (all-zero) review
        ir[..., W // 2:] = np.where(rng.random(lead + (R, W - W // 2)) < 0.5, 0, ir[..., W // 2:])
    else:
        ur = rng.integers(0, V, size=lead + (T,))
        ir = rng.integers(0, V, size=lead + (T,))
        fill = rng.integers(1, T + 1, size=lead)
        pos = np.arange(T)
        ur = np.where(pos < fill[..., None], ur, 0)  # zero padded tails like data.py:198-199
        if not negs:
            ur[0] = 0                            # an all-padding document
            ir[-1] = ir[0]                       # duplicate rows
    this = rng.integers(0, V, size=lead + (T,))
    who = rng.integers(0, U + 2, size=lead + (10,))   # includes the +1 sentinel (data.py:275)
    rev = rng.integers(0, I + 2, size=lead + (10,))
    y = rng.integers(1, 6, size=lead).astype(np.float32)
    data = [this, who, rev, ur, ir, uid, iid]
    return [np.ascontiguousarray(d.astype(np.int64)) for d in data], y


def to_t(data, y):
    return [torch.from_numpy(d) for d in data], torch.from_numpy(y)


def build_neumf(hp, seed):
    """GMF / MLP: Model(hp) + xavier_init (main.py:299-301,308-310); NeuMF: NeuMF(hp).init(gmf, mlp)
    from freshly built GMF / MLP models, NO xavier_init (main.py:324-326).  Returns (model, extras)
    where extras holds the GMF / MLP weights NeuMF.init was fed with."""
    from pytorch_models.NeuMF import GMF, MLP, NeuMF
    from utils import xavier_init
    stage = hp['neumf_stage']
    torch.manual_seed(seed)
    if stage in ('GMF', 'MLP'):
        model = (GMF if stage == 'GMF' else MLP)(hp)
        xavier_init(model)
        return model, {}
    gmf, mlp = GMF(hp), MLP(hp)
    xavier_init(gmf)
    xavier_init(mlp)
    with torch.no_grad():                        # pre-training stand-in: biases away from their constants
        for m in (gmf, mlp):
            m.user_bias.add_(torch.randn_like(m.user_bias) * 0.05)
            m.item_bias.add_(torch.randn_like(m.item_bias) * 0.05)
    extras = {}
    for tag, m in (('init_gmf', gmf), ('init_mlp', mlp)):
        for k, v in m.state_dict().items():
            extras['%s/%s' % (tag, k)] = v.detach().numpy().copy()
    model = NeuMF(hp)
    model.init(gmf, mlp)
    return model, extras


def build(hp, V, seed):
    mt = hp['model_type']
    if mt == 'NeuMF':
        model, extras = build_neumf(hp, seed)
        build.extras = extras
        return model, hp
    build.extras = {}
    tmp = tempfile.mkdtemp(prefix='r4r_golden_')
    hp = dict(hp, data_dir=tmp + '/')
    rng = np.random.default_rng(seed)
    wv = rng.random((V, hp['word_embed_size'])).astype(np.float32).tolist()
    with open(tmp + '/word2vec.pkl', 'wb') as f:
        pickle.dump(wv, f, 2)
    if mt in ('deepconn', 'deepconn++'):
        from pytorch_models.DeepCoNN import DeepCoNN as Model
    elif mt in ('transnet', 'transnet++'):
        from pytorch_models.TransNet import TransNet as Model
    elif mt == 'NARRE':
        from pytorch_models.NARRE import NARRE as Model
    else:
        from pytorch_models.MF import MF as Model
    from utils import xavier_init
    torch.manual_seed(seed)
    model = Model(hp)
    xavier_init(model)                           # main.py:377
    return model, hp


def run_case(name, hp, V, B, seed, steps=3):
    from loss import MSELoss
    rng = np.random.default_rng(1000 + seed)
    model, hp = build(hp, V, seed)
    mt = hp['model_type']
    is_tn = mt in ('transnet', 'transnet++')
    out = {}
    for k, v in model.state_dict().items():
        out['w/' + k] = v.detach().numpy().copy()
    out.update(build.extras)
    batches = [make_batch(rng, hp, B, V), make_batch(rng, hp, max(1, B - 3), V)]  # ragged second batch
    crit = MSELoss(hp)

    model.eval()
    with torch.no_grad():
        for k, (d, y) in enumerate(batches):
            for s, arr in enumerate(d):
                out['b%d/%d' % (k, s)] = arr
            out['y%d' % k] = y
            o = model(to_t(d, y)[0])
            if is_tn:
                out['eval%d/src' % k] = o[0].numpy().copy()
                out['eval%d/tgt' % k] = o[1].numpy().copy()
                out['eval%d/transform' % k] = o[2].numpy().copy()
            else:
                out['eval%d' % k] = o.numpy().copy()
        nd, ny = make_batch(rng, hp, 3, V, negs=True)
        for s, arr in enumerate(nd):
            out['neg/%d' % s] = arr
        o = model(to_t(nd, ny)[0])
        out['neg_eval'] = (o[0] if is_tn else o).numpy().copy()

    if not is_tn:                                # TransNet's reference step crashes on torch>=1.5
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])
        for step in range(steps):
            d, y = to_t(*batches[step % 2])
            model.zero_grad()
            opt.zero_grad()
            o = model(d)
            se = crit(o, y, return_mean=False)
            out['se%d' % step] = se.detach().numpy().copy()
            torch.mean(se).backward()
            if step == 0:
                for k, p in model.named_parameters():
                    if p.grad is not None:
                        out['g0/' + k] = p.grad.numpy().copy()
            opt.step()
            if step in (0, steps - 1):
                for k, v in model.state_dict().items():
                    out['w%d/%s' % (step + 1, k)] = v.detach().numpy().copy()
        names = {id(p): k for k, p in model.named_parameters()}
        for p, st in opt.state.items():
            out['m%d/%s' % (steps, names[id(p)])] = st['exp_avg'].numpy().copy()
            out['v%d/%s' % (steps, names[id(p)])] = st['exp_avg_sq'].numpy().copy()

    keep = ('model_type', 'latent_size', 'word_embed_size', 'input_length', 'dropout', 'lr',
            'weight_decay', 'total_users', 'total_items', 'narre_num_reviews', 'narre_num_words',
            'batch_size') + (('neumf_stage',) if mt == 'NeuMF' else ())
    out['hp_keys'] = np.array(keep)
    out['hp_vals'] = np.array([str(hp[k]) for k in keep])
    out['vocab'] = np.array(V)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-22s %7.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(out)))


class DataAdam:
    """Adam with torch.optim.Adam's arithmetic (betas .9/.999, eps 1e-8, L2 decay on the grad,
    bias-corrected) that writes through ``p.data`` -- what every optimiser did on torch 0.4, the
    version the reference targets (README.md:20).  Writing through .data does not bump autograd's
    version counter, so the reference's three backward passes over one retained graph
    (main.py:38-50) run instead of raising.  Generator-side helper, not
    reference code; its arithmetic is the one the other fixtures pin against torch.optim.Adam."""

    def __init__(self, params, lr, weight_decay):
        self.params = list(params)
        self.lr, self.wd = lr, weight_decay
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        import math
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.setdefault(id(p), {'t': 0, 'm': torch.zeros_like(p.data), 'v': torch.zeros_like(p.data)})
            st['t'] += 1
            g = p.grad.data + self.wd * p.data
            st['m'].mul_(0.9).add_(g, alpha=0.1)
            st['v'].mul_(0.999).addcmul_(g, g, value=1 - 0.999)
            bc1, bc2 = 1 - 0.9 ** st['t'], 1 - 0.999 ** st['t']
            p.data.addcdiv_(st['m'], st['v'].sqrt() / math.sqrt(bc2) + 1e-8, value=-self.lr / bc1)


def run_transnet_training(name, hp, V, B, seed, steps=3):
    """The reference's TransNet step (main.py:26-53) with write-through optimisers grouped like
    utils.init_transnet_optim (utils.py:70-92); appends the trajectory to <name>.npz."""
    from loss import MSELoss
    rng = np.random.default_rng(1000 + seed)
    model, hp = build(hp, V, seed)
    batches = [make_batch(rng, hp, B, V), make_batch(rng, hp, max(1, B - 3), V)]
    path = os.path.join(OUT, name + '.npz')
    out = dict(np.load(path))
    crit = MSELoss(hp)
    kw = dict(lr=hp['lr'], weight_decay=hp['weight_decay'])
    fm_params = list(model.source_fm.parameters())
    if hp['model_type'] == 'transnet++':
        fm_params += [model.user_embedding.weight, model.item_embedding.weight]
    optimizer = [DataAdam(model.source.parameters(), **kw), DataAdam(fm_params, **kw),
                 DataAdam(model.target.parameters(), **kw), DataAdam(model.parameters(), **kw)]
    model.train()
    for step in range(steps):
        d, y = to_t(*batches[step % 2])
        model.zero_grad()
        for o in optimizer:
            o.zero_grad()
        all_output = model(d)
        optimizer_source, optimizer_source_fm, optimizer_target, optimizer_all = optimizer
        loss_target = crit(all_output[1], y)
        loss_target.backward(retain_graph=True)
        optimizer_target.step()
        loss_transform = all_output[2]
        loss_transform.backward(retain_graph=True)
        optimizer_source.step()
        loss_source = crit(all_output[0], y, return_mean=False)
        out['tn_se%d' % step] = loss_source.detach().numpy().copy()
        out['tn_aux%d' % step] = np.array([float(loss_target.detach()), float(loss_transform.detach())], np.float32)
        torch.mean(loss_source).backward()
        optimizer_source_fm.step()
        if step in (0, steps - 1):
            for k, v in model.state_dict().items():
                out['tn_w%d/%s' % (step + 1, k)] = v.detach().numpy().copy()
    # the training batches are the same b0 / b1 stored by run_case (same rng stream): assert it
    for k, (dd, yy) in enumerate(batches):
        for s_, arr in enumerate(dd):
            assert np.array_equal(out['b%d/%d' % (k, s_)], arr)
    np.savez_compressed(path, **out)
    print('%-22s + TransNet training trajectory, %7.1f KB' % (name, os.path.getsize(path) / 1024))


def common_paths():
    """get_common_path strings for the shipped defaults (hyper_params.py:3-48), as data."""
    import json
    from hyper_params import get_common_path, hyper_params as ref_hp     # creates saved_* in the scratch cwd
    out = {}
    for mt in ('bias_only', 'MF', 'MF_dot', 'deepconn', 'deepconn++', 'NARRE', 'transnet', 'transnet++'):
        hp = dict(ref_hp, model_type=mt, only_reviews=False)          # the key the reference forgets (fact 10)
        out[mt] = get_common_path(hp)
    json.dump(out, open(os.path.join(OUT, 'common_paths.json'), 'w'), indent=1)
    print('common_paths.json', len(out), 'entries')


def main():
    os.chdir(tempfile.mkdtemp(prefix='r4r_cwd_'))
    if len(sys.argv) > 1 and sys.argv[1] == 'neumf':       # only the NeuMF family (added later)
        run_case('neumf_gmf', base_hp('NeuMF', latent_size=8, neumf_stage='GMF'), V=4, B=13, seed=10)
        run_case('neumf_mlp', base_hp('NeuMF', latent_size=8, neumf_stage='MLP'), V=4, B=13, seed=11)
        run_case('neumf_full', base_hp('NeuMF', latent_size=8, neumf_stage='NeuMF'), V=4, B=13, seed=12)
        return
    common_paths()
    run_case('mf_bias_only', base_hp('bias_only'), V=4, B=13, seed=1)
    run_case('mf_dot', base_hp('MF_dot', latent_size=8), V=4, B=13, seed=2)
    run_case('mf_full', base_hp('MF', latent_size=6), V=4, B=13, seed=3)
    run_case('deepconn_e20', base_hp('deepconn'), V=120, B=5, seed=4)
    run_case('deepconn_e64', base_hp('deepconn', word_embed_size=64, input_length=50, latent_size=7),
             V=90, B=4, seed=5)
    run_case('deepconnpp_e20', base_hp('deepconn++'), V=120, B=5, seed=6)
    run_case('narre_e16', base_hp('NARRE', word_embed_size=16), V=80, B=4, seed=7)
    run_case('transnet_e16', base_hp('transnet', word_embed_size=16, input_length=21), V=80, B=4, seed=8)
    run_case('transnetpp_e16', base_hp('transnet++', word_embed_size=16, input_length=21), V=80, B=4, seed=9)
    run_transnet_training('transnet_e16', base_hp('transnet', word_embed_size=16, input_length=21), V=80, B=4, seed=8)
    run_transnet_training('transnetpp_e16', base_hp('transnet++', word_embed_size=16, input_length=21), V=80, B=4,
                          seed=9)
    run_case('neumf_gmf', base_hp('NeuMF', latent_size=8, neumf_stage='GMF'), V=4, B=13, seed=10)
    run_case('neumf_mlp', base_hp('NeuMF', latent_size=8, neumf_stage='MLP'), V=4, B=13, seed=11)
    run_case('neumf_full', base_hp('NeuMF', latent_size=8, neumf_stage='NeuMF'), V=4, B=13, seed=12)


if __name__ == '__main__':
    main()
