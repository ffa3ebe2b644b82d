"""Adam and the single-optimiser training step, restated (CPU oracle).

TEST INFRASTRUCTURE -- never imported by the product package.

Reference map (file:line under /root/reference):
  adam_step   main.py:94-96 -> torch.optim.Adam(lr, weight_decay) with the
              library defaults betas=(0.9, 0.999), eps=1e-8, amsgrad=False,
              L2 weight decay added to the gradient, bias-corrected
  train_step  main.py:26-32,56-60 (zero_grad -> forward -> per-example SE ->
              sum for the metric -> mean -> backward -> step)

Adam skips parameters whose ``.grad`` is None (e.g. DeepCoNN's unused ``final``
MLP and bias vectors in 'deepconn' mode -- SURVEY.md fact 7); their state is
never created and their step counter never advances.
"""
import math

import torch

from .models import model_forward, mse_loss, trainable_names


class AdamState:
    def __init__(self):
        self.m = {}
        self.v = {}
        self.t = {}


def adam_step(params, grads, state, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
    """In-place Adam update of ``params[name]`` for every name in ``grads``
    whose gradient is not None."""
    b1, b2 = betas
    for name, g in grads.items():
        if g is None:
            continue
        p = params[name]
        if name not in state.t:
            state.m[name] = torch.zeros_like(p)
            state.v[name] = torch.zeros_like(p)
            state.t[name] = 0
        state.t[name] += 1
        t = state.t[name]
        if weight_decay != 0:
            g = g + weight_decay * p
        m, v = state.m[name], state.v[name]
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = v.sqrt() / math.sqrt(bc2) + eps
        p.addcdiv_(m, denom, value=-(lr / bc1))


def train_step(params, data, y, hyper_params, state, masks=None):
    """One reference training step (non-TransNet models).  Mutates ``params``
    and ``state``.  Returns (sum of per-example SE, grads dict)."""
    names = trainable_names(params)
    leaves = {k: params[k].detach().clone().requires_grad_(True) for k in names}
    full = dict(params)
    full.update(leaves)
    out = model_forward(full, data, hyper_params, train=True, masks=masks)
    se = mse_loss(out, y, return_mean=False)
    loss = se.mean()
    loss.backward()
    grads = {k: leaves[k].grad for k in names}
    adam_step(params, grads, state, hyper_params['lr'], hyper_params['weight_decay'])
    return float(se.detach().sum()), grads


def transnet_train_step(params, data, y, hyper_params, states, masks=None):
    """TransNet's three-optimiser step (main.py:26-53 with utils.init_transnet_optim,
    utils.py:70-92) with the torch-0.4 semantics the reference was written for: ONE forward,
    three backward passes over the retained graph, and an optimiser step between them that
    writes through ``.data`` (no autograd version bump), so later backward passes see
    post-step weights wherever a weight is a saved tensor and pre-step activations everywhere.
    Gradients accumulate across the three passes (zeroed once per batch).

    ``states`` = dict(source=AdamState(), source_fm=AdamState(), target=AdamState()).
    Mutates ``params`` and ``states``; returns (per-example source SE, loss_target, loss_transform).
    ``masks``: dropout multipliers by site name (tests inject the device-drawn ones).
    """
    mt = hyper_params['model_type']
    names = trainable_names(params)
    leaves = {k: params[k].detach().clone().requires_grad_(True) for k in names}
    full = dict(params)
    full.update(leaves)
    src_pred, tgt_pred, transform = model_forward(full, data, hyper_params, train=True, masks=masks)

    groups = {
        'target': [k for k in names if k.startswith('target.')],
        'source': [k for k in names if k.startswith('source.')],
        'source_fm': [k for k in names if k.startswith('source_fm.')] +
                     (['user_embedding.weight', 'item_embedding.weight'] if mt == 'transnet++' else []),
    }

    def step(group):
        data_view = {k: leaves[k].data for k in groups[group]}          # write-through, no version bump
        grads = {k: (None if leaves[k].grad is None else leaves[k].grad.detach().clone()) for k in groups[group]}
        adam_step(data_view, grads, states[group], hyper_params['lr'], hyper_params['weight_decay'])

    loss_target = mse_loss(tgt_pred, y)
    loss_target.backward(retain_graph=True)
    step('target')
    transform.backward(retain_graph=True)
    step('source')
    se = mse_loss(src_pred, y, return_mean=False)
    se.mean().backward()
    step('source_fm')
    for k in names:
        params[k] = leaves[k].detach()
    return se.detach(), float(loss_target.detach()), float(transform.detach())
