"""Fused dense Adam on the HIP path, with torch.optim.Adam's calling surface
(``Adam(params, lr=, weight_decay=)``, ``.zero_grad()``, ``.step()``,
``.state_dict()``), so the host loop text is unchanged (main.py:94-96,26-29,60).

Semantics restated from torch.optim.Adam as the reference uses it: betas
(0.9, 0.999), eps 1e-8, L2 weight decay added to the gradient, bias-corrected,
dense -- parameters whose ``.grad`` is None are skipped and their step counter
does not advance.
"""
import ctypes

import torch

from . import _lib


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.state = {}                       # id(param) -> dict(step, exp_avg, exp_avg_sq)
        self.step_dev = None                  # device int64 [1]: completed steps (hipGraph mode, graph.py)
        self.param_groups = [dict(params=self.params, lr=self.lr, betas=self.betas, eps=self.eps,
                                  weight_decay=self.weight_decay)]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is None:
                continue
            if set_to_none:
                p.grad = None
            else:
                p.grad.detach_()
                p.grad.zero_()

    def _state(self, p):
        st = self.state.get(id(p))
        if st is None:
            st = dict(step=0, exp_avg=torch.zeros_like(p, memory_format=torch.contiguous_format),
                      exp_avg_sq=torch.zeros_like(p, memory_format=torch.contiguous_format))
            self.state[id(p)] = st
        return st

    @torch.no_grad()
    def step(self):
        lr = float(self.param_groups[0]['lr'])
        by_step = {}
        for p in self.params:
            if p.grad is None:
                continue
            if not p.is_cuda:
                raise RuntimeError('reviews4rec_amd.optim.Adam: parameters must live on a ROCm device; '
                                   'the HIP path has no CPU fallback')
            if not (p.is_contiguous() and p.dtype == torch.float32):
                raise RuntimeError('reviews4rec_amd.optim.Adam: fp32 contiguous parameters only')
            st = self._state(p)
            st['step'] += 1
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            by_step.setdefault(st['step'], []).append((p, g, st))
        lib = _lib.lib()
        stream = _lib.current_stream()
        for step, items in by_step.items():
            n = len(items)
            arr = ctypes.c_uint64 * n
            P = arr(*[p.data_ptr() for p, _, _ in items])
            G = arr(*[g.data_ptr() for _, g, _ in items])
            M = arr(*[st['exp_avg'].data_ptr() for _, _, st in items])
            V = arr(*[st['exp_avg_sq'].data_ptr() for _, _, st in items])
            numel = (ctypes.c_int64 * n)(*[p.numel() for p, _, _ in items])
            rc = lib.r4r_adam_multi(n, P, G, M, V, numel, lr, self.betas[0], self.betas[1], self.eps,
                                    self.weight_decay, step, _lib.ptr(self.step_dev), stream)
            _lib.check(rc, 'r4r_adam_multi')
        if self.step_dev is not None:
            if len(by_step) > 1:
                raise RuntimeError('Adam(graph mode): every parameter must share one step count')
            _lib.check(lib.r4r_counter_add(_lib.ptr(self.step_dev), 1, stream), 'r4r_counter_add')

    def enable_device_step(self):
        """Keep the step count in device memory (needed before capturing step() in a hipGraph).
        Every parameter must already have optimizer state (run one eager step first) so the
        counter can be initialised from the common step count."""
        steps = {st['step'] for st in self.state.values()}
        if len(steps) > 1:
            raise RuntimeError('Adam(graph mode): parameters are at different step counts %r' % (steps,))
        dev = self.params[0].device
        self.step_dev = torch.tensor([steps.pop() if steps else 0], dtype=torch.int64, device=dev)

    def state_dict(self):
        if self.step_dev is not None:                     # graph mode: the device counter is the truth
            done = int(self.step_dev.item())
            for st in self.state.values():
                st['step'] = done
        return {'state': {i: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state[id(p)].items()}
                          for i, p in enumerate(self.params) if id(p) in self.state},
                'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}

    def load_state_dict(self, sd):
        for i, st in sd['state'].items():
            p = self.params[int(i)]
            self.state[id(p)] = {k: (v.to(p.device).clone() if torch.is_tensor(v) else v) for k, v in st.items()}
        for k, v in sd['param_groups'][0].items():
            self.param_groups[0][k] = v
        self.lr = float(self.param_groups[0]['lr'])
        if self.step_dev is not None:
            steps = {st['step'] for st in self.state.values()}
            self.step_dev.fill_(steps.pop() if len(steps) == 1 else 0)
