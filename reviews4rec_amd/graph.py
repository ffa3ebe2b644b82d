"""hipGraph capture of a whole module-engine training step.

The op-by-op autograd path issues ~45 (DeepCoNN) to ~130 (NARRE) tiny launches per step
through Python; at batch 128 that is 2-3x more host time than kernel time.  ``GraphedStep``
captures ONE full step -- zero_grad, forward, per-example SE, mean, backward, fused Adam --
into a hipGraph (``torch.cuda.CUDAGraph``; every C-ABI call goes to the capture stream) and
replays it per batch: the MI355X answer to a launch-bound loop, instead of a tracing compiler.

What makes the step capturable:
  * batches are copied into static device buffers (shapes fixed at capture; a ragged last
    batch falls back to the eager path),
  * the two per-step scalars that are otherwise kernel ARGUMENTS live in device memory and are
    advanced by the captured kernels themselves: the dropout Philox offset
    (``ops.DropoutState.device_counter``) and Adam's step count (``optim.Adam.step_dev``),
  * nothing in the C ABI allocates or synchronises.

Host-loop semantics are those of main.train (main.py:26-60): the running sum of SE is kept in
``self.sse`` on the device.
"""
import torch

from . import ops


class GraphedStep:
    def __init__(self, model, criterion, optimizer, example_data, example_y, warmup=2):
        """optimizer: one fused Adam, or TransNet's list [source, source_fm, target, all]
        (utils.init_transnet_optim) -- then the captured body is the 3-optimiser step of
        main.py:35-53."""
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.tn = isinstance(optimizer, (list, tuple))
        self.optimizers = list(optimizer[:3]) if self.tn else [optimizer]
        dev = example_y.device
        self.static_data = [None if d is None else d.clone() for d in example_data]
        self.static_y = example_y.clone()
        self.sse = torch.zeros((), dtype=torch.float32, device=dev)
        self.shapes = [None if d is None else tuple(d.shape) for d in example_data]
        if ops.DropoutState.device_counter is None:
            ops.DropoutState.device_counter = torch.tensor([ops.DropoutState.offset], dtype=torch.int64, device=dev)
        # Warm-up on a side stream (lazy inits: kernel attributes, Adam state tensors, allocator
        # pools).  The warm-up steps must not count as training: parameters, optimiser state and
        # the dropout stream position are snapshotted and restored around them.
        params = [p for p in model.parameters()]
        snap_p = [p.detach().clone() for p in params]
        snap_s = [{k: {n: (v.clone() if torch.is_tensor(v) else v) for n, v in st.items()}
                   for k, st in o.state.items()} for o in self.optimizers]
        snap_off = (ops.DropoutState.offset, ops.DropoutState.device_counter.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager(self.static_data, self.static_y, count=False)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for p, sp in zip(params, snap_p):
                p.copy_(sp)
            for o, snap in zip(self.optimizers, snap_s):
                for k, st in o.state.items():
                    old = snap.get(k)
                    for n in ('exp_avg', 'exp_avg_sq'):
                        st[n].copy_(old[n]) if old is not None else st[n].zero_()
                    st['step'] = old['step'] if old is not None else 0
        ops.DropoutState.offset = snap_off[0]
        ops.DropoutState.device_counter.copy_(snap_off[1])
        for o in self.optimizers:
            o.enable_device_step()
        self.graph = torch.cuda.CUDAGraph()
        self._zero_grads(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self._body(self.static_data, self.static_y, count=True)
        self.graph_se = self.last_se                         # static output of the captured step
        self.warmup_steps = warmup

    def _zero_grads(self, set_to_none=True):
        self.model.zero_grad(set_to_none=set_to_none)
        for o in (self.optimizer if self.tn else [self.optimizer]):
            o.zero_grad(set_to_none=set_to_none)

    def _body(self, data, y, count):
        out = self.model(data)
        if self.tn:
            opt_source, opt_source_fm, opt_target = self.optimizers
            self.criterion(out[1], y).backward(retain_graph=True)
            opt_target.step()
            out[2].backward(retain_graph=True)
            opt_source.step()
            se = self.criterion(out[0], y, return_mean=False)
            if count:
                self.sse.add_(se.detach().sum())
            torch.mean(se).backward()
            opt_source_fm.step()
        else:
            se = self.criterion(out, y, return_mean=False)
            if count:
                self.sse.add_(se.detach().sum())
            torch.mean(se).backward()
            self.optimizer.step()
        self.last_se = se

    def _eager(self, data, y, count=True):
        self._zero_grads()
        self._body(data, y, count)

    def __call__(self, data, y):
        """One training step on (data, y).  Same-shape batches replay the graph."""
        if [None if d is None else tuple(d.shape) for d in data] != self.shapes:
            self._eager(data, y)                             # ragged tail batch
            return self.last_se
        for dst, src in zip(self.static_data, data):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        self.static_y.copy_(y, non_blocking=True)
        self.graph.replay()
        return self.graph_se
