"""Native fused training engine for DeepCoNN ('deepconn' mode).

``DeepCoNNEngine(model, ...)`` takes the drop-in ``DeepCoNN`` module, moves its
trainable parameters into ONE flat device buffer (the module's Parameters become
views of it, so ``model(data)``, ``state_dict()`` and checkpoints keep working) and
runs a whole training step -- forward, loss, backward, gradient all-reduce, Adam --
as two C calls: ``r4r_deepconn_step`` (6 kernels) and ``r4r_adam_multi`` (1 kernel).

Host-loop contract restated from main.py:26-60: zero_grad -> forward -> per-example SE
(summed into the running train metric) -> mean -> backward -> optimizer.step().  The
running metric lives on the device (``sse``) and is read once per epoch, which gives
the same number as the reference's per-batch ``float(torch.sum(loss))`` without a
device->host sync per step.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ptr


GEMM_MATH_REFRESH = 32        # steps between two reads of max |conv weights| (one small D2H sync each)
_MATH_OWNER = [None]          # the engine whose table / weights the process-wide scales currently describe


def apply_gemm_math(table, conv_weights=()):
    """R4R_GEMM_MATH=f16x2 switches the projection GEMM to fp16-split operands with fp32 accumulation (an
    opt-in experiment, csrc/project_f16.hip; never the default).  Its exact power-of-two scales come from
    here: the frozen table's maximum and the conv weights' (re-read every GEMM_MATH_REFRESH steps by the
    engines; the device-side scale keeps two binades of headroom, Adam moves a weight by <= lr per step)."""
    mode = os.environ.get('R4R_GEMM_MATH', 'f32')
    if mode not in ('f32', 'f16x2'):
        raise ValueError("R4R_GEMM_MATH must be 'f32' or 'f16x2', got %r" % (mode,))
    lib = _lib.lib()
    if mode == 'f16x2':
        wmax = max([float(w.detach().abs().max().item()) for w in conv_weights] + [1e-30])
        _lib.check(lib.r4r_gemm_math(1, float(table.detach().abs().max().item()), wmax), 'r4r_gemm_math')
    else:
        _lib.check(lib.r4r_gemm_math(0, 0.0, 0.0), 'r4r_gemm_math')
    return mode


def step_or_nothing(fn):
    """train_step decorator: a step that RAISES (a shard larger than batch_size, a batch-cap check of the C entry point,
    no memory for the workspace -- all raised before anything is launched) must not consume its step number.  The
    temporally blocked sweeps count pending gradient-zero Adam updates from step_count, so a number consumed by a step
    that never ran became one weight-decay / moment-decay update of the ID tables that dense Adam never made."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kw):
        before = self.step_count
        self._step_launched = False                          # set by the engine once the step's first launch is enqueued
        try:
            return fn(self, *args, **kw)
        except BaseException:
            # (a step that raises AFTER it enqueued work -- a later buffer of the data-parallel exchange that does not
            # fit, a C argument check of a later call -- has consumed its number: the dense parameters may already
            # carry update N, and a retry under the same number would apply it twice)
            if not self._step_launched:
                self.step_count = before
            raise
    return wrapped


class _RawPtr:
    """A device address standing in for a tensor where only ``ptr(t)`` is taken: a batch inside a span loader's ring."""
    __slots__ = ('p', 'keep')

    def __init__(self, p, keep=None):
        self.p, self.keep = int(p), keep

    def data_ptr(self):
        return self.p


def _span_slots(desc, batch):
    """The eight device addresses of batch `batch` of a span loader (this, who, what, user doc, item doc, uid, iid, y)."""
    slots = (ctypes.c_uint64 * 8)()
    _lib.check(_lib.lib().r4r_span_batch(desc.words, int(batch), slots), 'r4r_span_batch')
    return slots


class _Spans:
    """K training steps per host call (include/r4r.h, "Spans"; csrc/span.hip): main.py:23-60's loop body enqueued from C
    for the FULL batches of a device-side loader's epoch -- the batch construction for a group of batches per launch,
    the family's native step per batch.  The kernels and their arguments are the per-step path's, so an epoch run
    through spans leaves the same bits as the same epoch run through train_step (tests/test_gpu_span.py); what the
    per-step path pays per batch on the host (the iterator, six tensor slices, two ctypes calls of 25 and 35
    arguments: 70-100 us, more than NARRE's or MF_dot's whole step takes on the device) is paid once per SPAN_STEPS
    batches.  Steps a span cannot carry -- a ragged last batch, a step whose conv-rule probe reads a counter back --
    go through train_step in their place."""
    SPAN_STEPS = 64              # steps per host call (a call cannot be interrupted; 64 steps are 1.5-7 ms)

    def _span_ok(self, desc):
        return False

    def _span_limit(self, desc, ahead=0):
        """How many consecutive steps, starting `ahead` steps from now, one call may carry (0: that step is its own)."""
        return self.SPAN_STEPS

    def _span_dp_ok(self):
        """Can this engine's DATA-PARALLEL step be enqueued from C (the exchange included)?"""
        return False

    def train_epoch(self, reader, counts=None):
        """One epoch over ``reader`` (data.DataLoader on the device): every batch of ``reader.iter()``, in order, as
        main.train's loop would train on it.  -> (ratings, batches), or None when this reader / engine pair has no
        span form (the caller iterates).  Data parallel: ``counts`` = the global size of every step's batch
        (DataParallel.epoch_counts); steps in which every rank holds a full batch go through the span entry, the
        others -- and engines whose exchange is a Python-side call -- through train_step."""
        dp = getattr(self, 'dp', None)
        dp_on = dp is not None and dp.on
        if os.environ.get('R4R_SPANS', '1') == '0' or (dp_on and (counts is None or not self._span_dp_ok())):
            return None
        desc = reader.span_descriptor() if hasattr(reader, 'span_descriptor') else None
        if desc is None or not self._span_ok(desc) or (desc.full_batches == 0 and not dp_on):
            return None
        kw = {'defer_sweep': True} if getattr(self, 'TEMPORAL_SWEEP', False) else {}
        nb, total, b = desc.full_batches, len(reader), 0
        if dp_on:
            # (every rank walks the SAME list of steps -- len(counts) of them -- and issues one exchange per step,
            # whether from a span or from train_step: a shard that has run out contributes an empty batch)
            total = len(counts)
            full = desc.batch_size * dp.world
            nb = min(nb, next((k for k, c in enumerate(counts) if c != full), len(counts)))
        while b < total:
            k = min(self._span_limit(desc), self.SPAN_STEPS, nb - b) if b < nb else 0
            if k <= 0:                                       # the ragged tail, a probing step, a step some rank has less for
                data, y = reader.batch(b)
                self.train_step(data, y, n_global=int(counts[b]) if dp_on else int(y.shape[0]), **kw)
                b += 1
                continue
            announce = b + k < nb and self._span_limit(desc, ahead=k) > 0
            self._span(desc, b, k, announce)
            b += k
        return float(desc.n_ratings), float(total)


def load_named_moments(who, m, v, sd):
    """Copy a checkpoint's per-parameter Adam moments into the engine's views -- after checking every one of them (a
    missing name or a size mismatch raises before anything was written: a rejected checkpoint leaves the engine as it
    was).  -> the checkpoint's scalar state (step, dropout_offset, lr, weight_decay, betas, eps)."""
    scalars = (int(sd['step']), int(sd['dropout_offset']), float(sd['lr']), float(sd['weight_decay']),
               tuple(sd['betas']), float(sd['eps']))
    for k in m:
        for mine, theirs in ((m[k], sd['exp_avg'][k]), (v[k], sd['exp_avg_sq'][k])):
            if theirs.numel() != mine.numel():
                raise ValueError('%s.load_state_dict: %s has %d moment elements, the model %d'
                                 % (who, k, theirs.numel(), mine.numel()))
    for k in m:
        m[k].copy_(sd['exp_avg'][k].to(m[k].device).view_as(m[k]))
        v[k].copy_(sd['exp_avg_sq'][k].to(v[k].device).view_as(v[k]))
    return scalars


def gather_entry_fields(owner, fields, n, B_pad, world, all_gather):
    """Data parallel: a rank's compact entries -> the gathered arrays the `r4r_*_rows_apply` launches read, with ONE
    packing launch, ONE all_gather and one unpacking launch (csrc/dp_pack.hip).  fields: [(tensor of this rank's n
    entries -- contiguous, int64 or float32, [n] or [n, w]; ignored when n == 0 --, w, torch dtype)]; ids (int64) are
    padded with -1, values with 0.  all_gather(out, block) (None: one rank, nothing to gather).  -> the gathered tensors
    [world * B_pad] / [world * B_pad, w], buffers owned by `owner` and reused from step to step."""
    lib = _lib.lib()
    nf = len(fields)
    units = [w * (2 if dt == torch.int64 else 1) for _, w, dt in fields]
    key = ('entry_fields', B_pad, world) + tuple((w, dt) for _, w, dt in fields)
    cache = owner.__dict__.setdefault('_entry_field_bufs', {})
    if key not in cache:
        widths = (ctypes.c_int * nf)(*units)
        nbytes = lib.r4r_dp_block_bytes(nf, widths, B_pad)
        dev = owner.dev
        block = torch.zeros(max(nbytes, 4), dtype=torch.uint8, device=dev)
        blocks = block if all_gather is None else torch.zeros(world * max(nbytes, 4), dtype=torch.uint8, device=dev)
        outs = [torch.empty((world * B_pad,) + ((w,) if w > 1 else ()), dtype=dt, device=dev) for _, w, dt in fields]
        cache[key] = (widths, (ctypes.c_int * nf)(*[1 if dt == torch.int64 else 0 for _, _, dt in fields]), block, blocks, outs,
                      (ctypes.c_uint64 * nf)(*[o.data_ptr() for o in outs]))
    widths, ones, block, blocks, outs, dst = cache[key]
    src = (ctypes.c_uint64 * nf)(*[(t.data_ptr() if n > 0 else 0) for t, _, _ in fields])
    if n > 0:
        for t, w, dt in fields:
            if not (t.is_contiguous() and t.dtype == dt and t.numel() == n * w):
                raise RuntimeError('gather_entry_fields: a field is not a contiguous [n, %d] %s tensor' % (w, dt))
    _lib.check(lib.r4r_dp_pack(nf, src, widths, ones, n, B_pad, ptr(block), _lib.current_stream()), 'r4r_dp_pack')
    if all_gather is not None:
        all_gather(blocks, block)
    _lib.check(lib.r4r_dp_unpack(nf, dst, widths, ptr(blocks), world, B_pad, _lib.current_stream()), 'r4r_dp_unpack')
    return outs


def pad4(E):
    return (int(E) + 3) // 4 * 4


def pad_width(E):
    """Row width the native engines give the frozen table and the conv-weight slots.  Any width is rounded up to whole
    float4 (the kernels read 16-byte pieces); tables wide enough for project-then-gather (E >= 128) are rounded up
    to a whole K chunk of 16 floats, which makes every row a whole number of 64-byte pieces: the projection GEMM
    fetches table rows and weight rows in 64-byte pieces, and at the reference's E = 300 (1,200-byte rows) three of
    four pieces straddled two 64-byte requests (profiles/r03f: 28 L2 requests per 16-piece load instruction)."""
    E = int(E)
    return (E + 15) // 16 * 16 if E >= 128 else pad4(E)


def padded_word_table(table):
    """The HIP kernels read table rows and conv-weight windows as float4: rows must be 16-byte aligned (64-byte
    aligned from E = 128 on: pad_width).  The
    reference accepts any ``word_embed_size`` (hyper_params.py:64, common_pytorch_models.py:15; GloVe-50 is a
    common choice), so a table whose width is not a multiple of 4 (16) gets a zero-padded device copy -- it is frozen
    (Embedding.from_pretrained, DeepCoNN.py:15), one copy at engine construction is all it takes -- and the conv
    weights live in the engines' flat buffers with the same padded width, their Parameters being the [..., :E]
    views.  Exact: every added term of the convolution is 0 * 0; the pad columns of the weights get a zero
    gradient (g * 0) and a zero weight-decay term, so Adam leaves them at exactly 0."""
    V, E = table.shape
    E4 = pad_width(E)
    if E4 == E:
        return table
    out = torch.zeros((V, E4), dtype=table.dtype, device=table.device)
    out[:, :E].copy_(table.detach())
    return out


class _SweepSchedule:
    """Host side of the temporally blocked ID-table sweeps (include/r4r.h, r4r_mf_step): every table element is
    current through step `_tb_base`, and since then every training step has visited the chunks on the
    period-`_tb_period` schedule (1: every chunk, every step).  A step that leaves the schedule in force -- another
    period, or no deferral -- visits every chunk under the OLD one and starts the new one."""
    _tb_base, _tb_period, _tb_used = 0, 1, False

    @staticmethod
    def configured_period(hp):
        """Visit period of the blocked sweeps: hyper_params['sweep_period'] (default 8), overridden by R4R_SWEEP_PERIOD,
        capped by R4R_SWEEP_PERIOD_MAX (the C side applies at most 8 pending updates per visit); 1 = the plain dense sweep."""
        cap = int(os.environ.get('R4R_SWEEP_PERIOD_MAX', 8))
        return max(1, min(cap, int(os.environ.get('R4R_SWEEP_PERIOD', hp.get('sweep_period', 8)))))

    def _schedule(self, defer):
        """(period, base, all_chunks, period afterwards) of the next training step."""
        want = self.sweep_period if (defer and self.has_tables) else 1
        period = self._tb_period
        return period, self._tb_base, int(want != period or period == 1), want

    def _scheduled(self, sweep_all, want, step):
        if sweep_all:
            self._tb_base, self._tb_period = int(step), want
        if want > 1:
            self._tb_used = True

    def check_schedule(self):
        """Read the sweeps' device-side flag (one int): raises if more updates were ever pending than a visit applies.
        (`check_announcements` is the same call under its round-3 name.)"""
        return self.check_announcements()

    def _pending(self, last_step=None):
        """the last completed step, if the tables may be behind it (else None)"""
        step = int(self.step_count if last_step is None else last_step)
        return step if (self.has_tables and self._ws is not None and self._tb_base < step) else None


def flush_before_state_dict(engine, model):
    """The temporally blocked sweeps leave table chunks no recent batch named up to `sweep_period - 1` steps behind
    until engine.flush(); the reference's Parameters are always current (main.py:125 reads state_dict() whenever it
    likes).  A state_dict pre-hook on the model and every submodule brings the pending updates in first, so
    ``model.state_dict()`` / ``torch.save(model.state_dict())`` are the model at any point of an epoch.  (A direct read
    of ``model.user_embedding.weight`` between flushes cannot be intercepted: call ``engine.flush()`` first.)"""
    hook = _FlushHook(engine)
    return [m.register_state_dict_pre_hook(hook) for m in model.modules()]


class _FlushHook:
    """The pre-hook itself: holds the engine weakly (the model must not keep its engine alive) and pickles as an inert
    hook (``torch.save(model)`` / ``copy.deepcopy(model)`` of a hooked module must keep working: a copy has no engine)."""

    def __init__(self, engine):
        import weakref
        self._ref = weakref.ref(engine)

    def __call__(self, module, prefix, keep_vars):
        e = self._ref()
        if e is not None:
            e.flush()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._ref = lambda: None

    def __deepcopy__(self, memo):
        h = _FlushHook.__new__(_FlushHook)
        h._ref = lambda: None
        return h


def slot_view(flat, o, s, shape, E_model, E):
    """View of flat[o : o + s] with a parameter's shape; conv weights ([F, 1, 3, E_model]) of a padded engine are
    the leading E_model columns of their [F, 1, 3, E] slot."""
    shape = tuple(shape)
    if E != E_model and len(shape) == 4 and shape[-1] == E_model:
        n = 1
        for d in shape[:-1]:
            n *= d
        if n * E == s:
            return flat[o:o + s].view(*shape[:-1], E)[..., :E_model]
    return flat[o:o + s].view(shape)


class _ConvRule:
    """The engines' automatic choice between the two convolution algorithms (conv_algo = 0).

    project-then-gather's work follows the batch's DISTINCT tokens, the direct conv's its positions, so
    which is faster depends on the data (at cfg5's million-word vocabulary the direct
    conv wins by 10-45 % on full-length or uniformly drawn documents, projection wins on Amazon-shaped
    ones).  The distinct count of a batch is on the device anyway (the token compaction's counter; the
    gather kernel leaves it next to the live counter), so the rule MEASURES: shapes the static rule sends
    to the direct conv (narrow windows, small launches) stay there; everything else starts on projection,
    and at training step PROBE_AT and every PROBE_EVERY steps after it the step runs projection, its
    counters are read back (one stream sync per probe) and r4r_conv_pick's cost model decides what the
    following steps run.  Deterministic: the probe points are fixed step numbers and the decision is a
    pure function of the probed batch.  Both algorithms compute the same function (parity-tested against
    each other and the reference), so a switch only changes fp32 summation order."""
    PROBE_AT, PROBE_EVERY = 4, 1024
    SMALL_LAUNCH_MAX_V = 262144       # vocabulary up to which small training launches try the projection

    def _rule_reset(self):
        self._rule_n = 0                 # training steps the rule has seen
        self._rule_choice = None         # None until the first probe
        self.conv_rows = None            # distinct rows (all towers) of the last probe

    def _rule_state(self):
        """What a checkpoint must carry for a resumed run to pick the algorithm its uninterrupted twin runs (the
        two sum in different orders: without it a resume is right to rounding, not to the bit)."""
        return {'n': self._rule_n, 'choice': self._rule_choice}

    def _rule_load(self, st):
        if st:
            self._rule_n, self._rule_choice = int(st['n']), (None if st['choice'] is None else int(st['choice']))

    def _rule_request(self, docs_per_tower, T, training):
        """-> (algorithm to request from the C step, the one that will actually run, probe this step?)"""
        lib = _lib.lib()
        if getattr(self, 'gemm_math', 'f32') == 'f16x2':
            self._math_steps = getattr(self, '_math_steps', 0) + int(bool(training))
            if _MATH_OWNER[0] is not self or (training and self._math_steps % GEMM_MATH_REFRESH == 0):
                apply_gemm_math(self.table, self._conv_weights())       # (several engines in one process: re-own)
                _MATH_OWNER[0] = self
        req, probe = self.conv_algo, False
        static = lib.r4r_conv_algo(0, docs_per_tower, T, self.E, 100) if req == 0 else req
        # Training launches the static rule sends to the direct conv (E < 128, under 65,536 positions) start on
        # the projection as well when the vocabulary is small enough for its compaction to be cheap: since the
        # projection GEMM cuts a partial round into column parts they are faster there too (E = 64, B = 8 .. 64:
        # 0.040 .. 0.059 ms against 0.050 .. 0.082), and the probe below lets the measured rule confirm or revert.
        small = (req == 0 and static != 2 and bool(training) and self.V <= self.SMALL_LAUNCH_MAX_V
                 and lib.r4r_conv_algo(2, docs_per_tower, T, self.E, 100) == 2)       # (== 2: no R4R_CONV_ALGO pin)
        if req == 0 and (static == 2 or small):
            n = self._rule_n
            probe = bool(training) and n >= self.PROBE_AT and (n - self.PROBE_AT) % self.PROBE_EVERY == 0
            req = 2 if (probe or self._rule_choice is None) else self._rule_choice
            if training:
                self._rule_n += 1
        return req, lib.r4r_conv_algo(req, docs_per_tower, T, self.E, 100), probe

    def _rule_peek(self, docs_per_tower, T, ahead=0):
        """For spans of TRAINING steps.  -> (how many consecutive steps, starting `ahead` steps from now, run on one
        request without a probe -- 0: that step probes --, the request, the algorithm that runs, whether the rule
        counts these steps).  Changes nothing; `_rule_advance` after the steps ran."""
        if getattr(self, 'gemm_math', 'f32') == 'f16x2':
            return 0, None, None, False                      # (the fp16-split GEMM refreshes its scales between steps)
        # (r4r_conv_algo is a pure function of its arguments and the process environment: asked once per shape -- a span
        # call is on the critical path of a short timed region, and three C calls per look were 5 us of it)
        memo = self.__dict__.setdefault('_rule_memo', {})
        key = (docs_per_tower, T)
        if key not in memo:
            lib = _lib.lib()
            memo[key] = tuple(lib.r4r_conv_algo(r, docs_per_tower, T, self.E, 100) for r in (0, 1, 2))
        resolved = memo[key]
        req = self.conv_algo
        static = resolved[0] if req == 0 else req
        small = req == 0 and static != 2 and self.V <= self.SMALL_LAUNCH_MAX_V and resolved[2] == 2
        if not (req == 0 and (static == 2 or small)):
            return 1 << 30, req, resolved[req], False
        n = self._rule_n + int(ahead)
        to_probe = self.PROBE_AT - n if n <= self.PROBE_AT else (self.PROBE_AT - n) % self.PROBE_EVERY
        req = 2 if self._rule_choice is None else self._rule_choice
        return to_probe, req, resolved[req], True

    def _rule_advance(self, steps, counted):
        if counted:
            self._rule_n += int(steps)

    def _rule_decide(self, counters, docs_total, T):
        """`counters`: per tower the int32 [live, last] pair of the token buffer the probe step used."""
        torch.cuda.current_stream(self.dev).synchronize()
        self.conv_rows = int(sum(int(c.view(torch.int32)[1]) for c in counters))
        if self.conv_rows > 0:
            self._rule_choice = int(_lib.lib().r4r_conv_pick(self.E, T, docs_total, self.conv_rows, self.V))

    def _conv_weights(self):
        """The towers' conv weights (what the fp16-split GEMM's weight scale is taken from)."""
        return [p for k, p in self.model.named_parameters() if k.endswith('convs.0.weight')]

    @property
    def conv_choice(self):
        """'project' / 'direct' once measured, None before the first probe (or when conv_algo pins one)."""
        return {None: None, 1: 'direct', 2: 'project'}[self._rule_choice]


class DeepCoNNEngine(_ConvRule, _Spans):
    def __init__(self, model, lr=0.002, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, dp=None,
                 seed=0x5EED5EED, rank=0, conv_algo=0):
        hp = model.hyper_params
        if hp['model_type'] != 'deepconn':
            raise ValueError("DeepCoNNEngine implements model_type 'deepconn' (FM head); use the module path "
                             "+ reviews4rec_amd.optim.Adam for %r" % (hp['model_type'],))
        self.model, self.hp, self.dp = model, hp, dp
        self.conv_algo = int(conv_algo)          # 0 auto, 1 direct conv, 2 project-then-gather (include/r4r.h)
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.table = model.word2vec.weight
        if not self.table.is_cuda:
            raise RuntimeError('DeepCoNNEngine: move the model to a ROCm device first; the HIP path has no '
                               'CPU fallback')
        self.dev = self.table.device
        self.E_model = int(self.table.shape[1])
        self.table = padded_word_table(self.table)           # word_embed_size % 4 != 0: zero-padded copy (exact)
        self.V, self.E = self.table.shape
        self.L = hp['latent_size']
        self.gemm_math = apply_gemm_math(self.table, self._conv_weights())
        lib = _lib.lib()
        n = lib.r4r_deepconn_nparam()
        off, size, total = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)(), ctypes.c_int64()
        _lib.check(lib.r4r_deepconn_layout(self.E, self.L, off, size, ctypes.byref(total)), 'r4r_deepconn_layout')
        m = model
        self.slots = [m.user_conv.convs[0].weight, m.user_conv.convs[0].bias, m.user_conv.fc.weight,
                      m.user_conv.fc.bias, m.item_conv.convs[0].weight, m.item_conv.convs[0].bias,
                      m.item_conv.fc.weight, m.item_conv.fc.bias, m.fm.V, m.fm.lin.weight, m.fm.lin.bias,
                      m.global_bias]
        self.offsets, self.sizes, self.total = list(off), list(size), int(total.value)
        self.flat_p = torch.zeros(self.total, dtype=torch.float32, device=self.dev)
        for p, o, s in zip(self.slots, self.offsets, self.sizes):
            view = slot_view(self.flat_p, o, s, p.shape, self.E_model, self.E)
            assert view.shape == p.shape and (p.numel() == s or self.E != self.E_model), (tuple(p.shape), s)
            view.copy_(p.data)
            p.data = view                       # the Parameter now aliases the flat buffer
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.sse = torch.zeros(1, dtype=torch.float32, device=self.dev)     # running sum of SE
        self.step_count = 0
        self.seed = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        # data-parallel gradient exchange: 'allreduce' (all-reduce flat_g, then r4r_adam_multi) or
        # 'gather' (all_gather the flat buffers, sum + Adam in one launch); see autotune_exchange()
        self.exchange = os.environ.get('R4R_DP_EXCHANGE', 'allreduce')
        if self.exchange not in ('allreduce', 'gather', 'peer'):
            raise ValueError("R4R_DP_EXCHANGE must be 'allreduce', 'gather' or 'peer', got %r" % (self.exchange,))
        self._peer = None               # dist.PeerExchange, built on first use ('peer': device-side exchange over mapped buffers)
        self._gathered = None
        self._ws = None
        self._ws_key = None
        self._out = {}
        # token-state double buffering (project-then-gather): see prefetch_tokens()
        self._side = None
        self._prepared = None           # (key, buffer, event) of a batch whose tokens are already compacted
        self._last_buf = 1
        self._step_done = [None, None]  # per buffer: event after the last step that read it
        self._rule_reset()

    # ------------------------------------------------------------------ buffers
    def _workspace(self, B, T):
        key = (B, T)
        if self._ws_key != key:
            nb = _lib.lib().r4r_deepconn_ws_bytes(B, T, self.E, self.L, self.V)
            # zero-filled: the token-flag region must be all-zero on first use (kept zero by the kernels)
            self._ws = torch.zeros(max(nb, 256), dtype=torch.uint8, device=self.dev)
            self._ws_key = key
            self._prepared = None       # token state lived in the old workspace
            self._step_done = [None, None]
        return self._ws

    def _outputs(self, B):
        if B not in self._out:
            self._out[B] = (torch.empty(B, dtype=torch.float32, device=self.dev),
                            torch.empty(B, dtype=torch.float32, device=self.dev))
        return self._out[B]

    def dropout_multipliers(self, B, T):
        """[B, 2L] multipliers the last training step drew (columns: user tower L, item tower L)."""
        off = _lib.lib().r4r_deepconn_ws_mult_offset(B, T, self.E, self.L, self.V)
        raw = self._workspace(B, T)[off:off + B * 2 * self.L * 4]
        return raw.view(torch.float32).view(B, 2 * self.L).clone()

    # -------------------------------------------------------------------- steps
    def _indices(self, data):
        user_idx, item_idx = data[3], data[4]
        n = data[5].numel()
        user_idx = user_idx.reshape(n, -1)
        item_idx = item_idx.reshape(n, -1)
        if not (user_idx.is_cuda and user_idx.dtype == torch.int64):
            raise RuntimeError('DeepCoNNEngine: batches must be int64 tensors on the ROCm device')
        return user_idx.contiguous(), item_idx.contiguous(), n

    @staticmethod
    def _key(user_idx, item_idx, n):
        return (user_idx.data_ptr(), item_idx.data_ptr(), n, user_idx.shape[1])

    def prefetch_tokens(self, next_data):
        """Compact the distinct tokens of the NEXT batch on a side stream, concurrently with the
        step that is running now (the compaction depends only on the indices; the running step's
        projection GEMM is MFMA-bound and leaves the memory system idle).  The next train_step /
        predict on that very batch then skips its own mark + compact launches."""
        user_idx, item_idx, n = self._indices(next_data)
        T = user_idx.shape[1]
        if self._ws_key != (n, T):
            return                                           # different shape: that step will build its own
        lib = _lib.lib()
        req = self.conv_algo or (self._rule_choice or 0)
        if lib.r4r_conv_algo(req, n, T, self.E, 100) != 2:
            return                                           # that step runs the direct conv: no token state
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream(self.dev)
        if self._prepared is not None:                       # an unused prepared state: drop it
            self._discard_prepared(main)
        buf = self._last_buf ^ 1                             # the buffer the running step does NOT use
        if self._step_done[buf] is not None:
            self._side.wait_event(self._step_done[buf])      # its last reader has finished
        else:
            self._side.wait_stream(main)                     # first use: order after everything issued so far
        rc = lib.r4r_deepconn_tokens(ptr(user_idx), ptr(item_idx), ptr(self._ws), self._ws.numel(), n, T,
                                     self.E, self.L, self.V, 2, buf, 0, self._side.cuda_stream)
        _lib.check(rc, 'r4r_deepconn_tokens')
        ev = torch.cuda.Event()
        ev.record(self._side)
        # keep the index tensors alive until the side stream is done with them
        self._prepared = (self._key(user_idx, item_idx, n), buf, ev, (user_idx, item_idx))

    def _discard_prepared(self, main):
        """Drop a prepared token state nobody consumed (a wrong guess, an eval in between)."""
        key, pbuf, pev, keep = self._prepared
        if pev is not None:
            main.wait_event(pev)
        u, i = keep
        _lib.check(_lib.lib().r4r_deepconn_tokens(ptr(u), ptr(i), ptr(self._ws), self._ws.numel(), key[2], key[3],
                                                  self.E, self.L, self.V, 2, pbuf, 1, main.cuda_stream),
                   'r4r_deepconn_tokens(discard)')
        if self._side is not None:
            dropped = torch.cuda.Event()
            dropped.record(main)
            self._step_done[pbuf] = dropped                  # the side stream must see the reset
        self._last_buf = pbuf ^ 1
        self._prepared = None

    def _launch(self, data, y, grad, training, inv_denom, next_data=None, adam_step=0):
        user_idx, item_idx, n = self._indices(data)
        T = user_idx.shape[1]
        pred, se = self._outputs(n)
        ws = self._workspace(n, T)
        p_drop = float(self.hp['dropout'])
        main = torch.cuda.current_stream(self.dev)
        req, algo, probe = self._rule_request(n, T, grad)
        projecting = algo == 2                               # only that algorithm has token state
        nxt = None
        if next_data is not None and grad and projecting:
            nu, ni, nn = self._indices(next_data)
            if (nn, nu.shape[1]) == (n, T):                  # same shape: same workspace layout
                nxt = (nu, ni)
        ready = 0
        if projecting and self._prepared is not None and self._prepared[0] == self._key(user_idx, item_idx, n):
            buf, ev = self._prepared[1], self._prepared[2]
            if ev is not None:
                main.wait_event(ev)
            self._prepared = None
            ready = 1
        else:
            if self._prepared is not None and (nxt is not None or not projecting):
                self._discard_prepared(main)                 # its buffer is needed, or nobody will consume it
            buf = (self._prepared[1] ^ 1) if self._prepared is not None else self._last_buf ^ 1
        rc = _lib.lib().r4r_deepconn_step(
            ptr(self.table), self.V, ptr(user_idx), ptr(item_idx), ptr(y), ptr(self.flat_p),
            ptr(self.flat_g) if grad else None, ptr(pred), ptr(se), ptr(self.sse) if y is not None else None,
            ptr(ws), ws.numel(), n, T, self.E, self.L, p_drop, int(training), self.seed, self.offset,
            float(inv_denom), req, buf, ready,
            ptr(nxt[0]) if nxt else None, ptr(nxt[1]) if nxt else None,
            ptr(self.flat_m) if adam_step else None, ptr(self.flat_v) if adam_step else None,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, int(adam_step), main.cuda_stream)
        _lib.check(rc, 'r4r_deepconn_step')
        self._step_launched = True
        if nxt is not None:
            self._prepared = (self._key(nxt[0], nxt[1], n), buf ^ 1, None, nxt)
        self._last_buf = buf
        if self._side is not None:                           # only pay for the event when prefetching is in use
            done = torch.cuda.Event()
            done.record(main)
            self._step_done[buf] = done
        if training and p_drop > 0.0:
            self.offset += n * 2 * self.L
        if probe and projecting:
            lib = _lib.lib()
            at = [lib.r4r_deepconn_ws_count_offset(n, T, self.E, self.L, self.V, t, buf) for t in range(2)]
            self._rule_decide([ws[a:a + 8] for a in at], 2 * n, T)
        return pred, se

    @step_or_nothing
    @torch.no_grad()
    def train_step(self, data, y, n_global=None, next_data=None):
        """One optimisation step on this rank's shard.  Returns the per-example SE tensor
        (device); the running sum is in ``self.sse``.  ``next_data``: the batch that will be
        trained on next, if known -- its token marks / compaction ride on this step's backward /
        gradient-reduce launches, and that step then starts at its projection GEMM."""
        n = data[5].numel()
        y = y.reshape(-1).contiguous()
        dp_on = self.dp is not None and self.dp.on
        if n == 0 and not dp_on:                             # nothing to train on: no step, no state change
            return torch.empty(0, dtype=torch.float32, device=self.dev)
        denom = float(n_global if n_global is not None else n * (self.dp.world if self.dp else 1))
        self.step_count += 1
        if dp_on and n == 0:                                 # an empty shard contributes a zero gradient
            self.flat_g.zero_()
            self._exchange_and_update(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.step_count)
            return torch.empty(0, dtype=torch.float32, device=self.dev)
        if not dp_on:
            # single process: the launch that finishes the gradients is also the Adam update
            _, se = self._launch(data, y, grad=True, training=self.model.training, inv_denom=1.0 / denom,
                                 next_data=next_data, adam_step=self.step_count)
            return se
        _, se = self._launch(data, y, grad=True, training=self.model.training, inv_denom=1.0 / denom,
                             next_data=next_data)
        self._exchange_and_update(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.step_count)
        return se

    # ------------------------------------------------------------------ spans (K steps per host call)
    def _span_ok(self, desc):
        return desc.review and len(desc.doc_shape) == 1

    def _span_limit(self, desc, ahead=0):
        return self._rule_peek(desc.batch_size, desc.doc_shape[0], ahead)[0]

    def _span_dp_ok(self):
        # the exchange must be one the C side can issue itself: RCCL through the package's own communicator
        return (self.exchange in ('allreduce', 'gather') and getattr(self.dp, 'stream_rccl', None) is not None
                and self.dp.stream_rccl.comm)

    @torch.no_grad()
    def _span(self, desc, first, steps, announce):
        """Steps on batches first .. first + steps - 1 of the loader behind `desc` in ONE C call (r4r_deepconn_span); the
        engine's per-step state moves as `steps` train_step calls with next_data would have moved it.  Data parallel
        (every rank a full batch): r4r_deepconn_span_dp -- gradients, RCCL's collective, the update, per step, from C."""
        lib = _lib.lib()
        B, T = desc.batch_size, desc.doc_shape[0]
        pred, se = self._outputs(B)
        ws = self._workspace(B, T)
        main = torch.cuda.current_stream(self.dev)
        _, req, algo, counted = self._rule_peek(B, T)
        projecting = algo == 2
        slots = _span_slots(desc, first)
        ready = 0
        if projecting and self._prepared is not None and self._prepared[0] == (slots[3], slots[4], B, T):
            buf, ev = self._prepared[1], self._prepared[2]
            if ev is not None:
                main.wait_event(ev)
            self._prepared = None
            ready = 1
        else:
            if self._prepared is not None:
                self._discard_prepared(main)
            buf = self._last_buf ^ 1
        training, p_drop = self.model.training, float(self.hp['dropout'])
        draws = B * 2 * self.L if (training and p_drop > 0.0) else 0
        done = ctypes.c_int64(0)
        if self.dp is not None and self.dp.on:
            if not self._span_dp_ok():
                raise RuntimeError('DeepCoNNEngine: the data-parallel span needs the on-stream RCCL communicator and the '
                                   "'allreduce' or 'gather' exchange (got %r)" % (self.exchange,))
            comm = self.dp.stream_rccl
            gather = self.exchange == 'gather'
            if gather:
                self._exchange_prepare('gather')
            fn = ctypes.cast(comm.lib.ncclAllGather if gather else comm.lib.ncclAllReduce, ctypes.c_void_p).value
            rc = lib.r4r_deepconn_span_dp(
                desc.words, int(first), int(steps), int(bool(announce)), ctypes.byref(desc.built), ctypes.byref(done),
                ptr(self.table), self.V, ptr(self.flat_p), ptr(self.flat_g), ptr(pred), ptr(se), ptr(self.sse), ptr(ws),
                ws.numel(), T, self.E, self.L, p_drop, int(training), self.seed, self.offset, draws,
                1.0 / float(B * self.dp.world), req, buf, ready, ptr(self.flat_m), ptr(self.flat_v), self.total, self.lr,
                self.betas[0], self.betas[1], self.eps, self.wd, self.step_count + 1, int(gather), fn, comm.comm,
                self.dp.world, ptr(self._gathered) if gather else None, main.cuda_stream)
        else:
            rc = lib.r4r_deepconn_span(
                desc.words, int(first), int(steps), int(bool(announce)), ctypes.byref(desc.built), ctypes.byref(done),
                ptr(self.table), self.V, ptr(self.flat_p), ptr(self.flat_g), ptr(pred), ptr(se), ptr(self.sse), ptr(ws),
                ws.numel(), T, self.E, self.L, p_drop, int(training), self.seed, self.offset, draws, 1.0 / float(B), req,
                buf, ready, ptr(self.flat_m), ptr(self.flat_v), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                self.step_count + 1, main.cuda_stream)
        k = int(done.value)                                  # (an error names the step it stopped at: the state follows)
        self.step_count += k
        self.offset += k * draws
        self._rule_advance(k, counted)
        if k:
            self._last_buf = (buf + k - 1) & 1
            if self._side is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                self._step_done = [ev, ev]
        if rc == 0 and announce and projecting:
            nx = _span_slots(desc, first + steps)
            self._prepared = ((nx[3], nx[4], B, T), self._last_buf ^ 1, None, (_RawPtr(nx[3], desc), _RawPtr(nx[4], desc)))
        _lib.check(rc, 'r4r_deepconn_span_dp' if (self.dp is not None and self.dp.on) else 'r4r_deepconn_span')

    def _exchange_prepare(self, how):
        """What this rank needs, by itself, before exchange form `how` can run (no collective in here)."""
        if how == 'gather' and self._gathered is None:
            self._gathered = torch.empty(self.dp.world * self.total, dtype=torch.float32, device=self.dev)

    def _exchange_and_update(self, p, g, m, v, step):
        """Sum the ranks' gradients and apply Adam update number `step` to (p, m, v)."""
        if self.exchange == 'peer':
            if self._peer is None:
                from .dist import PeerExchange
                self._peer = PeerExchange(self.total, self.dev, self.dp.group)
                self._peer_epoch = 0
            self._peer_epoch += 1                            # (its own count: autotune / scratch calls advance it too)
            gathered = self._peer.exchange(g, self._peer_epoch)
            # guarded by the exchange's timed_out word: after a wait that gave up (a rank that never raised its flag)
            # the slots are stale or partial, and every rank would apply a different sum -- the update becomes a
            # no-op on the device and check_exchange() raises on the host (predict / state_dict / every epoch end)
            rc = _lib.lib().r4r_adam_gathered_guarded(ptr(p), gathered, self.dp.world, ptr(g), ptr(m), ptr(v),
                                                      self.total, self.lr, self.betas[0], self.betas[1], self.eps,
                                                      self.wd, int(step), self._peer.local.data_ptr() + 4,
                                                      _lib.current_stream())
            _lib.check(rc, 'r4r_adam_gathered_guarded')
            return
        if self.exchange == 'gather':
            self._exchange_prepare('gather')
            self.dp.gather_flat(g, self._gathered)
            rc = _lib.lib().r4r_adam_gathered(ptr(p), ptr(self._gathered), self.dp.world, ptr(g), ptr(m), ptr(v),
                                              self.total, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                              int(step), _lib.current_stream())
            _lib.check(rc, 'r4r_adam_gathered')
            return
        self.dp.allreduce_flat(g)
        one = ctypes.c_uint64 * 1
        rc = _lib.lib().r4r_adam_multi(1, one(p.data_ptr()), one(g.data_ptr()), one(m.data_ptr()), one(v.data_ptr()),
                                       (ctypes.c_int64 * 1)(self.total), self.lr, self.betas[0], self.betas[1],
                                       self.eps, self.wd, int(step), None, _lib.current_stream())
        _lib.check(rc, 'r4r_adam_multi')

    def autotune_exchange(self, trials=20):
        """Time both gradient-exchange forms on the job's own fabric (scratch buffers: the model is
        not touched) and keep the faster one; every rank takes the same decision (the slowest
        rank's time counts).  Call it once, outside any timed region.  -> {'allreduce': ms, 'gather': ms}"""
        if not (self.dp is not None and self.dp.on):
            return {}
        scratch = [torch.zeros_like(self.flat_p) for _ in range(4)]
        keep, res = self.exchange, {}

        def agreed_ok(ok):
            f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.dev)
            torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MIN, group=self.dp.group)
            return int(f.item()) == 1

        def wait(limit_s):
            """Drain the stream with a deadline: a candidate whose collective never ends must not park this thread."""
            import time
            ev = torch.cuda.Event()
            ev.record()
            deadline = time.monotonic() + limit_s
            while not ev.query():
                if time.monotonic() > deadline:
                    raise RuntimeError('autotune_exchange: the %r exchange did not finish within %.0f s on this fabric; pin '
                                       'the other one with R4R_DP_EXCHANGE, or R4R_DP_RCCL=0 for torch.distributed\'s '
                                       'collectives' % (self.exchange, limit_s))
                time.sleep(0.0005)

        def time_one(how):
            self.exchange = how
            for _ in range(3):
                self._exchange_and_update(*scratch, 1)
            wait(float(os.environ.get('R4R_RCCL_CHECK_TIMEOUT', 30.0)))
            torch.distributed.barrier(group=self.dp.group)
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(trials):
                self._exchange_and_update(*scratch, 1)
            t1.record()
            wait(float(os.environ.get('R4R_RCCL_CHECK_TIMEOUT', 30.0)))
            return t0.elapsed_time(t1) / trials

        def candidates():
            out = {}
            for how in ('allreduce', 'gather'):
                # Everything a rank does ALONE for a form -- its buffers, its checks -- happens in _exchange_prepare, and
                # the ranks agree on the outcome BEFORE the form's first collective: a form one rank cannot set up is
                # dropped on every rank.  (A failure inside a collective cannot be agreed on any more -- the other ranks
                # are in it; there the deadline of wait() turns a hang into an error.)
                try:
                    self._exchange_prepare(how)
                    err = None
                except Exception as e:                       # noqa: BLE001
                    err = '%s: %s' % (type(e).__name__, e)
                if not agreed_ok(err is None):
                    import warnings
                    warnings.warn('autotune_exchange: the %r exchange cannot be set up here (%s); dropped on every rank'
                                  % (how, err or 'another rank reported it'), RuntimeWarning)
                    continue
                t = torch.tensor([time_one(how)], dtype=torch.float64, device=self.dev)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.dp.group)
                out[how] = float(t.item())
            return out

        res = candidates()
        if not res and getattr(self.dp, 'stream_rccl', None) is not None:
            # neither form works through the package's own communicator: every rank drops it together and the
            # exchange goes through torch.distributed's collectives (the same sums on the group's own stream)
            import warnings
            warnings.warn("autotune_exchange: both exchange forms failed on the on-stream communicator; using "
                          "torch.distributed's collectives", RuntimeWarning)
            comm, self.dp.stream_rccl = self.dp.stream_rccl, None
            comm.close(abort=True)                           # (every rank is here: the decision was agreed; a hung collective dies with it)
            res = candidates()
        if not res:
            self.exchange = keep
            raise RuntimeError('autotune_exchange: no gradient exchange form works on this job (see the warnings above)')
        self.exchange = keep if (os.environ.get('R4R_DP_EXCHANGE') and keep in res) else min(res, key=res.get)
        return res

    def check_exchange(self):
        """Raise if a peer exchange of this engine timed out (one int read back from the device: called by predict(),
        state_dict() and main.train at every epoch end, never per step)."""
        if self._peer is not None:
            self._peer.check()

    def close(self):
        """Release the peer exchange's IPC mappings and segment (collective: every rank calls it)."""
        if self._peer is not None:
            peer, self._peer = self._peer, None
            peer.check()
            peer.close()

    @torch.no_grad()
    def predict(self, data, y=None):
        """Eval-mode forward (no dropout, no gradients).  Returns (pred, se or None)."""
        self.check_exchange()
        if data[5].numel() == 0:                             # an empty batch: nothing to launch
            e = torch.empty(tuple(data[5].shape), dtype=torch.float32, device=self.dev)
            return e, (e.clone() if y is not None else None)
        if y is not None:
            y = y.reshape(-1).contiguous()
        pred, se = self._launch(data, y, grad=False, training=False, inv_denom=1.0)
        shape = tuple(data[5].shape)
        return pred.view(shape), (se.view(shape) if y is not None else None)

    def state_dict(self):
        """Optimiser-side state of the fused step (the weights themselves live in the model's
        state_dict): Adam moments and step count, the dropout stream position."""
        self.check_exchange()
        return {'exp_avg': self.flat_m.clone(), 'exp_avg_sq': self.flat_v.clone(), 'step': self.step_count,
                'dropout_offset': self.offset, 'lr': self.lr, 'weight_decay': self.wd, 'betas': self.betas,
                'eps': self.eps, 'conv_rule': self._rule_state()}

    def load_state_dict(self, sd):
        if sd['exp_avg'].numel() != self.total:
            raise ValueError('DeepCoNNEngine.load_state_dict: %d moment elements for a %d-element layout'
                             % (sd['exp_avg'].numel(), self.total))
        self._rule_load(sd.get('conv_rule'))
        self.flat_m.copy_(sd['exp_avg'].to(self.dev))
        self.flat_v.copy_(sd['exp_avg_sq'].to(self.dev))
        self.step_count = int(sd['step'])
        self.offset = int(sd['dropout_offset'])
        self.lr, self.wd = float(sd['lr']), float(sd['weight_decay'])
        self.betas, self.eps = tuple(sd['betas']), float(sd['eps'])

    def grads(self):
        """Named views of the flat gradient buffer (reference parameter names)."""
        names = ['user_conv.convs.0.weight', 'user_conv.convs.0.bias', 'user_conv.fc.weight', 'user_conv.fc.bias',
                 'item_conv.convs.0.weight', 'item_conv.convs.0.bias', 'item_conv.fc.weight', 'item_conv.fc.bias',
                 'fm.V', 'fm.lin.weight', 'fm.lin.bias', 'global_bias']
        return {k: slot_view(self.flat_g, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
                zip(names, self.slots, self.offsets, self.sizes)}

    def moments(self):
        names = list(self.grads())
        return ({k: slot_view(self.flat_m, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
                 zip(names, self.slots, self.offsets, self.sizes)},
                {k: slot_view(self.flat_v, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
                 zip(names, self.slots, self.offsets, self.sizes)})


class MFEngine(_SweepSchedule, _Spans):
    """Native step for model_type 'MF_dot' / 'bias_only' (csrc/mf_engine.hip, r4r_mf_step): forward,
    loss, backward and the dense Adam update of MF.py / main.py:56-60,94-96 in two launches; the
    dense gradient of an ID table is never materialised.  Same calling surface as DeepCoNNEngine
    (train_step / predict / sse / state_dict).  Under data parallelism (``dp``) the step splits into
    r4r_mf_grad -> one all_gather of the ranks' compact rows -> r4r_mf_apply; with ``R4R_DP_EXCHANGE=peer`` (and at most
    PEER_MAX_ENTRIES gathered ratings) into r4r_mf_grad_push -> r4r_mf_apply_peer, the exchange riding on the two launches
    over peer-mapped memory (dist.PeerExchange's segments) with no collective call in the step."""
    MAX_TRAIN_BATCH = 1 << 20    # r4r_mf_step's limit (csrc/mf_engine.hip: MF_MAX_B_STEP)
    PEER_MAX_ENTRIES = 2048      # r4r_mf_apply_peer's limit on world * B_pad (csrc/mf_engine.hip: MF_SCAN_MAX_B)

    def __init__(self, model, lr=0.002, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, seed=0x5EED5EED, rank=0,
                 dp=None):
        self.dp = dp if (dp is not None and dp.on) else None
        hp = model.hyper_params
        if hp['model_type'] not in ('MF_dot', 'bias_only'):
            raise ValueError("MFEngine implements model_type 'MF_dot' and 'bias_only', got %r" % (hp['model_type'],))
        self.model, self.hp = model, hp
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.has_tables = hp['model_type'] == 'MF_dot'
        self.D = int(hp['latent_size']) if self.has_tables else 0
        self.params = ([model.user_embedding.weight, model.item_embedding.weight] if self.has_tables else [None, None]) \
            + [model.user_bias, model.item_bias, model.global_bias]
        live = [p for p in self.params if p is not None]
        if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in live):
            raise RuntimeError('MFEngine: move the model to a ROCm device first (fp32, contiguous); the HIP '
                               'path has no CPU fallback')
        self.dev = live[0].device
        self.n_users, self.n_items = model.user_bias.numel(), model.item_bias.numel()
        self.m = [None if p is None else torch.zeros_like(p) for p in self.params]
        self.v = [None if p is None else torch.zeros_like(p) for p in self.params]
        self.sse = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.step_count = 0
        self.seed = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        self._ws, self._ws_B, self._out = None, None, {}
        # visit period of the temporally blocked table sweep (include/r4r.h; 1 = the plain dense sweep)
        self.sweep_period = self.configured_period(hp)
        # the schedule in force: every table element is current through step _tb_base, chunks have been visited on the
        # period-_tb_period schedule since (1: every step)
        self._tb_base, self._tb_period = 0, 1
        self._sd_hooks = flush_before_state_dict(self, model)
        self.exchange = os.environ.get('R4R_DP_EXCHANGE', '') if self.dp is not None else ''
        self._peer, self._peer_epoch, self._peer_B = None, 0, None   # dist.PeerExchange of the blocks ('peer'), built on first use

    TEMPORAL_SWEEP = True        # train_step(..., defer_sweep=True) + flush(): the table sweep, temporally blocked

    def _ptrs(self, tensors):
        return (ctypes.c_uint64 * 5)(*[0 if t is None else t.data_ptr() for t in tensors])

    def check_exchange(self):
        """Raise if a peer-mapped exchange of this engine timed out (one int read back from the device)."""
        if self._peer is not None:
            self._peer.check()

    def close(self):
        """Release the peer exchange's IPC mappings and segment (collective: every rank calls it)."""
        if self._peer is not None:
            peer, self._peer = self._peer, None
            peer.check()
            peer.close()

    def flush(self, check=True, last_step=None):
        """Apply every pending table update of the temporally blocked sweep (no-op when nothing is pending).
        last_step: the last COMPLETED step (default: step_count)."""
        step = self._pending(last_step)
        if step is None:
            return
        _lib.check(_lib.lib().r4r_mf_rows_flush(
            self._ptrs(self.params), self._ptrs(self.m), self._ptrs(self.v), self.n_users, self.n_items, self.D,
            ptr(self._ws), self._ws.numel(), self._ws_B, self._tb_period, self._tb_base,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step, _lib.current_stream()), 'r4r_mf_rows_flush')
        self._tb_base = step
        if check:
            self.check_announcements()

    def check_announcements(self):
        """Raise if the scheduled sweep ever found more than 8 updates pending (a step outside the schedule
        without a flush: cannot happen through this class)."""
        if self._ws is None or not self.has_tables:
            return
        off = _lib.lib().r4r_mf_ws_flag_offset(self._ws_B, self.D, self.n_users, self.n_items)
        if int(self._ws[off:off + 4].view(torch.int32).item()):
            raise RuntimeError('MFEngine: the temporally blocked table sweep found more pending updates than it can apply')

    def _workspace(self, B):
        """One workspace per batch size (the ragged last batch alternates with the full ones); the
        row tags at the head of the buffer are shared state, so they move with the switch."""
        if self._ws_B != B:
            cache = self.__dict__.setdefault('_ws_cache', {})
            nxt = cache.get(B)
            if nxt is None:
                nb = _lib.lib().r4r_mf_ws_bytes(B, self.D, self.n_users, self.n_items)
                nxt = cache[B] = torch.zeros(max(nb, 256), dtype=torch.uint8, device=self.dev)   # tags start at zero
            if self._ws is not None:
                keep = _lib.lib().r4r_mf_ws_persist_bytes(B, self.D, self.n_users, self.n_items)
                nxt[:keep].copy_(self._ws[:keep])
            self._ws, self._ws_B = nxt, B
        return self._ws

    def _launch(self, data, y, train_mode, inv_denom, adam_step, defer=False):
        uid, iid = data[5].reshape(-1), data[6].reshape(-1)
        if not (uid.is_cuda and uid.dtype == torch.int64):
            raise RuntimeError('MFEngine: batches must be int64 tensors on the ROCm device')
        uid, iid = uid.contiguous(), iid.contiguous()
        n = uid.numel()
        if not adam_step:
            self.flush()                                     # a forward reads the tables: they catch up first
        period, base, sweep_all, want = self._schedule(defer) if adam_step else (1, 0, 1, 1)
        if n not in self._out:
            self._out[n] = (torch.empty(n, dtype=torch.float32, device=self.dev),
                            torch.empty(n, dtype=torch.float32, device=self.dev))
        pred, se = self._out[n]
        ws = self._workspace(n)
        rc = _lib.lib().r4r_mf_step(
            ptr(uid), ptr(iid), ptr(y), self._ptrs(self.params),
            self._ptrs(self.m) if adam_step else None, self._ptrs(self.v) if adam_step else None,
            self.n_users, self.n_items, self.D, ptr(pred), ptr(se), ptr(self.sse) if adam_step else None,
            ptr(ws), ws.numel(), n, float(self.hp['dropout']), int(train_mode), self.seed, self.offset,
            float(inv_denom), period, base, sweep_all,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, int(adam_step),
            _lib.current_stream())
        _lib.check(rc, 'r4r_mf_step')
        self._step_launched = True
        if adam_step:
            self._scheduled(sweep_all, want, int(adam_step))
        if train_mode and float(self.hp['dropout']) > 0.0:
            self.offset += n * 2 * self.D
        return pred, se

    # ------------------------------------------------------------------ spans (K steps per host call)
    def _span_ok(self, desc):
        return not desc.review and desc.batch_size <= self.MAX_TRAIN_BATCH

    @torch.no_grad()
    def _span(self, desc, first, steps, announce):
        """`steps` consecutive train_step(defer_sweep=True) calls as ONE C call (r4r_mf_span)."""
        lib, B = _lib.lib(), desc.batch_size
        if B not in self._out:
            self._out[B] = (torch.empty(B, dtype=torch.float32, device=self.dev),
                            torch.empty(B, dtype=torch.float32, device=self.dev))
        pred, se = self._out[B]
        ws = self._workspace(B)
        training, p_drop = self.model.training, float(self.hp['dropout'])
        draws = B * 2 * self.D if (training and p_drop > 0.0) else 0
        want = self.sweep_period if self.has_tables else 1
        base, period, done = ctypes.c_int64(self._tb_base), ctypes.c_int(self._tb_period), ctypes.c_int64(0)
        rc = lib.r4r_mf_span(
            desc.words, int(first), int(steps), ctypes.byref(done), self._ptrs(self.params), self._ptrs(self.m),
            self._ptrs(self.v), self.n_users, self.n_items, self.D, ptr(pred), ptr(se), ptr(self.sse), ptr(ws), ws.numel(),
            p_drop, int(training), self.seed, self.offset, draws, 1.0 / float(B), self._tb_period, want,
            ctypes.byref(base), ctypes.byref(period), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
            self.step_count + 1, _lib.current_stream())
        k = int(done.value)
        self.step_count += k
        self.offset += k * draws
        self._tb_base, self._tb_period = int(base.value), int(period.value)
        if k and want > 1:
            self._tb_used = True
        _lib.check(rc, 'r4r_mf_span')

    @torch.no_grad()
    def _train_step_dp(self, data, y, n_global, defer=False):
        """Data parallel: this rank's compact gradient rows into a packed block, ONE
        all_gather of the blocks, then the same tagged sweep over all ranks' entries in rank order
        on every rank (r4r_mf_grad / r4r_mf_apply): replicas stay bit-identical.  defer (the same on every rank): the
        sweep runs on the schedule (include/r4r.h, r4r_mf_step)."""
        lib, dist = _lib.lib(), torch.distributed
        uid, iid = data[5].reshape(-1).contiguous(), data[6].reshape(-1).contiguous()
        n, world = uid.numel(), self.dp.world
        B_pad = int(self.hp.get('batch_size', 0))            # every rank's shard fits the configured batch: pad to it
        if n_global is not None and n > B_pad:
            # (taking the size-agreement branch on THIS rank only would leave the others in a different
            # collective: a hang, not an error)
            raise RuntimeError("data parallel: this rank's shard has %d ratings but hyper_params['batch_size'] is %d; "
                               "pass n_global=None to let the ranks agree on the sizes first" % (n, B_pad))
        sizes_known = n_global is not None                   # (the peer exchange's buffers are sized once, for the configured batch)
        if n_global is None:                                 # (otherwise agree on the sizes first: one more collective + a sync)
            sizes = torch.tensor([n], dtype=torch.int64, device=self.dev)
            all_sizes = torch.empty(world, dtype=torch.int64, device=self.dev)
            dist.all_gather_into_tensor(all_sizes, sizes, group=self.dp.group)
            B_pad = int(all_sizes.max().item())
            n_global = int(all_sizes.sum().item())
        key = ('dp', B_pad)
        if key not in self._out:
            nb = lib.r4r_mf_dp_block_bytes(B_pad, self.D)
            self._out[key] = (torch.empty(max(B_pad, 1), dtype=torch.float32, device=self.dev),
                              torch.empty(max(B_pad, 1), dtype=torch.float32, device=self.dev),
                              torch.zeros(nb, dtype=torch.uint8, device=self.dev),
                              torch.zeros(world * nb, dtype=torch.uint8, device=self.dev))
        pred, se, block, blocks = self._out[key]
        step = int(self.step_count)
        period, base, sweep_all, want = self._schedule(defer)
        ws = self._workspace(world * B_pad)
        pending = self.has_tables and period > 1             # (rows may carry pending updates: the forward catches them up)
        if self.exchange == 'peer' and sizes_known and 0 < world * B_pad <= self.PEER_MAX_ENTRIES:
            # no collective call: the block goes into every rank's gathered buffer from the gradient launch, the update
            # launch waits for the ranks' flags (csrc/peer.hip's protocol; two buffers alternate by step parity)
            from . import dist as _dist
            if self._peer is None or self._peer_B != B_pad:
                if self._peer is not None:
                    self.close()
                self._peer = _dist.PeerExchange(lib.r4r_mf_dp_block_bytes(B_pad, self.D) // 4, self.dev, self.dp.group)
                self._peer_B = B_pad
            peer = self._peer
            self._peer_epoch += 1
            epoch, par = self._peer_epoch & 0x7fffffff, self._peer_epoch & 1
            _lib.check(lib.r4r_mf_grad_push(
                ptr(uid), ptr(iid), ptr(y), self._ptrs(self.params),
                self._ptrs(self.m) if pending else None, self._ptrs(self.v) if pending else None,
                self.n_users, self.n_items, self.D, ptr(pred), ptr(se), None, n, B_pad, float(self.hp['dropout']),
                int(self.model.training), self.seed, self.offset, 1.0 / float(n_global),
                ptr(ws) if pending else None, period, base,
                self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step,
                peer._dst[par].data_ptr(), peer._flg.data_ptr(), peer.local.data_ptr(), peer.rank, world, epoch,
                _lib.current_stream()), 'r4r_mf_grad_push')
            self._step_launched = True
            _lib.check(lib.r4r_mf_apply_peer(
                peer.gathered[par], world, B_pad, self._ptrs(self.params), self._ptrs(self.m), self._ptrs(self.v),
                self.n_users, self.n_items, self.D, ptr(ws), ws.numel(), period, base, sweep_all, ptr(se), n, ptr(self.sse),
                self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step,
                peer._mine, epoch, peer.local.data_ptr() + 4, peer.TIMEOUT_S, _lib.current_stream()), 'r4r_mf_apply_peer')
            return self._dp_step_done(sweep_all, want, step, n, se)
        _lib.check(lib.r4r_mf_grad(ptr(uid), ptr(iid), ptr(y), self._ptrs(self.params),
                                   self._ptrs(self.m) if pending else None, self._ptrs(self.v) if pending else None,
                                   self.n_users, self.n_items, self.D,
                                   ptr(pred), ptr(se), ptr(block), None, n, B_pad, float(self.hp['dropout']),
                                   int(self.model.training), self.seed, self.offset, 1.0 / float(n_global),
                                   ptr(ws) if pending else None, period, base,
                                   self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step,
                                   _lib.current_stream()), 'r4r_mf_grad')
        self._step_launched = True
        self.dp.all_gather(blocks, block)
        _lib.check(lib.r4r_mf_apply(ptr(blocks), world, B_pad, self._ptrs(self.params), self._ptrs(self.m),
                                    self._ptrs(self.v), self.n_users, self.n_items, self.D, ptr(ws), ws.numel(),
                                    period, base, sweep_all, ptr(se), n, ptr(self.sse),
                                    self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step,
                                    _lib.current_stream()), 'r4r_mf_apply')
        return self._dp_step_done(sweep_all, want, step, n, se)

    def _dp_step_done(self, sweep_all, want, step, n, se):
        self._scheduled(sweep_all, want, step)
        if self.model.training and float(self.hp['dropout']) > 0.0:
            self.offset += n * 2 * self.D
        # (this rank's share of the running metric -- the host loop sums the ranks -- rode on r4r_mf_apply)
        return se[:n]

    @step_or_nothing
    @torch.no_grad()
    def train_step(self, data, y, n_global=None, next_data=None, defer_sweep=False):
        """One optimisation step.  Returns the per-example SE tensor (device); the running sum is
        in ``self.sse``.  defer_sweep: the Adam sweep over the two ID tables is temporally blocked -- a chunk no
        rating names is visited every `sweep_period`-th step and takes its pending updates together, rows a rating
        names catch up on the way: same bits, a fraction of the traffic.  `flush()`, `state_dict()`, `predict()` and a
        step without defer_sweep bring the tables up to date; code that reads the embedding Parameters directly calls
        `flush()` before.  (next_data: accepted for the engines' common calling surface; nothing is announced.)"""
        n = data[5].numel()
        y = y.reshape(-1).contiguous()
        if n == 0 and self.dp is None:                       # nothing to train on: no step, no state change
            return torch.empty(0, dtype=torch.float32, device=self.dev)
        self.step_count += 1
        if self.dp is not None:
            return self._train_step_dp(data, y, n_global, defer_sweep)
        _, se = self._launch(data, y, self.model.training, 1.0 / float(n_global if n_global is not None else n),
                             self.step_count, defer_sweep)
        return se

    @torch.no_grad()
    def predict(self, data, y=None):
        """Eval-mode forward (no dropout, no gradients).  Returns (pred, se or None)."""
        if data[5].numel() == 0:                             # an empty batch: nothing to launch
            e = torch.empty(tuple(data[5].shape), dtype=torch.float32, device=self.dev)
            return e, (e.clone() if y is not None else None)
        if y is not None:
            y = y.reshape(-1).contiguous()
        pred, se = self._launch(data, y, False, 1.0, 0)
        shape = tuple(data[5].shape)
        return pred.view(shape), (se.view(shape) if y is not None else None)

    def dropout_multipliers(self, B):
        """[B, 2D] multipliers the last training step drew (user row D, item row D)."""
        off = _lib.lib().r4r_mf_ws_mult_offset(B, self.D, self.n_users, self.n_items)
        return self._workspace(B)[off:off + B * 2 * self.D * 4].view(torch.float32).view(B, 2 * self.D).clone()

    def dense_grads(self, data):
        """Dense gradients of the LAST training step, rebuilt from its compact rows (introspection
        for tests; the step itself never builds them).  `data`: the batch of that step."""
        uid, iid = data[5].reshape(-1), data[6].reshape(-1)
        B = uid.numel()
        lib, ws = _lib.lib(), self._workspace(B)

        def view(which, cols):
            off = lib.r4r_mf_ws_grad_offset(B, self.D, self.n_users, self.n_items, which)
            return ws[off:off + B * cols * 4].view(torch.float32).view(B, cols) if cols else None
        g = view(2, 1)[:, 0]
        out = {'user_bias': torch.zeros_like(self.params[2]).index_add_(0, uid, g),
               'item_bias': torch.zeros_like(self.params[3]).index_add_(0, iid, g),
               'global_bias': g.sum().reshape(1)}
        if self.has_tables:
            out['user_embedding.weight'] = torch.zeros_like(self.params[0]).index_add_(0, uid, view(0, self.D))
            out['item_embedding.weight'] = torch.zeros_like(self.params[1]).index_add_(0, iid, view(1, self.D))
        return out

    def moments(self):
        self.flush()
        names = ['user_embedding.weight', 'item_embedding.weight', 'user_bias', 'item_bias', 'global_bias']
        return ({k: t for k, t in zip(names, self.m) if t is not None},
                {k: t for k, t in zip(names, self.v) if t is not None})

    def state_dict(self):
        m, v = self.moments()
        return {'exp_avg': {k: t.clone() for k, t in m.items()}, 'exp_avg_sq': {k: t.clone() for k, t in v.items()},
                'step': self.step_count, 'dropout_offset': self.offset, 'lr': self.lr, 'weight_decay': self.wd,
                'betas': self.betas, 'eps': self.eps}

    def load_state_dict(self, sd):
        # validate FIRST: a rejected checkpoint must leave the engine as it was -- its sweep schedule included (what is
        # pending under the schedule is discarded only together with the state it belongs to).  The moment views are taken
        # WITHOUT a flush: main.py:306-308's order is model.load_state_dict, then this -- the tables already hold the
        # checkpoint's rows, and the pending gradient-zero updates of the state being replaced (old moments, old step
        # numbers) must not be applied to them.
        keep = (self._tb_base, self._tb_period)
        self._tb_base = self.step_count                      # nothing pending while moments() builds its views
        try:
            m, v = self.moments()
            scalars = load_named_moments('MFEngine', m, v, sd)
        except BaseException:
            self._tb_base, self._tb_period = keep
            raise
        self.step_count, self.offset, self.lr, self.wd, self.betas, self.eps = scalars
        self._tb_base, self._tb_period = self.step_count, 1  # the loaded tables are current through the loaded step
        # row tags written by earlier steps of THIS process must not collide with resumed step numbers
        for ws in self.__dict__.get('_ws_cache', {}).values():
            ws.zero_()


class NarreEngine(_ConvRule, _Spans):
    """Native step for NARRE (csrc/narre_engine.hip, r4r_narre_step): TextCNN over the B*R review
    documents of each side, both attention scorers, the ID vectors, `final`, the bias head, SE,
    the backward and the dense Adam update in five launches (the op-by-op path needs ~130).
    Dense parameters live in one flat buffer (the module's Parameters alias it), the ID tables
    and bias vectors stay where they are and are updated by a tagged sweep that never builds
    their dense gradient.  Same surface as DeepCoNNEngine; single process only."""

    NAMES = ['user_conv.convs.0.weight', 'user_conv.convs.0.bias', 'user_conv.fc.weight', 'user_conv.fc.bias',
             'item_conv.convs.0.weight', 'item_conv.convs.0.bias', 'item_conv.fc.weight', 'item_conv.fc.bias',
             'attention_scorer_user.0.weight', 'attention_scorer_user.0.bias', 'attention_scorer_user.3.weight',
             'attention_scorer_user.3.bias', 'attention_scorer_item.0.weight', 'attention_scorer_item.0.bias',
             'attention_scorer_item.3.weight', 'attention_scorer_item.3.bias', 'final.1.weight', 'final.1.bias',
             'final.3.weight', 'final.3.bias', 'global_bias']
    ROW_NAMES = ['user_embedding.weight', 'item_embedding.weight', 'user_bias', 'item_bias']
    MODEL_TYPE = 'NARRE'
    C = 'narre'                  # prefix of the C entry points
    NTOWER = 2                   # TextCNN towers (token states per buffer)

    def __init__(self, model, lr=0.002, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, seed=0x5EED5EED, rank=0,
                 conv_algo=0, dp=None):
        if dp is not None and dp.on:
            self.dp = dp
        hp = model.hyper_params
        if hp['model_type'] not in (self.MODEL_TYPE if isinstance(self.MODEL_TYPE, tuple) else (self.MODEL_TYPE,)):
            raise ValueError('%s implements model_type %r, got %r' % (type(self).__name__, self.MODEL_TYPE, hp['model_type']))
        self.model, self.hp = model, hp
        self.conv_algo = int(conv_algo)
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.table = self._word_table(model)
        if not self.table.is_cuda:
            raise RuntimeError('NarreEngine: move the model to a ROCm device first; the HIP path has no CPU fallback')
        self.dev = self.table.device
        self.E_model = int(self.table.shape[1])
        self.table = padded_word_table(self.table)           # word_embed_size % 4 != 0: zero-padded copy (exact)
        self.V, self.E = self.table.shape
        self.L = int(hp['latent_size'])
        self.gemm_math = apply_gemm_math(self.table, self._conv_weights())
        lib = _lib.lib()
        n = getattr(lib, 'r4r_%s_nparam' % self.C)()
        off, size, total = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)(), ctypes.c_int64()
        _lib.check(getattr(lib, 'r4r_%s_layout' % self.C)(self.E, self.L, *self._layout_extra(), off, size,
                                                          ctypes.byref(total)), 'layout')
        params = dict(model.named_parameters())
        self.slots = [params[k] for k in self.NAMES]
        self.offsets, self.sizes, self.total = list(off), list(size), int(total.value)
        self.flat_p = torch.zeros(self.total, dtype=torch.float32, device=self.dev)
        for p, o, s in zip(self.slots, self.offsets, self.sizes):
            view = slot_view(self.flat_p, o, s, p.shape, self.E_model, self.E)
            assert view.shape == p.shape and (p.numel() == s or self.E != self.E_model), (tuple(p.shape), s)
            view.copy_(p.data)
            p.data = view                       # the Parameter now aliases the flat buffer
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.rows = [params[k] for k in self.ROW_NAMES]
        if not all(p.is_contiguous() and p.dtype == torch.float32 for p in self.rows):
            raise RuntimeError('NarreEngine: fp32 contiguous ID tables / bias vectors only')
        self.rows_m = [torch.zeros_like(p) for p in self.rows]
        self.rows_v = [torch.zeros_like(p) for p in self.rows]
        self.n_users, self.n_items = self._cardinalities()
        self.sse = torch.zeros(self.SSE_SLOTS, dtype=torch.float32, device=self.dev)
        self.step_count = 0
        self.seed = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        self._ws, self._ws_key, self._out = None, None, {}
        self._prepared, self._last_buf = None, 1
        self._algo_req = self.conv_algo
        self._rule_reset()

    SSE_SLOTS = 1

    @staticmethod
    def _word_table(model):
        return model.word2vec.weight

    def _layout_extra(self):
        return ()

    def _cardinalities(self):
        return self.rows[-2].numel(), self.rows[-1].numel()

    @staticmethod
    def _p4(tensors):
        return (ctypes.c_uint64 * 4)(*[t.data_ptr() for t in tensors])

    def _fields(self, data):
        n = data[5].numel()
        ur, ir = data[3], data[4]
        R, T = ur.shape[-2], ur.shape[-1]
        if tuple(ir.shape[-2:]) != (R, T):
            raise RuntimeError('NarreEngine: user and item reviews must share [num_reviews, num_words]')
        who, rev = data[1].reshape(n, -1), data[2].reshape(n, -1)
        if who.shape[1] != R or rev.shape[1] != R:
            raise RuntimeError('NarreEngine: %d neighbour ids for %d reviews (NARRE.py concatenates them)'
                               % (who.shape[1], R))
        f = [ur.reshape(n, R, T), ir.reshape(n, R, T), rev, who, data[5].reshape(-1), data[6].reshape(-1)]
        if not all(t.is_cuda and t.dtype == torch.int64 for t in f):
            raise RuntimeError('NarreEngine: batches must be int64 tensors on the ROCm device')
        return [t.contiguous() for t in f], n, R, T

    def _workspace(self, B, R, T):
        key = (B, R, T)
        if self._ws_key != key:
            cache = self.__dict__.setdefault('_ws_cache', {})
            nxt = cache.get(key)
            if nxt is None:
                nb = self._ws_bytes(B, R, T)
                nxt = cache[key] = torch.zeros(max(nb, 256), dtype=torch.uint8, device=self.dev)
            if self._ws is not None:                         # the row tags head the buffer: shared state
                keep = self._persist_bytes(B, R, T)
                nxt[:keep].copy_(self._ws[:keep])
            self._ws, self._ws_key = nxt, key
            self._prepared = None                            # token state lived in the other workspace
        return self._ws

    def _persist_bytes(self, B, R, T):
        return 256 * (-(-self.n_users * 4 // 256) + -(-self.n_items * 4 // 256))

    def _ws_bytes(self, B, R, T):
        return _lib.lib().r4r_narre_ws_bytes(B, R, T, self.E, self.L, self.V, self.n_users, self.n_items)

    def _ws_offset(self, B, R, T, which):
        return _lib.lib().r4r_narre_ws_offset(B, R, T, self.E, self.L, self.V, self.n_users, self.n_items, which)

    def _draws(self, R):
        return 4 * R * self.L + 3 * self.L                   # dropout draws per rating

    def _step(self, f, y, pred, se, ws, n, R, T, train_mode, inv_denom, adam_step, buf, ready, nxt):
        return _lib.lib().r4r_narre_step(
            ptr(self.table), self.V, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), ptr(f[4]), ptr(f[5]), ptr(y),
            ptr(self.flat_p), ptr(self.flat_g) if adam_step else None,
            ptr(self.flat_m) if (adam_step and not self._grads_only()) else None,     # data parallel / split step: gradients only
            ptr(self.flat_v) if (adam_step and not self._grads_only()) else None, self._p4(self.rows),
            self._p4(self.rows_m) if (adam_step and not self._grads_only()) else None,
            self._p4(self.rows_v) if (adam_step and not self._grads_only()) else None,
            self.n_users, self.n_items, ptr(pred), ptr(se), ptr(self.sse) if adam_step else None,
            ptr(ws), ws.numel(), n, R, T, self.E, self.L, float(self.hp['dropout']), int(train_mode), self.seed,
            self.offset, float(inv_denom), self._algo_req, buf, ready,
            ptr(nxt[0]) if nxt else None, ptr(nxt[1]) if nxt else None,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, int(adam_step), _lib.current_stream())

    def _launch(self, data, y, train_mode, inv_denom, adam_step, next_data=None):
        f, n, R, T = self._fields(data)
        if n not in self._out:
            self._out[n] = (torch.empty(n, dtype=torch.float32, device=self.dev),
                            torch.empty(n, dtype=torch.float32, device=self.dev))
        pred, se = self._out[n]
        ws = self._workspace(n, R, T)
        self._algo_req, algo, probe = self._rule_request(n * R, T, bool(adam_step))
        projecting = algo == 2                               # only that algorithm has token state
        nxt = None
        if next_data is not None and adam_step and projecting:
            nf, nn, nR, nT = self._fields(next_data)
            if (nn, nR, nT) == (n, R, T):
                nxt = nf
        key = tuple(t.data_ptr() for t in f[:self.NTOWER]) + (n, R, T)
        ready = 0
        if projecting and self._prepared is not None and self._prepared[0] == key:
            buf, ready = self._prepared[1], 1
            self._prepared = None
        else:
            if self._prepared is not None:                   # a wrong guess, or a step that will not consume it:
                pb = self._prepared[1]                       # drop its token state
                for t in range(self.NTOWER):                 # the compaction counters of that buffer
                    at = self._ws_offset(n, R, T, 6 + 2 * t + pb)
                    ws[at:at + 4].zero_()
                self._prepared = None
                self._last_buf = pb ^ 1
            buf = self._last_buf ^ 1
        rc = self._step(f, y, pred, se, ws, n, R, T, train_mode, inv_denom, adam_step, buf, ready, nxt)
        _lib.check(rc, 'r4r_%s_step' % self.C)
        self._step_launched = True
        self._last_buf = buf
        if nxt is not None:
            self._prepared = (tuple(t.data_ptr() for t in nxt[:self.NTOWER]) + (n, R, T), buf ^ 1, nxt)
        if train_mode and float(self.hp['dropout']) > 0.0:
            self.offset += n * self._draws(R)
        if probe and projecting:
            at = [self._ws_offset(n, R, T, 6 + 2 * t + buf) for t in range(self.NTOWER)]
            self._rule_decide([ws[a:a + 8] for a in at], self.NTOWER * n * R, T)
        return pred, se

    # ------------------------------------------------------------------ spans (K steps per host call)
    def _span_doc(self, desc):
        """(R, T) of the loader's documents if this family trains on them, else None."""
        if not desc.review or len(desc.doc_shape) != 2:
            return None
        R, W = desc.doc_shape
        if desc.batch_size * (1 + R) > self.FUSED_MAX_ENTRIES or self.L > 32 or R > 32:
            return None                                      # (train_step's split form: per step)
        return R, W

    def _span_towers(self, slots):
        return (slots[3], slots[4])                          # the documents the TextCNN towers read: user, item

    def _span_ok(self, desc):
        return self._span_doc(desc) is not None

    def _span_limit(self, desc, ahead=0):
        R, T = self._span_doc(desc)
        return self._rule_peek(desc.batch_size * R, T, ahead)[0]

    def _span_call(self, lib, desc, first, steps, announce, done, pred, se, ws, R, T, draws, buf, ready, stream):
        return lib.r4r_narre_span(
            desc.words, int(first), int(steps), int(bool(announce)), ctypes.byref(desc.built), ctypes.byref(done),
            ptr(self.table), self.V, ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v),
            self._p4(self.rows), self._p4(self.rows_m), self._p4(self.rows_v), self.n_users, self.n_items, ptr(pred),
            ptr(se), ptr(self.sse), ptr(ws), ws.numel(), R, T, self.E, self.L, float(self.hp['dropout']),
            int(self.model.training), self.seed, self.offset, draws, 1.0 / float(desc.batch_size), self._algo_req, buf,
            ready, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count + 1, stream)

    @torch.no_grad()
    def _span(self, desc, first, steps, announce):
        """`steps` consecutive train_step calls (each announcing its successor) as ONE C call (r4r_<family>_span)."""
        lib, B = _lib.lib(), desc.batch_size
        R, T = self._span_doc(desc)
        if B not in self._out:
            self._out[B] = (torch.empty(B, dtype=torch.float32, device=self.dev),
                            torch.empty(B, dtype=torch.float32, device=self.dev))
        pred, se = self._out[B]
        ws = self._workspace(B, R, T)
        _, self._algo_req, algo, counted = self._rule_peek(B * R, T)
        projecting = algo == 2
        key = self._span_towers(_span_slots(desc, first)) + (B, R, T)
        ready = 0
        if projecting and self._prepared is not None and self._prepared[0] == key:
            buf, ready = self._prepared[1], 1
            self._prepared = None
        else:
            if self._prepared is not None:                   # a state nobody will consume: drop it (as _launch does)
                pb = self._prepared[1]
                for t in range(self.NTOWER):
                    at = self._ws_offset(B, R, T, 6 + 2 * t + pb)
                    ws[at:at + 4].zero_()
                self._prepared = None
                self._last_buf = pb ^ 1
            buf = self._last_buf ^ 1
        draws = B * self._draws(R) if (self.model.training and float(self.hp['dropout']) > 0.0) else 0
        done = ctypes.c_int64(0)
        rc = self._span_call(lib, desc, first, steps, announce, done, pred, se, ws, R, T, draws, buf, ready,
                             _lib.current_stream())
        k = int(done.value)
        self.step_count += k
        self.offset += k * draws
        self._rule_advance(k, counted)
        if k:
            self._last_buf = (buf + k - 1) & 1
        if rc == 0 and announce and projecting:
            nx = self._span_towers(_span_slots(desc, first + steps))
            self._prepared = (nx + (B, R, T), self._last_buf ^ 1, [_RawPtr(a, desc) for a in nx])
        _lib.check(rc, 'r4r_%s_span' % self.C)

    # ---- data parallel: gradients only on this rank (the C step with flat_m = NULL), C1 -- one
    # all-reduce of the flat dense gradient + the flat Adam -- and C2: the ranks' compact ID rows gathered
    # (ids -1 pad ragged shards) into the same tagged sweep on every rank.  Subclasses with ID rows give
    # _dp_payload / _dp_apply.
    dp = None

    DP_COLS = 1                  # nonzero: the family has ID rows to exchange (subclasses without: 0)
    FUSED_MAX_ENTRIES = 4096     # B (1 + R) the fused launch's entry waves hold (csrc/step_device.h: NROW_MAX_ENTRIES)
    ROWS_APPLY_MAX = 16384       # ... and r4r_narre_rows_apply's; beyond it: r4r_narre_rows_apply_large (any count)
    _rows_scratch = None
    _split = False               # this step runs as gradients -> flat Adam -> row apply (the data-parallel form on one rank)

    def _grads_only(self):
        return self.dp is not None or self._split

    def _dp_doc_shape(self, data):
        """(reviews per rating, words per document) of this engine's batches, also for an empty shard."""
        return int(data[3].shape[-2]), int(data[3].shape[-1])  # (a shard's slice keeps these dims when empty)

    def _dp_cols(self, R):
        """(int64 ids, floats) per rating in the gathered payload."""
        return 2 + 2 * R, 1 + 2 * (1 + R) * self.L

    def _dp_payload(self, f, n, R, T, ids, vals):
        # per rating: [uid, iid, R neighbour ids of the user table, R of the item table] and
        # [d loss/d pred, (1 + R) x L gradient rows of the user table, then of the item table]
        ws, L = self._workspace(n, R, T), self.L
        view = lambda which, cols, dt: ws[self._ws_offset(n, R, T, which):][:n * (1 + R) * cols * (8 if dt == torch.int64 else 4)] \
            .view(dt).view(n * (1 + R), cols)                # noqa: E731
        off = self._ws_offset(n, R, T, 5)
        vals[:n, 0] = ws[off:off + n * 4].view(torch.float32)
        for t in range(2):
            gid, grow = view(3 + t, 1, torch.int64)[:, 0], view(1 + t, L, torch.float32)
            ids[:n, t] = gid[:n]
            ids[:n, 2 + t * R:2 + (t + 1) * R] = gid[n:].view(n, R)
            base = 1 + t * (1 + R) * L
            vals[:n, base:base + L] = grow[:n]
            vals[:n, base + L:base + (1 + R) * L] = grow[n:].reshape(n, R * L)

    def _dp_apply(self, all_ids, all_vals, B_all, ws, nb, R, T):
        L = self.L
        gids, grows = [], []
        for t in range(2):                                   # entries in (rank, rating, [self, neighbours]) order
            gids.append(torch.cat([all_ids[:, t:t + 1], all_ids[:, 2 + t * R:2 + (t + 1) * R]], dim=1).reshape(-1).contiguous())
            base = 1 + t * (1 + R) * L
            grows.append(all_vals[:, base:base + (1 + R) * L].reshape(-1, L).contiguous())
        g_entry = torch.zeros((B_all, 1 + R), dtype=torch.float32, device=self.dev)
        g_entry[:, 0] = all_vals[:, 0]
        entries = B_all * (1 + R)
        if entries > self.ROWS_APPLY_MAX or L > 32:          # beyond the entry waves that keep every id in LDS / a row in registers
            lib = _lib.lib()
            need = lib.r4r_rows_large_ws_bytes(entries)
            if self._rows_scratch is None or self._rows_scratch.numel() < need:
                self._rows_scratch = torch.empty(need, dtype=torch.uint8, device=self.dev)
            _lib.check(lib.r4r_narre_rows_apply_large(
                ptr(gids[0]), ptr(gids[1]), ptr(grows[0]), ptr(grows[1]), ptr(g_entry), entries,
                self._p4(self.rows), self._p4(self.rows_m), self._p4(self.rows_v), self.n_users, self.n_items,
                ptr(ws), ws.numel(), nb, R, T, self.E, L, self.V, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                int(self.step_count), ptr(self._rows_scratch), self._rows_scratch.numel(), _lib.current_stream()),
                'r4r_narre_rows_apply_large')
            return
        _lib.check(_lib.lib().r4r_narre_rows_apply(
            ptr(gids[0]), ptr(gids[1]), ptr(grows[0]), ptr(grows[1]), ptr(g_entry), B_all * (1 + R),
            self._p4(self.rows), self._p4(self.rows_m), self._p4(self.rows_v), self.n_users, self.n_items,
            ptr(ws), ws.numel(), nb, R, T, self.E, L, self.V, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
            int(self.step_count), _lib.current_stream()), 'r4r_narre_rows_apply')

    @torch.no_grad()
    def _train_step_dp(self, data, y, n_global, next_data):
        lib, dist = _lib.lib(), torch.distributed
        solo = self.dp is None                               # the split step of a single process: no collective at all
        n, world = data[5].numel(), (1 if solo else self.dp.world)
        B_pad = n if solo else int(self.hp.get('batch_size', 0))
        if solo and n_global is None:
            n_global = n
        if n_global is not None and n > B_pad:
            # (taking the size-agreement branch on THIS rank only would leave the others in a different
            # collective: a hang, not an error)
            raise RuntimeError("data parallel: this rank's shard has %d ratings but hyper_params['batch_size'] is %d; "
                               "pass n_global=None to let the ranks agree on the sizes first" % (n, B_pad))
        if n_global is None:
            sizes = torch.tensor([n], dtype=torch.int64, device=self.dev)
            all_sizes = torch.empty(world, dtype=torch.int64, device=self.dev)
            dist.all_gather_into_tensor(all_sizes, sizes, group=self.dp.group)
            B_pad, n_global = int(all_sizes.max().item()), int(all_sizes.sum().item())
        y = y.reshape(-1).contiguous()
        self.step_count += 1
        se = torch.empty(0, dtype=torch.float32, device=self.dev)
        if n > 0:
            _, se = self._launch(data, y, self.model.training, 1.0 / float(n_global), self.step_count, next_data)
        else:
            self.flat_g.zero_()                              # an empty shard contributes a zero gradient
        if not solo:
            self.dp.allreduce_flat(self.flat_g)
        one = ctypes.c_uint64 * 1
        _lib.check(lib.r4r_adam_multi(1, one(self.flat_p.data_ptr()), one(self.flat_g.data_ptr()),
                                      one(self.flat_m.data_ptr()), one(self.flat_v.data_ptr()),
                                      (ctypes.c_int64 * 1)(self.total), self.lr, self.betas[0], self.betas[1], self.eps,
                                      self.wd, int(self.step_count), None, _lib.current_stream()), 'r4r_adam_multi')
        if self.DP_COLS:
            self._dp_rows(data, n, B_pad, world, solo)
        return se

    BLOCKS_DP = True             # the entries travel as one packed block per rank (r4r_narre_dp_block; subclasses: their own forms)

    def _dp_rows(self, data, n, B_pad, world, solo):
        """The ID tables' half of the data-parallel step: the ranks' compact entries, gathered, into the same update on
        every rank.  Up to ROWS_APPLY_MAX entries (and latent_size 32): one packing launch, ONE all_gather, and the update
        reads the ranks' blocks directly; beyond, the generic form (per-rating payload rows, sliced on the host side)."""
        R, T = self._dp_doc_shape(data)
        if self.BLOCKS_DP and 0 < world * B_pad * (1 + R) <= self.ROWS_APPLY_MAX and self.L <= 32:
            lib = _lib.lib()
            key = ('dp_blocks', B_pad, world, R)
            if key not in self._out:
                nb_bytes = lib.r4r_narre_dp_block_bytes(B_pad, R, self.L)
                block = torch.zeros(nb_bytes, dtype=torch.uint8, device=self.dev)
                self._out[key] = (block, block if solo else torch.zeros(world * nb_bytes, dtype=torch.uint8, device=self.dev))
            block, blocks = self._out[key]
            nb = max(n, 1)                                   # (the workspace of this rank's own shape holds the row tags)
            ws = self._workspace(nb, R, T)
            _lib.check(lib.r4r_narre_dp_block(ptr(ws), ws.numel(), n, R, T, self.E, self.L, self.V, self.n_users, self.n_items,
                                              ptr(block), B_pad, _lib.current_stream()), 'r4r_narre_dp_block')
            if not solo:
                self.dp.all_gather(blocks, block)
            _lib.check(lib.r4r_narre_rows_apply_blocks(
                ptr(blocks), world, B_pad, self._p4(self.rows), self._p4(self.rows_m), self._p4(self.rows_v), self.n_users,
                self.n_items, ptr(ws), ws.numel(), nb, R, T, self.E, self.L, self.V, self.lr, self.betas[0], self.betas[1],
                self.eps, self.wd, int(self.step_count), _lib.current_stream()), 'r4r_narre_rows_apply_blocks')
            return
        id_cols, val_cols = self._dp_cols(R)
        ids = torch.full((B_pad, id_cols), -1, dtype=torch.int64, device=self.dev)
        vals = torch.zeros((B_pad, val_cols), dtype=torch.float32, device=self.dev)
        if n > 0:
            f, _, R, T = self._fields(data)
            self._dp_payload(f, n, R, T, ids, vals)
        if solo:
            all_ids, all_vals = ids, vals
        else:
            all_ids = torch.empty((world, B_pad, id_cols), dtype=torch.int64, device=self.dev)
            all_vals = torch.empty((world, B_pad, val_cols), dtype=torch.float32, device=self.dev)
            self.dp.all_gather(all_ids.view(-1), ids.view(-1))
            self.dp.all_gather(all_vals.view(-1), vals.view(-1))
        nb = max(n, 1)                                       # (the workspace of this rank's own shape holds the row tags)
        self._dp_apply(all_ids.view(world * B_pad, id_cols), all_vals.view(world * B_pad, val_cols), world * B_pad,
                       self._workspace(nb, R, T), nb, R, T)

    @step_or_nothing
    @torch.no_grad()
    def train_step(self, data, y, n_global=None, next_data=None):
        if self.dp is not None:
            return self._train_step_dp(data, y, n_global, next_data)
        n = data[5].numel()
        if type(self) is NarreEngine and n > 0 and (n * (1 + int(data[3].shape[-2])) > self.FUSED_MAX_ENTRIES
                                                    or self.L > 32 or int(data[3].shape[-2]) > 32):
            # more ID entries than the fused launch's entry waves hold (or rows wider than they keep in registers:
            # latent_size / narre_num_reviews 33 .. 64): the step runs in the data-parallel form --
            # gradients, the flat Adam, then the row apply -- on this one process
            self._split = True
            try:
                return self._train_step_dp(data, y, n_global, next_data)
            finally:
                self._split = False
        y = y.reshape(-1).contiguous()
        if n == 0:                                           # nothing to train on: no step, no state change
            return torch.empty(0, dtype=torch.float32, device=self.dev)
        self.step_count += 1
        _, se = self._launch(data, y, self.model.training, 1.0 / float(n_global if n_global is not None else n),
                             self.step_count, next_data)
        return se

    @torch.no_grad()
    def predict(self, data, y=None):
        if data[5].numel() == 0:                             # an empty batch: nothing to launch
            e = torch.empty(tuple(data[5].shape), dtype=torch.float32, device=self.dev)
            return e, (e.clone() if y is not None else None)
        if y is not None:
            y = y.reshape(-1).contiguous()
        pred, se = self._launch(data, y, False, 1.0, 0)
        shape = tuple(data[5].shape)
        return pred.view(shape), (se.view(shape) if y is not None else None)

    def _ws_view(self, data, which, cols, dtype=torch.float32):
        f, n, R, T = self._fields(data)
        off = self._ws_offset(n, R, T, which)
        rows = n if which in (0, 5) else n * (1 + R)
        item = 8 if dtype == torch.int64 else 4
        return self._workspace(n, R, T)[off:off + rows * cols * item].view(dtype).view(rows, cols)

    def dropout_multipliers(self, data):
        """[B, 4RL + 3L] multipliers the last training step drew (site-major, include/r4r.h)."""
        f, n, R, T = self._fields(data)
        return self._ws_view(data, 0, self._draws(R)).clone()

    def grads(self, data):
        """Gradients of the LAST training step by reference parameter name; the ID-table / bias
        gradients are rebuilt from their compact rows (introspection for tests)."""
        out = {k: slot_view(self.flat_g, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
               zip(self.NAMES, self.slots, self.offsets, self.sizes)}
        B = data[5].numel()
        g = self._ws_view(data, 5, 1)[:, 0]
        for t, name in enumerate(self.ROW_NAMES[:2]):
            ids = self._ws_view(data, 3 + t, 1, torch.int64)[:, 0]
            out[name] = torch.zeros_like(self.rows[t]).index_add_(0, ids, self._ws_view(data, 1 + t, self.L))
            out[self.ROW_NAMES[2 + t]] = torch.zeros_like(self.rows[2 + t]).index_add_(0, ids[:B], g)
        return out

    def moments(self):
        m = {k: slot_view(self.flat_m, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in zip(self.NAMES, self.slots, self.offsets, self.sizes)}
        v = {k: slot_view(self.flat_v, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in zip(self.NAMES, self.slots, self.offsets, self.sizes)}
        m.update(zip(self.ROW_NAMES, self.rows_m))
        v.update(zip(self.ROW_NAMES, self.rows_v))
        return m, v

    def state_dict(self):
        m, v = self.moments()
        return {'exp_avg': {k: t.clone() for k, t in m.items()}, 'exp_avg_sq': {k: t.clone() for k, t in v.items()},
                'step': self.step_count, 'dropout_offset': self.offset, 'lr': self.lr, 'weight_decay': self.wd,
                'betas': self.betas, 'eps': self.eps, 'conv_rule': self._rule_state()}

    def load_state_dict(self, sd):
        m, v = self.moments()
        self.step_count, self.offset, self.lr, self.wd, self.betas, self.eps = \
            load_named_moments(type(self).__name__, m, v, sd)
        self._rule_load(sd.get('conv_rule'))
        for ws in self.__dict__.get('_ws_cache', {}).values():
            ws.zero_()
        self._prepared = None


class DeepCoNNPPEngine(NarreEngine):
    """Native step for DeepCoNN++ (model_type 'deepconn++': TextCNN towers + `final` MLP + ID
    biases; csrc/deepconnpp_engine.hip, r4r_deepconnpp_step).  Shares NarreEngine's machinery: flat
    dense buffer aliased by the module's Parameters, the two ID bias vectors updated by a tagged
    sweep, token double-buffering.  The reference's `fm` module is constructed but unused in this
    mode (DeepCoNN.py:64-72) and is left alone."""
    NAMES = ['user_conv.convs.0.weight', 'user_conv.convs.0.bias', 'user_conv.fc.weight', 'user_conv.fc.bias',
             'item_conv.convs.0.weight', 'item_conv.convs.0.bias', 'item_conv.fc.weight', 'item_conv.fc.bias',
             'final.0.weight', 'final.0.bias', 'final.3.weight', 'final.3.bias', 'global_bias']
    ROW_NAMES = ['user_bias', 'item_bias']
    MODEL_TYPE = 'deepconn++'
    C = 'deepconnpp'
    DP_COLS = 1

    def __init__(self, model, dp=None, **kw):
        super().__init__(model, dp=dp, **kw)

    def _dp_doc_shape(self, data):
        return 1, int(data[3].shape[-1])

    def _dp_rows(self, data, n, B_pad, world, solo):
        """(uid, iid, d loss / d pred) per rating: one packing launch, ONE all_gather, one unpacking launch, the update."""
        R, T = self._dp_doc_shape(data)
        fields = [(None, 1, torch.int64), (None, 1, torch.int64), (None, 1, torch.float32)]
        if n > 0:
            f = self._fields(data)[0]
            off = self._ws_offset(n, R, T, 5)
            fields = [(f[2], 1, torch.int64), (f[3], 1, torch.int64),
                      (self._workspace(n, R, T)[off:off + n * 4].view(torch.float32), 1, torch.float32)]
        uid_all, iid_all, g_all = gather_entry_fields(self, fields, n, B_pad, world, None if solo else self.dp.all_gather)
        nb = max(n, 1)                                       # (the workspace of this rank's own shape holds the row tags)
        ws = self._workspace(nb, R, T)
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts])   # noqa: E731
        _lib.check(_lib.lib().r4r_deepconnpp_rows_apply(
            ptr(uid_all), ptr(iid_all), ptr(g_all), world * B_pad, p2(self.rows), p2(self.rows_m), p2(self.rows_v),
            self.n_users, self.n_items, ptr(ws), ws.numel(), nb, T, self.E, self.L, self.V, self.lr, self.betas[0],
            self.betas[1], self.eps, self.wd, int(self.step_count), _lib.current_stream()), 'r4r_deepconnpp_rows_apply')

    def _fields(self, data):
        n = data[5].numel()
        f = [data[3].reshape(n, -1), data[4].reshape(n, -1), data[5].reshape(-1), data[6].reshape(-1)]
        if f[0].shape != f[1].shape:
            raise RuntimeError('DeepCoNNPPEngine: user and item documents must share input_length')
        if not all(t.is_cuda and t.dtype == torch.int64 for t in f):
            raise RuntimeError('DeepCoNNPPEngine: batches must be int64 tensors on the ROCm device')
        return [t.contiguous() for t in f], n, 1, f[0].shape[1]

    def _span_doc(self, desc):
        if not desc.review or len(desc.doc_shape) != 1 or desc.batch_size > 32768:
            return None
        return 1, desc.doc_shape[0]

    def _span_call(self, lib, desc, first, steps, announce, done, pred, se, ws, R, T, draws, buf, ready, stream):
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts])   # noqa: E731
        return lib.r4r_deepconnpp_span(
            desc.words, int(first), int(steps), int(bool(announce)), ctypes.byref(desc.built), ctypes.byref(done),
            ptr(self.table), self.V, ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v),
            p2(self.rows), p2(self.rows_m), p2(self.rows_v), self.n_users, self.n_items, ptr(pred), ptr(se),
            ptr(self.sse), ptr(ws), ws.numel(), T, self.E, self.L, float(self.hp['dropout']), int(self.model.training),
            self.seed, self.offset, draws, 1.0 / float(desc.batch_size), self._algo_req, buf, ready, self.lr,
            self.betas[0], self.betas[1], self.eps, self.wd, self.step_count + 1, stream)

    def _ws_bytes(self, B, R, T):
        return _lib.lib().r4r_deepconnpp_ws_bytes(B, T, self.E, self.L, self.V, self.n_users, self.n_items)

    def _ws_offset(self, B, R, T, which):
        return _lib.lib().r4r_deepconnpp_ws_offset(B, T, self.E, self.L, self.V, self.n_users, self.n_items, which)

    def _draws(self, R):
        return 3 * self.L

    def _step(self, f, y, pred, se, ws, n, R, T, train_mode, inv_denom, adam_step, buf, ready, nxt):
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts])   # noqa: E731
        return _lib.lib().r4r_deepconnpp_step(
            ptr(self.table), self.V, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), ptr(y),
            ptr(self.flat_p), ptr(self.flat_g) if adam_step else None,
            ptr(self.flat_m) if (adam_step and self.dp is None) else None,     # data parallel: gradients only
            ptr(self.flat_v) if (adam_step and self.dp is None) else None, p2(self.rows),
            p2(self.rows_m) if (adam_step and self.dp is None) else None,
            p2(self.rows_v) if (adam_step and self.dp is None) else None,
            self.n_users, self.n_items, ptr(pred), ptr(se), ptr(self.sse) if adam_step else None,
            ptr(ws), ws.numel(), n, T, self.E, self.L, float(self.hp['dropout']), int(train_mode), self.seed,
            self.offset, float(inv_denom), self._algo_req, buf, ready,
            ptr(nxt[0]) if nxt else None, ptr(nxt[1]) if nxt else None,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, int(adam_step), _lib.current_stream())

    def grads(self, data):
        out = {k: slot_view(self.flat_g, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
               zip(self.NAMES, self.slots, self.offsets, self.sizes)}
        f, n, R, T = self._fields(data)
        off = self._ws_offset(n, R, T, 5)
        g = self._workspace(n, R, T)[off:off + n * 4].view(torch.float32)
        out['user_bias'] = torch.zeros_like(self.rows[0]).index_add_(0, f[2], g)
        out['item_bias'] = torch.zeros_like(self.rows[1]).index_add_(0, f[3], g)
        return out


class TransNetEngine(NarreEngine, _SweepSchedule):
    """Native step for TransNet / TransNet++ (csrc/transnet_engine.hip, r4r_transnet_step): three
    TextCNN towers, the source MLP, both factorisation machines, the three losses of main.py:35-53
    and their three disjoint parameter groups in ONE backward and one flat Adam (the three
    optimisers of utils.init_transnet_optim share lr, weight decay and step count; include/r4r.h
    has the argument for why the reference's three-pass step consumes exactly these gradients).
    ``sse`` holds [sum of source SE, sum of per-batch target MSE, sum of per-batch transform loss].
    Replaces the whole optimiser list of the host loop; single process only."""
    NAMES = ['source.user_conv.convs.0.weight', 'source.user_conv.convs.0.bias',
             'source.item_conv.convs.0.weight', 'source.item_conv.convs.0.bias',
             'target.conv.convs.0.weight', 'target.conv.convs.0.bias',
             'source.user_conv.fc.weight', 'source.user_conv.fc.bias',
             'source.item_conv.fc.weight', 'source.item_conv.fc.bias',
             'target.conv.fc.weight', 'target.conv.fc.bias',
             'source.project.0.weight', 'source.project.0.bias', 'source.project.2.weight', 'source.project.2.bias',
             'source_fm.V', 'source_fm.lin.weight', 'source_fm.lin.bias',
             'target.fm.V', 'target.fm.lin.weight', 'target.fm.lin.bias']
    MODEL_TYPE = ('transnet', 'transnet++')
    C = 'transnet'
    NTOWER = 3
    SSE_SLOTS = 3

    TEMPORAL_SWEEP = True        # train_step(..., defer_sweep=True) + flush(): the ID-vector sweep, temporally blocked

    def __init__(self, model, dp=None, **kw):
        if dp is not None and dp.on:
            self.dp = dp
        self.plus = int(model.hyper_params['model_type'] == 'transnet++')
        # visit period of the temporally blocked ID-vector sweep (include/r4r.h; 1 = the plain dense sweep)
        self.sweep_period = self.configured_period(model.hyper_params)
        self._tb_base, self._tb_period, self._defer_req, self._tb_now = 0, 1, False, (1, 0, 1, 1)
        if not self.plus:
            self.DP_COLS = 0                                 # plain TransNet: no ID rows to exchange
        self.ROW_NAMES = ['user_embedding.weight', 'item_embedding.weight'] if self.plus else []
        self._hp_counts = (int(model.hyper_params['total_users']) + 2, int(model.hyper_params['total_items']) + 2)
        super().__init__(model, dp=dp, **kw)
        self._sd_hooks = flush_before_state_dict(self, model)

    @staticmethod
    def _word_table(model):
        return model.target.word2vec.weight

    def _layout_extra(self):
        return (self.plus,)

    def _cardinalities(self):
        return (self.rows[0].shape[0], self.rows[1].shape[0]) if self.plus else self._hp_counts

    def _fields(self, data):
        n = data[5].numel()
        f = [data[3].reshape(n, -1), data[4].reshape(n, -1), data[0].reshape(n, -1), data[5].reshape(-1), data[6].reshape(-1)]
        if not (f[0].shape == f[1].shape == f[2].shape):
            raise RuntimeError('TransNetEngine: the three documents of a rating must share input_length')
        if not all(t.is_cuda and t.dtype == torch.int64 for t in f):
            raise RuntimeError('TransNetEngine: batches must be int64 tensors on the ROCm device')
        return [t.contiguous() for t in f], n, 1, f[0].shape[1]

    def _span_doc(self, desc):
        if not desc.review or len(desc.doc_shape) != 1 or desc.batch_size > 32768:
            return None
        return 1, desc.doc_shape[0]

    def _span_towers(self, slots):
        return (slots[3], slots[4], slots[0])                # user documents, item documents, the review being rated

    def _span_call(self, lib, desc, first, steps, announce, done, pred, se, ws, R, T, draws, buf, ready, stream):
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts]) if ts else None   # noqa: E731
        want = self.sweep_period if self.has_tables else 1   # (main.train: defer_sweep=True for this family)
        base, period = ctypes.c_int64(self._tb_base), ctypes.c_int(self._tb_period)
        rc = lib.r4r_transnet_span(
            desc.words, int(first), int(steps), int(bool(announce)), ctypes.byref(desc.built), ctypes.byref(done),
            ptr(self.table), self.V, ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v),
            p2(self.rows), p2(self.rows_m), p2(self.rows_v), self.n_users, self.n_items, ptr(pred), ptr(se),
            ptr(self.sse), ptr(ws), ws.numel(), T, self.E, self.L, self.plus, float(self.hp['dropout']),
            int(self.model.training), self.seed, self.offset, draws, 1.0 / float(desc.batch_size), self._algo_req, buf,
            ready, self._tb_period, want, ctypes.byref(base), ctypes.byref(period), self.lr, self.betas[0],
            self.betas[1], self.eps, self.wd, self.step_count + 1, stream)
        self._tb_base, self._tb_period = int(base.value), int(period.value)
        if int(done.value) and want > 1:
            self._tb_used = True
        return rc

    def _ws_bytes(self, B, R, T):
        return _lib.lib().r4r_transnet_ws_bytes(B, T, self.E, self.L, self.plus, self.V, self.n_users, self.n_items)

    def _ws_offset(self, B, R, T, which):
        return _lib.lib().r4r_transnet_ws_offset(B, T, self.E, self.L, self.plus, self.V, self.n_users, self.n_items, which)

    def _persist_bytes(self, B, R, T):
        return self._ws_offset(B, R, T, 4)

    def _draws(self, R):
        return 5 * self.L + 10

    # ---- the temporally blocked sweep (TransNet++): a step that was told `defer_sweep=True` visits the table chunks on
    # the schedule; rows a rating names catch up inside the step, everything else that reads the tables flushes first.
    @property
    def has_tables(self):
        return bool(self.plus)

    def _launch(self, data, y, train_mode, inv_denom, adam_step, next_data=None):
        if not adam_step:
            self.flush()                                     # a forward reads the tables: they catch up first
        return super()._launch(data, y, train_mode, inv_denom, adam_step, next_data)

    @step_or_nothing
    @torch.no_grad()
    def train_step(self, data, y, n_global=None, next_data=None, defer_sweep=False):
        """defer_sweep: TransNet++'s ID-vector sweep is temporally blocked (a chunk no rating names is visited every
        `sweep_period`-th step and takes its pending updates together; rows a rating names catch up on the way: same
        bits, a fraction of the traffic).  `flush()`, `state_dict()`, `predict()` and a step without defer_sweep bring
        the tables up to date; code that reads the embedding Parameters directly calls `flush()` before."""
        self._tb_now = self._schedule(bool(defer_sweep))
        before = self.step_count
        se = super().train_step(data, y, n_global, next_data)
        if self.step_count > before:
            self._scheduled(self._tb_now[2], self._tb_now[3], self.step_count)
        return se

    def flush(self, check=True, last_step=None):
        """Apply every pending ID-vector update (no-op when nothing is pending).  last_step: the last COMPLETED step
        (default: step_count)."""
        step = self._pending(last_step)
        if step is None:
            return
        B, R, T = self._ws_key
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts])   # noqa: E731
        _lib.check(_lib.lib().r4r_transnet_rows_flush(
            p2(self.rows), p2(self.rows_m), p2(self.rows_v), self.n_users, self.n_items, ptr(self._ws), self._ws.numel(),
            B, T, self.E, self.L, self.V, self._tb_period, self._tb_base,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step, _lib.current_stream()), 'r4r_transnet_rows_flush')
        self._tb_base = step
        if check:
            self.check_announcements()

    def check_announcements(self):
        """Raise if the scheduled sweep ever found more than 8 updates pending (a step outside the schedule without a
        flush: cannot happen through this class).  Reads one int from the device."""
        if not self.plus or self._ws is None:
            return
        B, R, T = self._ws_key
        off = self._ws_offset(B, R, T, 5)
        if int(self._ws[off:off + 4].view(torch.int32).item()):
            raise RuntimeError('TransNetEngine: the temporally blocked ID-vector sweep found more pending updates than it can apply')

    def moments(self):
        self.flush()
        return super().moments()

    def load_state_dict(self, sd):
        # (NarreEngine.load_state_dict validates before it copies; the schedule changes only once it has succeeded, so a
        # rejected checkpoint leaves the pending sweep updates in force)
        keep = (self._tb_base, self._tb_period)
        self._tb_base = self.step_count                      # moments() below must not flush into state about to be replaced
        try:
            super().load_state_dict(sd)
        except BaseException:
            self._tb_base, self._tb_period = keep
            raise
        self._tb_base, self._tb_period = self.step_count, 1  # the loaded tables are current through the loaded step

    def _step(self, f, y, pred, se, ws, n, R, T, train_mode, inv_denom, adam_step, buf, ready, nxt):
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts]) if ts else None   # noqa: E731
        period, base, sweep_all, _ = self._tb_now if adam_step else (1, 0, 1, 1)
        return _lib.lib().r4r_transnet_step(
            ptr(self.table), self.V, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), ptr(f[4]), ptr(y),
            ptr(self.flat_p), ptr(self.flat_g) if adam_step else None,
            ptr(self.flat_m) if (adam_step and self.dp is None) else None,     # data parallel: gradients only
            ptr(self.flat_v) if (adam_step and self.dp is None) else None, p2(self.rows),
            p2(self.rows_m) if adam_step else None,          # (data parallel: still read, for the catch-up of named rows)
            p2(self.rows_v) if adam_step else None,
            self.n_users, self.n_items, ptr(pred), ptr(se), ptr(self.sse) if adam_step else None,
            ptr(ws), ws.numel(), n, T, self.E, self.L, self.plus, float(self.hp['dropout']), int(train_mode), self.seed,
            self.offset, float(inv_denom), self._algo_req, buf, ready,
            ptr(nxt[0]) if nxt else None, ptr(nxt[1]) if nxt else None, ptr(nxt[2]) if nxt else None,
            period, base, sweep_all,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, int(adam_step), _lib.current_stream())

    DP_COLS = 10
    BLOCKS_DP = False            # (NarreEngine's own block form is not this family's: beyond BLOCKS_MAX_ENTRIES the generic payload)
    BLOCKS_MAX_ENTRIES = 2048    # r4r_transnet_rows_apply_blocks' limit on world * B_pad

    def _dp_rows(self, data, n, B_pad, world, solo):
        """Up to BLOCKS_MAX_ENTRIES gathered ratings: one packing launch, ONE all_gather, one update launch that reads the
        ranks' blocks directly (no tagging launch, no host-side slicing); beyond, the generic form."""
        if not (0 < world * B_pad <= self.BLOCKS_MAX_ENTRIES):
            return super()._dp_rows(data, n, B_pad, world, solo)
        lib = _lib.lib()
        R, T = self._dp_doc_shape(data)
        key = ('dp_blocks', B_pad, world)
        if key not in self._out:
            nb_bytes = lib.r4r_transnet_dp_block_bytes(B_pad)
            block = torch.zeros(nb_bytes, dtype=torch.uint8, device=self.dev)
            self._out[key] = (block, block if solo else torch.zeros(world * nb_bytes, dtype=torch.uint8, device=self.dev))
        block, blocks = self._out[key]
        nb = max(n, 1)                                       # (the workspace of this rank's own shape holds the sweep's state)
        ws = self._workspace(nb, R, T)
        uid = iid = None
        if n > 0:
            f = self._fields(data)[0]
            uid, iid = f[3], f[4]
        _lib.check(lib.r4r_transnet_dp_block(ptr(uid) if n else None, ptr(iid) if n else None, ptr(ws), ws.numel(), n, T,
                                             self.E, self.L, self.V, self.n_users, self.n_items, ptr(block), B_pad,
                                             _lib.current_stream()), 'r4r_transnet_dp_block')
        if not solo:
            self.dp.all_gather(blocks, block)
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts])   # noqa: E731
        period, base, sweep_all, _ = self._tb_now            # (the same on every rank: the loops run in lockstep)
        _lib.check(lib.r4r_transnet_rows_apply_blocks(
            ptr(blocks), world, B_pad, period, base, sweep_all, p2(self.rows), p2(self.rows_m), p2(self.rows_v),
            self.n_users, self.n_items, ptr(ws), ws.numel(), nb, T, self.E, self.L, self.V, self.lr, self.betas[0],
            self.betas[1], self.eps, self.wd, int(self.step_count), _lib.current_stream()), 'r4r_transnet_rows_apply_blocks')

    def _dp_payload(self, f, n, R, T, ids, vals):
        ws = self._workspace(n, R, T)
        ids[:n, 0], ids[:n, 1] = f[3], f[4]
        for t in range(2):
            off = self._ws_offset(n, R, T, 1 + t)
            vals[:n, 5 * t:5 * t + 5] = ws[off:off + n * 20].view(torch.float32).view(n, 5)

    def _dp_doc_shape(self, data):
        return 1, int(data[3].shape[-1])

    def _dp_cols(self, R):
        return 2, 10

    def _dp_apply(self, all_ids, all_vals, B_all, ws, nb, R, T):
        uid_all, iid_all = all_ids[:, 0].contiguous(), all_ids[:, 1].contiguous()
        gu_all, gi_all = all_vals[:, :5].contiguous(), all_vals[:, 5:].contiguous()
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts])   # noqa: E731
        period, base, sweep_all, _ = self._tb_now            # (the same on every rank: the loops run in lockstep)
        _lib.check(_lib.lib().r4r_transnet_rows_apply(
            ptr(uid_all), ptr(iid_all), ptr(gu_all), ptr(gi_all), period, base, sweep_all,
            B_all, p2(self.rows), p2(self.rows_m), p2(self.rows_v),
            self.n_users, self.n_items, ptr(ws), ws.numel(), nb, T, self.E, self.L, self.V, self.lr, self.betas[0],
            self.betas[1], self.eps, self.wd, int(self.step_count), _lib.current_stream()), 'r4r_transnet_rows_apply')

    def aux(self, data):
        """[B, 3] of the LAST step on `data`: target prediction, its squared error, ||s_ir - t_ir||^2."""
        f, n, R, T = self._fields(data)
        off = self._ws_offset(n, R, T, 3)
        return self._workspace(n, R, T)[off:off + n * 12].view(torch.float32).view(n, 3).clone()

    @torch.no_grad()
    def predict(self, data, y=None):
        """Eval-mode forward: (source prediction, its SE or None) -- what eval.py scores; `aux` has the rest."""
        return super().predict(data, y)

    def grads(self, data):
        out = {k: slot_view(self.flat_g, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
               zip(self.NAMES, self.slots, self.offsets, self.sizes)}
        if self.plus:
            f, n, R, T = self._fields(data)
            ws = self._workspace(n, R, T)
            for t, name in enumerate(self.ROW_NAMES):
                off = self._ws_offset(n, R, T, 1 + t)
                rows = ws[off:off + n * 5 * 4].view(torch.float32).view(n, 5)
                out[name] = torch.zeros_like(self.rows[t]).index_add_(0, f[3 + t], rows)
        return out


class IdNetEngine(_SweepSchedule, _Spans):
    """Native step for the ID-only recommenders with dense layers -- model_type 'MF' (MF.py:60-68) and the
    NeuMF family (NeuMF.py: GMF / MLP / NeuMF) -- csrc/idnet_engine.hip, r4r_idnet_step: forward, loss,
    backward and the dense Adam update of main.py:56-60,94-96 in 4 launches (5 for NeuMF); the dense
    gradient of an ID table is never materialised.  Same calling surface as the other engines
    (train_step / predict / sse / state_dict).  Under data parallelism (``dp``) the step computes gradients
    only, the dense gradient is all-reduced into the flat Adam and the ranks' compact ID rows are gathered
    into the same tagged sweeps on every rank (r4r_idnet_rows_apply): replicas stay bit-identical."""
    VARIANTS = {'MF': 0, 'GMF': 1, 'MLP': 2, 'NeuMF': 3}
    # reference parameter names of the 8 flat slots (None: the variant has no such layer)
    SLOT_NAMES = {
        'MF': ['projection.1.weight', 'projection.1.bias', 'projection.3.weight', 'projection.3.bias', 'final.V',
               'final.lin.weight', 'final.lin.bias', 'global_bias'],
        'GMF': [None, None, None, None, None, 'final.weight', 'final.bias', 'global_bias'],
        'MLP': ['project.1.weight', 'project.1.bias', 'project.3.weight', 'project.3.bias', None, 'final.weight',
                'final.bias', 'global_bias'],
        'NeuMF': ['project.1.weight', 'project.1.bias', 'project.3.weight', 'project.3.bias', None, 'final.weight',
                  'final.bias', 'global_bias'],
    }
    TABLE_NAMES = {
        'MF': ['user_embedding.weight', 'item_embedding.weight'],
        'GMF': ['user_embedding.weight', 'item_embedding.weight'],
        'MLP': ['user_embedding.weight', 'item_embedding.weight'],
        'NeuMF': ['gmf_user_embedding.weight', 'gmf_item_embedding.weight', 'mlp_user_embedding.weight',
                  'mlp_item_embedding.weight'],
    }
    MAX_L, MAX_TRAIN_BATCH = 64, 32768

    @staticmethod
    def kind_of(model):
        mt = model.hyper_params['model_type']
        return 'MF' if mt == 'MF' else type(model).__name__          # NeuMF.py's classes share one model_type

    def __init__(self, model, lr=0.002, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, seed=0x5EED5EED, rank=0,
                 dp=None):
        self.dp = dp if (dp is not None and dp.on) else None
        self.kind = self.kind_of(model)
        if self.kind not in self.VARIANTS:
            raise ValueError('IdNetEngine implements MF / GMF / MLP / NeuMF, got %r' % (self.kind,))
        self.variant = self.VARIANTS[self.kind]
        hp = model.hyper_params
        self.model, self.hp = model, hp
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.L = int(hp['latent_size'])
        params = dict(model.named_parameters())
        self.tables = [params[k] for k in self.TABLE_NAMES[self.kind]]
        self.biases = [params['user_bias'], params['item_bias']]
        if not all(p.is_cuda for p in self.tables):
            raise RuntimeError('IdNetEngine: move the model to a ROCm device first; the HIP path has no CPU fallback')
        if not all(p.is_contiguous() and p.dtype == torch.float32 for p in self.tables + self.biases):
            raise RuntimeError('IdNetEngine: fp32 contiguous ID tables / bias vectors only')
        self.dev = self.tables[0].device
        self.n_users, self.n_items = int(self.tables[0].shape[0]), int(self.tables[1].shape[0])
        lib = _lib.lib()
        n = lib.r4r_idnet_nparam()
        off, size, total = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)(), ctypes.c_int64()
        _lib.check(lib.r4r_idnet_layout(self.variant, self.L, off, size, ctypes.byref(total)), 'r4r_idnet_layout')
        self.names = self.SLOT_NAMES[self.kind]
        self.offsets, self.sizes, self.total = list(off), list(size), int(total.value)
        self.flat_p = torch.zeros(self.total, dtype=torch.float32, device=self.dev)
        self.slots = []
        for name, o, s in zip(self.names, self.offsets, self.sizes):
            if name is None:
                assert s == 0
                self.slots.append(None)
                continue
            p = params[name]
            assert p.numel() == s, (name, tuple(p.shape), s)
            view = self.flat_p[o:o + s].view(p.shape)
            view.copy_(p.data)
            p.data = view                       # the Parameter now aliases the flat buffer
            self.slots.append(p)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.rows = self.tables + [None] * (4 - len(self.tables)) + self.biases
        self.rows_m = [None if p is None else torch.zeros_like(p) for p in self.rows]
        self.rows_v = [None if p is None else torch.zeros_like(p) for p in self.rows]
        self.sse = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.step_count = 0
        self.seed = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        self._ws, self._ws_B, self._out = None, None, {}
        # visit period of the temporally blocked table sweeps (include/r4r.h; 1 = the plain dense sweeps)
        self.sweep_period = self.configured_period(hp)
        self._tb_base, self._tb_period = 0, 1
        self._sd_hooks = flush_before_state_dict(self, model)

    TEMPORAL_SWEEP = True        # train_step(..., defer_sweep=True) + flush(): the table sweeps, temporally blocked
    has_tables = True

    @staticmethod
    def _p6(tensors):
        return (ctypes.c_uint64 * 6)(*[0 if t is None else t.data_ptr() for t in tensors])

    def flush(self, check=True, last_step=None):
        """Apply every pending table update of the temporally blocked sweeps (no-op when nothing is pending).
        last_step: the last COMPLETED step (default: step_count)."""
        step = self._pending(last_step)
        if step is None:
            return
        _lib.check(_lib.lib().r4r_idnet_rows_flush(
            self.variant, self._p6(self.rows), self._p6(self.rows_m), self._p6(self.rows_v), self.n_users, self.n_items,
            ptr(self._ws), self._ws.numel(), self._ws_B, self.L, self._tb_period, self._tb_base,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step, _lib.current_stream()), 'r4r_idnet_rows_flush')
        self._tb_base = step
        if check:
            self.check_announcements()

    def check_announcements(self):
        """Raise if the scheduled sweeps ever found more than 8 updates pending (cannot happen through this class)."""
        if self._ws is None:
            return
        off = _lib.lib().r4r_idnet_ws_offset(self.variant, self._ws_B, self.L, self.n_users, self.n_items, 3)
        if int(self._ws[off:off + 4].view(torch.int32).item()):
            raise RuntimeError('IdNetEngine: the temporally blocked table sweeps found more pending updates than they can apply')

    def draws(self):
        return 2 * self.L * (2 if self.variant == 3 else 1) + (0 if self.variant == 1 else 2 * self.L)

    def _workspace(self, B):
        if self._ws_B != B:
            lib = _lib.lib()
            nb = lib.r4r_idnet_ws_bytes(self.variant, B, self.L, self.n_users, self.n_items)
            nxt = torch.zeros(max(nb, 256), dtype=torch.uint8, device=self.dev)
            if self._ws is not None:                         # the row tags head the buffer: shared state
                keep = lib.r4r_idnet_ws_offset(self.variant, B, self.L, self.n_users, self.n_items, 2)
                nxt[:keep].copy_(self._ws[:keep])
            self._ws, self._ws_B = nxt, B
        return self._ws

    def _launch(self, data, y, train_mode, inv_denom, adam_step, sched=(1, 0, 1, 1)):
        uid, iid = data[5].reshape(-1).contiguous(), data[6].reshape(-1).contiguous()
        if not (uid.is_cuda and uid.dtype == torch.int64 and iid.is_cuda and iid.dtype == torch.int64):
            raise RuntimeError('IdNetEngine: batches must be int64 tensors on the ROCm device')
        n = uid.numel()
        if not adam_step:
            self.flush()                                     # a forward reads the tables: they catch up first
        period, base, sweep_all, _ = sched
        if adam_step and n > self.MAX_TRAIN_BATCH:
            raise RuntimeError('IdNetEngine: training batch %d > %d' % (n, self.MAX_TRAIN_BATCH))
        if n not in self._out:
            self._out[n] = (torch.empty(n, dtype=torch.float32, device=self.dev),
                            torch.empty(n, dtype=torch.float32, device=self.dev))
        pred, se = self._out[n]
        ws = self._workspace(n)
        apply = bool(adam_step) and self.dp is None          # data parallel: gradients only
        rc = _lib.lib().r4r_idnet_step(
            self.variant, ptr(uid), ptr(iid), ptr(y), ptr(self.flat_p), ptr(self.flat_g) if adam_step else None,
            ptr(self.flat_m) if apply else None, ptr(self.flat_v) if apply else None,
            self._p6(self.rows), self._p6(self.rows_m) if adam_step else None,   # (data parallel: still read, for the catch-up)
            self._p6(self.rows_v) if adam_step else None, self.n_users, self.n_items, ptr(pred), ptr(se),
            ptr(self.sse) if adam_step else None, ptr(ws), ws.numel(), n, self.L, float(self.hp['dropout']),
            int(train_mode), self.seed, self.offset, float(inv_denom), period, base, sweep_all,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd, int(adam_step), _lib.current_stream())
        _lib.check(rc, 'r4r_idnet_step')
        self._step_launched = True
        if train_mode and float(self.hp['dropout']) > 0.0:
            self.offset += n * self.draws()
        return pred, se

    # ------------------------------------------------------------------ spans (K steps per host call)
    def _span_ok(self, desc):
        return not desc.review and desc.batch_size <= self.MAX_TRAIN_BATCH

    @torch.no_grad()
    def _span(self, desc, first, steps, announce):
        """`steps` consecutive train_step(defer_sweep=True) calls as ONE C call (r4r_idnet_span)."""
        lib, B = _lib.lib(), desc.batch_size
        if B not in self._out:
            self._out[B] = (torch.empty(B, dtype=torch.float32, device=self.dev),
                            torch.empty(B, dtype=torch.float32, device=self.dev))
        pred, se = self._out[B]
        ws = self._workspace(B)
        training, p_drop = self.model.training, float(self.hp['dropout'])
        draws = B * self.draws() if (training and p_drop > 0.0) else 0
        want = self.sweep_period
        base, period, done = ctypes.c_int64(self._tb_base), ctypes.c_int(self._tb_period), ctypes.c_int64(0)
        rc = lib.r4r_idnet_span(
            desc.words, int(first), int(steps), ctypes.byref(done), self.variant, ptr(self.flat_p), ptr(self.flat_g),
            ptr(self.flat_m), ptr(self.flat_v), self._p6(self.rows), self._p6(self.rows_m), self._p6(self.rows_v),
            self.n_users, self.n_items, ptr(pred), ptr(se), ptr(self.sse), ptr(ws), ws.numel(), self.L, p_drop,
            int(training), self.seed, self.offset, draws, 1.0 / float(B), self._tb_period, want, ctypes.byref(base),
            ctypes.byref(period), self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count + 1,
            _lib.current_stream())
        k = int(done.value)
        self.step_count += k
        self.offset += k * draws
        self._tb_base, self._tb_period = int(base.value), int(period.value)
        if k and want > 1:
            self._tb_used = True
        _lib.check(rc, 'r4r_idnet_span')

    @step_or_nothing
    @torch.no_grad()
    def train_step(self, data, y, n_global=None, next_data=None, defer_sweep=False):
        """One optimisation step.  Returns the per-example SE tensor (device, reused by the next call);
        the running sum is in ``self.sse``.  defer_sweep: the Adam sweeps over the ID tables are temporally blocked
        (MFEngine.train_step has the contract; next_data: accepted for the engines' common calling surface)."""
        if self.dp is not None:
            return self._train_step_dp(data, y, n_global, bool(defer_sweep))
        n = data[5].numel()
        if n == 0:
            return torch.empty(0, dtype=torch.float32, device=self.dev)
        self.step_count += 1
        sched = self._schedule(bool(defer_sweep))
        _, se = self._launch(data, y.reshape(-1).contiguous(), self.model.training,
                             1.0 / float(n_global if n_global is not None else n), self.step_count, sched)
        self._scheduled(sched[2], sched[3], self.step_count)
        return se

    @torch.no_grad()
    def _train_step_dp(self, data, y, n_global, defer=False):
        lib, dist = _lib.lib(), torch.distributed
        n, world, L = data[5].numel(), self.dp.world, self.L
        B_pad = int(self.hp.get('batch_size', 0))
        # the scheduled sweeps under data parallelism: the same (period, base, all) on every rank -- the loops run in
        # lockstep, and so do these decisions
        sched = self._schedule(bool(defer))
        if n_global is not None and n > B_pad:
            # (taking the size-agreement branch on THIS rank only would leave the others in a different
            # collective: a hang, not an error)
            raise RuntimeError("data parallel: this rank's shard has %d ratings but hyper_params['batch_size'] is %d; "
                               "pass n_global=None to let the ranks agree on the sizes first" % (n, B_pad))
        if n_global is None:                                 # agree on the sizes first: one more collective + a sync
            sizes = torch.tensor([n], dtype=torch.int64, device=self.dev)
            all_sizes = torch.empty(world, dtype=torch.int64, device=self.dev)
            dist.all_gather_into_tensor(all_sizes, sizes, group=self.dp.group)
            B_pad, n_global = int(all_sizes.max().item()), int(all_sizes.sum().item())
        self.step_count += 1
        se = torch.empty(0, dtype=torch.float32, device=self.dev)
        if n > 0:
            _, se = self._launch(data, y.reshape(-1).contiguous(), self.model.training, 1.0 / float(n_global),
                                 self.step_count, sched)
        else:
            self.flat_g.zero_()                              # an empty shard contributes a zero gradient
        self.dp.allreduce_flat(self.flat_g)                  # C1: the dense gradient
        one = ctypes.c_uint64 * 1
        _lib.check(lib.r4r_adam_multi(1, one(self.flat_p.data_ptr()), one(self.flat_g.data_ptr()),
                                      one(self.flat_m.data_ptr()), one(self.flat_v.data_ptr()),
                                      (ctypes.c_int64 * 1)(self.total), self.lr, self.betas[0], self.betas[1], self.eps,
                                      self.wd, int(self.step_count), None, _lib.current_stream()), 'r4r_adam_multi')
        # C2: per rating (uid, iid) and (d loss / d pred, the compact rows of every table); -1 ids pad ragged shards --
        # one packing launch, ONE all_gather, one unpacking launch (gather_entry_fields)
        ntab = len(self.tables)
        fields = [(data[5].reshape(-1).contiguous() if n else None, 1, torch.int64),
                  (data[6].reshape(-1).contiguous() if n else None, 1, torch.int64),
                  (self._ws_view(n, 1, 1) if n else None, 1, torch.float32)]
        fields += [(self._ws_view(n, 4 + t, L) if n else None, L, torch.float32) for t in range(ntab)]
        got = gather_entry_fields(self, fields, n, B_pad, world, self.dp.all_gather)
        uid_all, iid_all, g_all, rows = got[0], got[1], got[2], got[3:]
        p2 = lambda ts: (ctypes.c_uint64 * 2)(*[t.data_ptr() for t in ts] + [0] * (2 - len(ts)))
        nb = max(n, 1)                                       # (the workspace of this rank's own shape holds the row tags)
        ws = self._workspace(nb)
        _lib.check(lib.r4r_idnet_rows_apply(
            self.variant, ptr(uid_all), ptr(iid_all), ptr(g_all), p2(rows[0::2]), p2(rows[1::2]),
            sched[0], sched[1], sched[2], world * B_pad,
            self._p6(self.rows), self._p6(self.rows_m), self._p6(self.rows_v), self.n_users, self.n_items, ptr(ws),
            ws.numel(), nb, L, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, int(self.step_count),
            _lib.current_stream()), 'r4r_idnet_rows_apply')
        self._scheduled(sched[2], sched[3], self.step_count)
        return se

    @torch.no_grad()
    def predict(self, data, y=None):
        """Eval-mode forward (no dropout, no gradients).  Returns (pred, se or None)."""
        if data[5].numel() == 0:
            e = torch.empty(tuple(data[5].shape), dtype=torch.float32, device=self.dev)
            return e, (e.clone() if y is not None else None)
        if y is not None:
            y = y.reshape(-1).contiguous()
        pred, se = self._launch(data, y, False, 1.0, 0)
        shape = tuple(data[5].shape)
        return pred.view(shape), (se.view(shape) if y is not None else None)

    # ---- introspection for the parity tests
    def _ws_view(self, B, which, cols):
        lib = _lib.lib()
        at = lib.r4r_idnet_ws_offset(self.variant, B, self.L, self.n_users, self.n_items, which)
        return self._workspace(B)[at:at + B * cols * 4].view(torch.float32).view(B, cols)

    def dropout_multipliers(self, B):
        """[B, draws] multipliers of the last training step: per table pair the user row [L] and the item
        row [L], then (variants with a projection) its 2L inputs."""
        return self._ws_view(B, 0, self.draws()).clone()

    def grads(self, data):
        """Reference-named gradients of the LAST training step on `data`: the dense ones are views of the
        flat buffer, the ID tables / bias vectors are rebuilt from their compact rows."""
        out = {k: slot_view(self.flat_g, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
               zip(self.names, self.slots, self.offsets, self.sizes) if k is not None}
        uid, iid = data[5].reshape(-1), data[6].reshape(-1)
        B = uid.numel()
        for t, name in enumerate(self.TABLE_NAMES[self.kind]):
            rows = self._ws_view(B, 4 + t, self.L)
            out[name] = torch.zeros_like(self.tables[t]).index_add_(0, uid if t % 2 == 0 else iid, rows)
        g = self._ws_view(B, 1, 1)[:, 0]
        out['user_bias'] = torch.zeros_like(self.biases[0]).index_add_(0, uid, g)
        out['item_bias'] = torch.zeros_like(self.biases[1]).index_add_(0, iid, g)
        return out

    def moments(self):
        self.flush()
        m = {k: slot_view(self.flat_m, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
             zip(self.names, self.slots, self.offsets, self.sizes) if k is not None}
        v = {k: slot_view(self.flat_v, o, s, p.shape, getattr(self, 'E_model', 0), getattr(self, 'E', 0)) for k, p, o, s in
             zip(self.names, self.slots, self.offsets, self.sizes) if k is not None}
        for t, name in enumerate(self.TABLE_NAMES[self.kind] + ['user_bias', 'item_bias']):
            i = t if t < len(self.tables) else 4 + t - len(self.tables)
            m[name], v[name] = self.rows_m[i], self.rows_v[i]
        return m, v

    def state_dict(self):
        """Adam moments BY PARAMETER NAME (like the other engines): the flat buffer's slot alignment and padding are
        layout details of a build, not of a checkpoint."""
        m, v = self.moments()                                # (flushes the blocked sweeps)
        return {'exp_avg': {k: t.clone() for k, t in m.items()}, 'exp_avg_sq': {k: t.clone() for k, t in v.items()},
                'step': self.step_count, 'dropout_offset': self.offset, 'lr': self.lr, 'weight_decay': self.wd,
                'betas': self.betas, 'eps': self.eps}

    def load_state_dict(self, sd):
        if torch.is_tensor(sd.get('exp_avg')):               # a checkpoint of the flat form (written before round 5)
            if sd['exp_avg'].numel() != self.total:
                raise ValueError('IdNetEngine.load_state_dict: a flat-form checkpoint of %d moment elements for a %d-element '
                                 'layout (flat checkpoints do not survive layout changes; re-save by name)'
                                 % (sd['exp_avg'].numel(), self.total))
            rows = list(sd['rows_exp_avg']) + list(sd['rows_exp_avg_sq'])
            for mine, theirs in zip(self.rows_m + self.rows_v, rows):
                if mine is not None and (theirs is None or theirs.numel() != mine.numel()):
                    raise ValueError('IdNetEngine.load_state_dict: an ID-table moment of the checkpoint does not fit the model')
            scalars = (int(sd['step']), int(sd['dropout_offset']), float(sd['lr']), float(sd['weight_decay']),
                       tuple(sd['betas']), float(sd['eps']))
            self._tb_base = self.step_count                  # (validated: what was pending belongs to the state replaced now)
            self.flat_m.copy_(sd['exp_avg'].to(self.dev))
            self.flat_v.copy_(sd['exp_avg_sq'].to(self.dev))
            for mine, theirs in zip(self.rows_m + self.rows_v, rows):
                if mine is not None:
                    mine.copy_(theirs.to(self.dev))
        else:
            # moment views WITHOUT a flush (MFEngine.load_state_dict has the argument): the tables may already hold the
            # checkpoint's rows, and the old state's pending updates must not be applied to them
            keep = (self._tb_base, self._tb_period)
            self._tb_base = self.step_count
            try:
                m, v = self.moments()
                scalars = load_named_moments('IdNetEngine', m, v, sd)
            except BaseException:
                self._tb_base, self._tb_period = keep
                raise
        self.step_count, self.offset, self.lr, self.wd, self.betas, self.eps = scalars
        self._tb_base, self._tb_period = self.step_count, 1  # the loaded tables are current through the loaded step
        # row tags written by earlier steps of THIS process must not collide with resumed step numbers
        if self._ws is not None:
            self._ws.zero_()
