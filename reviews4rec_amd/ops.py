"""torch.autograd glue over the C ABI (include/r4r.h).

Every Function here is a thin shim: it allocates outputs with torch (device
memory is torch's job), passes raw device pointers to libr4r_hip.so on torch's
current stream, and wires forward/backward together.  No arithmetic on the hot
path is done by PyTorch ops and there is no CPU fallback: tensors must be
fp32/int64, contiguous and on a ROCm device, and a missing library raises.
"""
import torch
from torch.autograd import Function

from . import _lib
from ._lib import call, ptr

NUM_FILTERS_MAX = 112


def _f32(t, name):
    if not t.is_cuda:
        raise RuntimeError('reviews4rec_amd.ops: %s must live on a ROCm device (got %s); the HIP path '
                           'has no CPU fallback' % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError('reviews4rec_amd.ops: %s must be float32, got %s' % (name, t.dtype))
    return t.contiguous()


def _i64(t, name):
    if not t.is_cuda:
        raise RuntimeError('reviews4rec_amd.ops: %s must live on a ROCm device (got %s)' % (name, t.device))
    if t.dtype != torch.int64:
        raise RuntimeError('reviews4rec_amd.ops: %s must be int64 (LongTensor), got %s' % (name, t.dtype))
    return t.contiguous()


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------ TextCNN
_PADDED_TABLES = {}     # (data_ptr, shape, _version) of a frozen table -> (weakref to it, its zero-padded device copy)


def _padded_table(table):
    """word_embed_size % 4 != 0 (the reference takes any width, hyper_params.py:64, common_pytorch_models.py:15):
    the kernels read 16-byte aligned rows, so the frozen table gets a cached zero-padded copy; the conv weight is
    padded per call (100 x 3 x E floats) and the pad columns of its gradient -- exactly 0: g * 0 -- are dropped.
    The entry holds a weak reference to the tensor it was made from: a later table of the same shape that the
    allocator places at the same address (a second model built in one process) must not get this one's copy."""
    import weakref
    E = table.shape[1]
    E4 = (E + 3) // 4 * 4
    if E4 == E:
        return table
    key = (table.data_ptr(), tuple(table.shape), table._version)
    hit = _PADDED_TABLES.get(key)
    if hit is not None and hit[0]() is table:
        return hit[1]
    _PADDED_TABLES.clear()                                  # one table per model: no unbounded growth
    pt = torch.zeros((table.shape[0], E4), dtype=table.dtype, device=table.device)
    pt[:, :E].copy_(table.detach())
    _PADDED_TABLES[key] = (weakref.ref(table), pt)
    return pt


def textcnn_fwd_raw(idx, table, conv_w, conv_b):
    """idx [N,T] -> (pooled [N,F] fp32, argmax [N,F] int32).  No autograd."""
    idx, table = _i64(idx, 'idx'), _f32(table, 'table')
    conv_w, conv_b = _f32(conv_w, 'conv_w'), _f32(conv_b, 'conv_b')
    if table.shape[1] % 4:
        E0 = table.shape[1]
        table = _padded_table(table)
        pw = torch.zeros(tuple(conv_w.shape[:-1]) + (table.shape[1],), dtype=conv_w.dtype, device=conv_w.device)
        pw[..., :E0].copy_(conv_w)
        conv_w = pw
    N, T = idx.shape
    V, E = table.shape
    F = conv_b.numel()
    pooled = torch.empty((N, F), dtype=torch.float32, device=idx.device)
    argmax = torch.empty((N, F), dtype=torch.int32, device=idx.device)
    nb = _lib.lib().r4r_textcnn_ws_bytes(N, T, E, F, V)
    ws = _workspace(nb, idx.device)
    call('r4r_textcnn_fwd', ptr(table), V, ptr(idx), ptr(conv_w), ptr(conv_b), ptr(pooled), ptr(argmax),
         ptr(ws), ws.numel(), N, T, E, F)
    return pooled, argmax


def textcnn_wgrad_raw(idx, table, g_pooled, argmax, conv_w_shape):
    idx, table, g_pooled = _i64(idx, 'idx'), _f32(table, 'table'), _f32(g_pooled, 'g_pooled')
    E_model = table.shape[1]
    table = _padded_table(table)
    conv_w_shape = tuple(conv_w_shape[:-1]) + (table.shape[1],)
    N, T = idx.shape
    V, E = table.shape
    F = g_pooled.shape[1]
    d_w = torch.empty(conv_w_shape, dtype=torch.float32, device=idx.device)
    d_b = torch.empty((F,), dtype=torch.float32, device=idx.device)
    nb = _lib.lib().r4r_textcnn_ws_bytes(N, T, E, F, V)
    ws = _workspace(nb, idx.device)
    call('r4r_textcnn_wgrad', ptr(table), V, ptr(idx), ptr(g_pooled), ptr(argmax), ptr(d_w), ptr(d_b),
         ptr(ws), ws.numel(), N, T, E, F)
    if E != E_model:
        d_w = d_w[..., :E_model].contiguous()
    return d_w, d_b


class TextCNNPool(Function):
    """pooled[n,f] = max_p relu(conv(word2vec[idx[n]]))[f,p]  (frozen table: no grad to it)."""

    @staticmethod
    def forward(ctx, idx, table, conv_w, conv_b):
        pooled, argmax = textcnn_fwd_raw(idx, table, conv_w, conv_b)
        ctx.save_for_backward(idx, table, argmax)
        ctx.w_shape = tuple(conv_w.shape)
        ctx.mark_non_differentiable(argmax)
        return pooled, argmax

    @staticmethod
    def backward(ctx, g_pooled, _g_arg):
        idx, table, argmax = ctx.saved_tensors
        d_w, d_b = textcnn_wgrad_raw(idx, table, g_pooled, argmax, ctx.w_shape)
        return None, None, d_w, d_b


# ------------------------------------------------------------------- linear
class Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        x, w, b = _f32(x, 'x'), _f32(w, 'w'), _f32(b, 'b')
        lead = x.shape[:-1]
        n_in, n_out = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, n_in)
        y = torch.empty((x2.shape[0], n_out), dtype=torch.float32, device=x.device)
        call('r4r_linear_fwd', ptr(x2), ptr(w), ptr(b), ptr(y), x2.shape[0], n_in, n_out, int(relu))
        ctx.save_for_backward(x2, w, y)
        ctx.relu, ctx.lead = bool(relu), lead
        ctx.need_gx = ctx.needs_input_grad[0]
        return y.view(*lead, n_out)

    @staticmethod
    def backward(ctx, g_y):
        x2, w, y = ctx.saved_tensors
        n_in, n_out = x2.shape[1], w.shape[0]
        g_y = _f32(g_y, 'g_y').reshape(-1, n_out)
        g_x = torch.empty_like(x2) if ctx.need_gx else None
        g_w = torch.empty_like(w)
        g_b = torch.empty((n_out,), dtype=torch.float32, device=w.device)
        nb = _lib.lib().r4r_linear_bwd_ws_bytes(x2.shape[0], n_in, n_out)
        ws = _workspace(nb, w.device) if nb else None
        call('r4r_linear_bwd', ptr(x2), ptr(w), ptr(y), ptr(g_y), ptr(g_x), ptr(g_w), ptr(g_b),
             ptr(ws), nb, x2.shape[0], n_in, n_out, int(ctx.relu))
        return (g_x.view(*ctx.lead, n_in) if g_x is not None else None), g_w, g_b, None


def linear(x, weight, bias, relu=False):
    return Linear.apply(x, weight, bias, relu)


# ------------------------------------------------------------------ dropout
class DropoutState:
    """Philox stream position for the dropout kernels (per process / per rank).

    Eager mode keeps the position on the host (``offset``).  For hipGraph capture
    (``device_counter`` set, see graph.py) it lives in a device int64 tensor that the dropout
    launch itself advances, so every replay of a captured step draws fresh masks."""
    seed = 0x5EED5EED
    offset = 0
    record = None          # dict: site -> multiplier tensor, when a test wants the masks
    device_counter = None  # torch int64 [1] on the device, or None

    @classmethod
    def manual_seed(cls, seed, rank=0):
        cls.seed = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        cls.offset = 0
        if cls.device_counter is not None:
            cls.device_counter.zero_()

    @classmethod
    def draw(cls, x, y, mult, p):
        """Launch the dropout kernel on x -> (y, mult) and advance the stream position."""
        n = x.numel()
        if cls.device_counter is not None:
            call('r4r_dropout_fwd', ptr(x), ptr(y), ptr(mult), n, float(p), cls.seed, 0, ptr(cls.device_counter))
        else:
            call('r4r_dropout_fwd', ptr(x), ptr(y), ptr(mult), n, float(p), cls.seed, cls.offset, None)
            cls.offset += (n + 3) // 4


class Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, site):
        x = _f32(x, 'x')
        y = torch.empty_like(x)
        mult = torch.empty_like(x)
        DropoutState.draw(x, y, mult, p)
        if DropoutState.record is not None and site is not None:
            DropoutState.record[site] = mult
        ctx.save_for_backward(mult)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (mult,) = ctx.saved_tensors
        g_y = _f32(g_y, 'g_y')
        g_x = torch.empty_like(g_y)
        call('r4r_mul', ptr(g_y), ptr(mult), ptr(g_x), g_y.numel())
        return g_x, None, None


def dropout(x, p, training, site=None):
    """nn.Dropout semantics; identity in eval mode or at p == 0 (no kernel launched)."""
    if not training or p == 0.0:
        return x
    return Dropout.apply(x, p, site)


class Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a, 'a'), _f32(b, 'b')
        out = torch.empty_like(a)
        call('r4r_mul', ptr(a), ptr(b), ptr(out), a.numel())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32(g, 'g')
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        call('r4r_mul', ptr(g), ptr(b), ptr(ga), g.numel())
        call('r4r_mul', ptr(g), ptr(a), ptr(gb), g.numel())
        return ga, gb


class Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a, 'a'), _f32(b, 'b')
        out = torch.empty_like(a)
        call('r4r_add', ptr(a), ptr(b), ptr(out), a.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def mul(a, b):
    return Mul.apply(a, b)


def add(a, b):
    return Add.apply(a, b)


# ----------------------------------------------------------------------- FM
class FM(Function):
    """TorchFM without global bias: x [N,n] -> [N]."""

    @staticmethod
    def forward(ctx, x, V, lin_w, lin_b):
        x, V, lin_w, lin_b = _f32(x, 'x'), _f32(V, 'V'), _f32(lin_w, 'lin_w'), _f32(lin_b, 'lin_b')
        N, n = x.shape
        k = V.shape[1]
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
        call('r4r_fm_fwd', ptr(x), ptr(V), ptr(lin_w), ptr(lin_b), ptr(out), N, n, k)
        ctx.save_for_backward(x, V, lin_w)
        ctx.lw_shape = tuple(lin_w.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, V, lin_w = ctx.saved_tensors
        N, n = x.shape
        k = V.shape[1]
        g_out = _f32(g_out, 'g_out')
        g_x, g_V = torch.empty_like(x), torch.empty_like(V)
        g_lw = torch.empty(ctx.lw_shape, dtype=torch.float32, device=x.device)
        g_lb = torch.empty((1,), dtype=torch.float32, device=x.device)
        call('r4r_fm_bwd', ptr(x), ptr(V), ptr(lin_w), ptr(g_out), ptr(g_x), ptr(g_V), ptr(g_lw), ptr(g_lb), N, n, k)
        return g_x, g_V, g_lw, g_lb


def fm(x, V, lin_w, lin_b):
    return FM.apply(x, V, lin_w, lin_b)


# ------------------------------------------------------- embeddings / biases
class SparseGradCapture:
    """Data-parallel mode for ID tables / bias vectors (dist.DataParallel turns it on).

    When ``active`` the backward of an embedding / bias gather does NOT build the dense table
    gradient; it records the compact contribution ``(row ids, gradient rows)`` under the table's
    storage address and hands autograd an uninitialised placeholder.  DataParallel all-gathers the
    compact lists and rebuilds the identical dense gradient on every rank
    (``rebuild_dense``), so a dense table gradient never crosses xGMI."""
    active = False
    contributions = {}          # table.data_ptr() -> list[(idx [n] int64, g [n, D] fp32)]

    @classmethod
    def record(cls, table_ptr, idx, g):
        cls.contributions.setdefault(table_ptr, []).append((idx, g))

    @classmethod
    def clear(cls):
        cls.contributions = {}


def rebuild_dense(idx, g, R, D, out=None):
    """Dense [R, D] gradient from a compact list, deterministic summation order."""
    idx, g = _i64(idx, 'idx').reshape(-1), _f32(g, 'g').reshape(-1, D)
    if out is None:
        out = torch.empty((R, D), dtype=torch.float32, device=g.device)
    call('r4r_embed_scatter_add_ordered', ptr(g), ptr(idx), ptr(out), R, D, idx.numel())
    return out


class EmbedGather(Function):
    """rows = table[idx]; backward is the reference's DENSE table gradient."""

    @staticmethod
    def forward(ctx, table, idx):
        table, idx = _f32(table, 'table'), _i64(idx, 'idx')
        R, D = table.shape
        flat = idx.reshape(-1)
        out = torch.empty((flat.numel(), D), dtype=torch.float32, device=table.device)
        call('r4r_embed_gather', ptr(table), ptr(flat), ptr(out), R, D, flat.numel())
        ctx.save_for_backward(flat)
        ctx.shape = (R, D)
        ctx.table_ptr = table.data_ptr()
        return out.view(*idx.shape, D)

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        R, D = ctx.shape
        g = _f32(g, 'g').reshape(-1, D)
        g_table = torch.empty((R, D), dtype=torch.float32, device=g.device)
        if SparseGradCapture.active:
            SparseGradCapture.record(ctx.table_ptr, flat, g)   # dense gradient rebuilt after the exchange
            return g_table, None
        call('r4r_embed_scatter_add', ptr(g), ptr(flat), ptr(g_table), R, D, flat.numel())
        return g_table, None


def embed(table, idx):
    return EmbedGather.apply(table, idx)


class RowDot(Function):
    @staticmethod
    def forward(ctx, a, c):
        a, c = _f32(a, 'a'), _f32(c, 'c')
        N, D = a.shape
        out = torch.empty((N,), dtype=torch.float32, device=a.device)
        call('r4r_rowdot_fwd', ptr(a), ptr(c), ptr(out), N, D)
        ctx.save_for_backward(a, c)
        return out

    @staticmethod
    def backward(ctx, g):
        a, c = ctx.saved_tensors
        g = _f32(g, 'g')
        ga, gc = torch.empty_like(a), torch.empty_like(c)
        call('r4r_rowdot_bwd', ptr(a), ptr(c), ptr(g), ptr(ga), ptr(gc), a.shape[0], a.shape[1])
        return ga, gc


def rowdot(a, c):
    return RowDot.apply(a, c)


class BiasHead(Function):
    """out = (r +) user_bias[uid] + item_bias[iid] + global_bias; with user_bias None: r + global_bias."""

    @staticmethod
    def forward(ctx, r, user_bias, item_bias, global_bias, uid, iid):
        global_bias = _f32(global_bias, 'global_bias')
        ctx.has_ids = user_bias is not None
        if ctx.has_ids:
            user_bias, item_bias = _f32(user_bias, 'user_bias'), _f32(item_bias, 'item_bias')
            uid, iid = _i64(uid, 'uid').reshape(-1), _i64(iid, 'iid').reshape(-1)
            N = uid.numel()
            ctx.save_for_backward(uid, iid)
            ctx.sizes = (user_bias.numel(), item_bias.numel())
            ctx.ptrs = (user_bias.data_ptr(), item_bias.data_ptr())
        if r is not None:
            r = _f32(r, 'r')
            N = r.numel()
        out = torch.empty((N,), dtype=torch.float32, device=global_bias.device)
        call('r4r_bias_head_fwd', ptr(r), ptr(user_bias), ptr(item_bias), ptr(global_bias), ptr(uid), ptr(iid),
             ptr(out), N)
        ctx.has_r = r is not None
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32(g, 'g')
        g_gb = torch.empty((1,), dtype=torch.float32, device=g.device)
        g_ub = g_ib = uid = iid = None
        RU = RI = 0
        if ctx.has_ids:
            uid, iid = ctx.saved_tensors
            RU, RI = ctx.sizes
            g_ub = torch.empty((RU,), dtype=torch.float32, device=g.device)
            g_ib = torch.empty((RI,), dtype=torch.float32, device=g.device)
            if SparseGradCapture.active:
                SparseGradCapture.record(ctx.ptrs[0], uid, g.reshape(-1, 1))
                SparseGradCapture.record(ctx.ptrs[1], iid, g.reshape(-1, 1))
                call('r4r_bias_head_bwd', ptr(g), None, None, None, None, ptr(g_gb), 0, 0, g.numel())
                return (g if ctx.has_r else None), g_ub, g_ib, g_gb, None, None
        call('r4r_bias_head_bwd', ptr(g), ptr(uid), ptr(iid), ptr(g_ub), ptr(g_ib), ptr(g_gb), RU, RI, g.numel())
        return (g if ctx.has_r else None), g_ub, g_ib, g_gb, None, None


def bias_head(r, user_bias, item_bias, global_bias, uid, iid):
    return BiasHead.apply(r, user_bias, item_bias, global_bias, uid, iid)


# ---------------------------------------------------------- NARRE attention
class NarreAttention(Function):
    @staticmethod
    def forward(ctx, x, other, W0, b0, w3, b3, p, training, site):
        x, other = _f32(x, 'x'), _f32(other, 'other')
        W0, b0, w3, b3 = _f32(W0, 'W0'), _f32(b0, 'b0'), _f32(w3, 'w3'), _f32(b3, 'b3')
        N, R, L = x.shape
        mult = None
        if training and p > 0.0:
            # the scorer's dropout multiplier, drawn by the same Philox kernel as every other site
            ones = torch.ones((N, R, L), dtype=torch.float32, device=x.device)
            mult = torch.empty_like(ones)
            scratch = torch.empty_like(ones)
            DropoutState.draw(ones, scratch, mult, p)
            if DropoutState.record is not None and site is not None:
                DropoutState.record[site] = mult
        out = torch.empty((N, L), dtype=torch.float32, device=x.device)
        h = torch.empty((N, R, L), dtype=torch.float32, device=x.device)
        a = torch.empty((N, R), dtype=torch.float32, device=x.device)
        call('r4r_narre_attn_fwd', ptr(x), ptr(other), ptr(W0), ptr(b0), ptr(w3), ptr(b3), ptr(mult),
             ptr(out), ptr(h), ptr(a), N, R, L)
        ctx.save_for_backward(x, other, W0, w3, h, a)
        ctx.mult = mult
        ctx.w3_shape = tuple(w3.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, other, W0, w3, h, a = ctx.saved_tensors
        N, R, L = x.shape
        g_out = _f32(g_out, 'g_out')
        g_x, g_other = torch.empty_like(x), torch.empty_like(other)
        g_W0 = torch.empty_like(W0)
        g_b0 = torch.empty((L,), dtype=torch.float32, device=x.device)
        g_w3 = torch.empty(ctx.w3_shape, dtype=torch.float32, device=x.device)
        g_b3 = torch.empty((1,), dtype=torch.float32, device=x.device)
        ws = _workspace(_lib.lib().r4r_narre_attn_ws_bytes(N, R, L), x.device)
        call('r4r_narre_attn_bwd', ptr(x), ptr(other), ptr(W0), ptr(w3), ptr(ctx.mult), ptr(h), ptr(a), ptr(g_out),
             ptr(g_x), ptr(g_other), ptr(g_W0), ptr(g_b0), ptr(g_w3), ptr(g_b3), ptr(ws), ws.numel(), N, R, L)
        return g_x, g_other, g_W0, g_b0, g_w3, g_b3, None, None, None


def narre_attention(x, other, W0, b0, w3, b3, p=0.0, training=False, site=None):
    return NarreAttention.apply(x, other, W0, b0, w3, b3, p, training, site)


class TransformLoss(Function):
    """mean_n ||a_n - b_n||^2 (TransNet's source/target representation loss)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a, 'a'), _f32(b, 'b')
        out = torch.empty((1,), dtype=torch.float32, device=a.device)
        call('r4r_sqdist_mean_fwd', ptr(a), ptr(b), ptr(out), a.shape[0], a.shape[1])
        ctx.save_for_backward(a, b)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32(g, 'g').reshape(1)
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        call('r4r_sqdist_mean_bwd', ptr(a), ptr(b), ptr(g), ptr(ga), ptr(gb), a.shape[0], a.shape[1])
        return ga, gb


def transform_loss(a, b):
    return TransformLoss.apply(a, b)


# --------------------------------------------------------------------- loss
def mse_fwd_bwd(out, y, denom, want_grad=True):
    """(per-example SE, d mean(SE)/d out) in one pass over the predictions."""
    out, y = _f32(out, 'out').reshape(-1), _f32(y, 'y').reshape(-1)
    se = torch.empty_like(out)
    g = torch.empty_like(out) if want_grad else None
    call('r4r_mse_fwd_bwd', ptr(out), ptr(y), ptr(se), ptr(g), out.numel(), float(denom))
    return se, g
