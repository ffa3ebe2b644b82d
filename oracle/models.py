"""Functional plain-PyTorch restatement of the reference models (CPU oracle).

TEST INFRASTRUCTURE -- never imported by the product package.

Every model is a pure function ``f(params, data, ...)`` over a ``dict`` of
tensors whose keys are the reference modules' ``state_dict()`` names, so a
reference checkpoint is directly usable as ``params``.  Stock ATen ops are used
on purpose (``F.conv2d``, ``F.embedding``, ``F.max_pool1d`` ...) because that
is where the reference's arithmetic lives.

Dropout: ``train=False`` -> identity (reference ``model.eval()``).  With
``train=True`` each dropout site looks up an injected multiplier tensor in
``masks`` (0 or 1/(1-p) per element, what ``nn.Dropout`` multiplies by); a site
without an injected mask draws from torch's global RNG like the reference.

Reference map (file:line under /root/reference):
  textcnn_forward   pytorch_models/common_pytorch_models.py:22-39
  fm_forward        pytorch_models/common_pytorch_models.py:49-57
  mf_forward        pytorch_models/MF.py:39-68
  neumf_forward     pytorch_models/NeuMF.py:25-38 (GMF), 60-74 (MLP), 120-143 (NeuMF)
  neumf_init        pytorch_models/NeuMF.py:100-118 (NeuMF.init from pre-trained GMF / MLP)
  deepconn_forward  pytorch_models/DeepCoNN.py:37-72
  narre_forward     pytorch_models/NARRE.py:53-124
  transnet_forward  pytorch_models/TransNet.py:25-37,55-61,83-122
  mse_loss          loss.py:7-11
  init_params       the constructors above + utils.py:65-68 (xavier_init)
"""
import math

import torch
import torch.nn.functional as F

NUM_FILTERS = 100      # common_pytorch_models.py:11
WINDOW = 3             # common_pytorch_models.py:7
FM_K = 8               # DeepCoNN.py:32, TransNet.py:50,77,79
TRANSNET_ID_DIM = 5    # TransNet.py:75-76
NUM_NEIGHBOURS = 10    # data.py:274-279


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _dropout(x, site, p, train, masks):
    if not train or p == 0.0:
        return x
    if masks is not None and site in masks:
        return x * masks[site]
    return F.dropout(x, p, True)


def _linear(params, prefix, x):
    return F.linear(x, params[prefix + '.weight'], params[prefix + '.bias'])


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def textcnn_forward(params, prefix, x, p=0.0, train=False, masks=None):
    """x: [N, T, E] fp32 word vectors -> [N, L].

    conv(1->100, window 3 x E, pad 2 rows) -> relu -> max over all T+2
    positions -> Linear(100 -> L) -> dropout.
    """
    w = params[prefix + '.convs.0.weight']            # [100, 1, 3, E]
    b = params[prefix + '.convs.0.bias']
    y = F.conv2d(x.unsqueeze(1), w, b, padding=(WINDOW - 1, 0))   # [N, 100, T+2, 1]
    y = F.relu(y).squeeze(-1)
    pooled = F.max_pool1d(y, y.size(2)).squeeze(-1)   # [N, 100]
    z = _linear(params, prefix + '.fc', pooled)
    return _dropout(z, prefix + '.dropout', p, train, masks)


def fm_forward(params, prefix, x):
    """Factorisation machine without global bias: [N, n] -> [N]."""
    V = params[prefix + '.V']
    s1 = (x @ V).pow(2).sum(1)
    s2 = (x.pow(2) @ V.pow(2)).sum(1)
    lin = _linear(params, prefix + '.lin', x)[:, 0]
    return 0.5 * (s1 - s2) + lin


def _flatten_negs(user_id):
    """The reference folds an optional negatives dim [B, 6, ...] into the batch."""
    if user_id.dim() > 1:
        return tuple(user_id.shape), user_id.shape[0] * user_id.shape[1]
    return (user_id.shape[0],), user_id.shape[0]


# --------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------
def mf_forward(params, data, model_type, p=0.0, train=False, masks=None):
    user_id, item_id = data[5], data[6]
    shape = user_id.shape
    u, i = user_id.reshape(-1), item_id.reshape(-1)
    base = params['user_bias'][u] + params['item_bias'][i] + params['global_bias']
    if model_type == 'bias_only':
        return base.view(shape)
    ue = _dropout(params['user_embedding.weight'][u], 'dropout.user', p, train, masks)
    ie = _dropout(params['item_embedding.weight'][i], 'dropout.item', p, train, masks)
    if model_type == 'MF_dot':
        return (base + (ue * ie).sum(-1)).view(shape)
    # 'MF': MLP on [u ; i] concatenated with the GMF vector, then an FM.
    h = _dropout(torch.cat([ue, ie], -1), 'projection.0', p, train, masks)
    h = F.relu(_linear(params, 'projection.1', h))
    h = _linear(params, 'projection.3', h)
    x = torch.cat([h, ue * ie], -1)
    return (base + fm_forward(params, 'final', x)).view(shape)


def neumf_forward(params, data, stage, p=0.0, train=False, masks=None):
    """stage 'GMF' / 'MLP' (the two pre-training models) or 'NeuMF' (their fusion)."""
    user_id, item_id = data[5], data[6]
    shape = user_id.shape
    u, i = user_id.reshape(-1), item_id.reshape(-1)
    base = params['user_bias'][u] + params['item_bias'][i] + params['global_bias']

    def emb(name, site):
        idx = u if 'user' in name else i
        return _dropout(params[name + '.weight'][idx], site, p, train, masks)

    def project(x):
        h = _dropout(x, 'project.0', p, train, masks)
        h = F.relu(_linear(params, 'project.1', h))
        return _linear(params, 'project.3', h)

    if stage == 'GMF':
        joint = emb('user_embedding', 'dropout.user') * emb('item_embedding', 'dropout.item')
    elif stage == 'MLP':
        joint = project(torch.cat([emb('user_embedding', 'dropout.user'), emb('item_embedding', 'dropout.item')], -1))
    elif stage == 'NeuMF':
        gmf = emb('gmf_user_embedding', 'dropout.gmf_user') * emb('gmf_item_embedding', 'dropout.gmf_item')
        mlp = project(torch.cat([emb('mlp_user_embedding', 'dropout.mlp_user'),
                                 emb('mlp_item_embedding', 'dropout.mlp_item')], -1))
        joint = torch.cat([gmf, mlp], -1)
    else:
        raise ValueError('unknown NeuMF stage %r' % (stage,))
    rating = _linear(params, 'final', joint)[:, 0]
    return (base + rating).view(shape)


def neumf_init(neumf, gmf, mlp):
    """NeuMF.init: returns the NeuMF parameter dict initialised from pre-trained GMF / MLP dicts
    (`neumf` supplies the entries init leaves alone: global_bias)."""
    P = dict(neumf)
    P['gmf_user_embedding.weight'] = gmf['user_embedding.weight'].clone()
    P['gmf_item_embedding.weight'] = gmf['item_embedding.weight'].clone()
    P['mlp_user_embedding.weight'] = mlp['user_embedding.weight'].clone()
    P['mlp_item_embedding.weight'] = mlp['item_embedding.weight'].clone()
    for k in ('project.1.weight', 'project.1.bias', 'project.3.weight', 'project.3.bias'):
        P[k] = mlp[k].clone()
    P['final.weight'] = torch.cat([gmf['final.weight'], mlp['final.weight']], -1)
    P['final.bias'] = 0.5 * (gmf['final.bias'] + mlp['final.bias'])
    P['user_bias'] = 0.5 * (gmf['user_bias'] + mlp['user_bias'])
    P['item_bias'] = 0.5 * (gmf['item_bias'] + mlp['item_bias'])
    return P


def deepconn_forward(params, data, model_type, p=0.0, train=False, masks=None):
    user_reviews, item_reviews, user_id, item_id = data[3], data[4], data[5], data[6]
    shape, n = _flatten_negs(user_id)
    table = params['word2vec.weight']
    ud = F.embedding(user_reviews.reshape(n, -1), table)
    idoc = F.embedding(item_reviews.reshape(n, -1), table)
    zu = textcnn_forward(params, 'user_conv', ud, p, train, masks)
    zi = textcnn_forward(params, 'item_conv', idoc, p, train, masks)
    x = torch.cat([zu, zi], -1)
    if model_type == 'deepconn':
        return (params['global_bias'] + fm_forward(params, 'fm', x)).view(shape)
    # deepconn++ : MLP head + biases
    h = F.relu(_linear(params, 'final.0', x))
    h = _dropout(h, 'final.2', p, train, masks)
    r = _linear(params, 'final.3', h)[:, 0]
    u, i = user_id.reshape(-1), item_id.reshape(-1)
    return (r + params['user_bias'][u] + params['item_bias'][i] + params['global_bias']).view(shape)


def _narre_attention(params, scorer, x, other, p, train, masks):
    h = F.relu(_linear(params, scorer + '.0', torch.cat([x, other], -1)))
    h = _dropout(h, scorer + '.2', p, train, masks)
    a = F.softmax(_linear(params, scorer + '.3', h)[:, :, 0], dim=-1)   # [N, R]; pads not masked
    return (a.unsqueeze(-1) * x).sum(1)


def narre_forward(params, data, p=0.0, train=False, masks=None):
    users_who, items_rev, user_reviews, item_reviews, user_id, item_id = data[1:7]
    shape, n = _flatten_negs(user_id)
    users_who = users_who.reshape(n, -1)
    items_rev = items_rev.reshape(n, -1)
    R_u, W_u = user_reviews.shape[-2], user_reviews.shape[-1]
    R_i, W_i = item_reviews.shape[-2], item_reviews.shape[-1]
    u, i = user_id.reshape(-1), item_id.reshape(-1)
    table = params['word2vec.weight']
    ud = F.embedding(user_reviews.reshape(n * R_u, W_u), table)
    idoc = F.embedding(item_reviews.reshape(n * R_i, W_i), table)
    zu = textcnn_forward(params, 'user_conv', ud, p, train, masks).view(n, R_u, -1)
    zi = textcnn_forward(params, 'item_conv', idoc, p, train, masks).view(n, R_i, -1)
    ue_t, ie_t = params['user_embedding.weight'], params['item_embedding.weight']
    au = _narre_attention(params, 'attention_scorer_user', zu, ie_t[items_rev], p, train, masks)
    ai = _narre_attention(params, 'attention_scorer_item', zi, ue_t[users_who], p, train, masks)
    au = au + _dropout(ue_t[u], 'dropout.user', p, train, masks)
    ai = ai + _dropout(ie_t[i], 'dropout.item', p, train, masks)
    h = _dropout(au * ai, 'final.0', p, train, masks)
    h = F.relu(_linear(params, 'final.1', h))
    r = _linear(params, 'final.3', h)[:, 0]
    return (r + params['user_bias'][u] + params['item_bias'][i] + params['global_bias']).view(shape)


def transnet_forward(params, data, model_type, p=0.0, train=False, masks=None):
    """Returns [source_pred, target_pred, mean ||source.ir - target.ir||^2]."""
    this_reviews, user_reviews, item_reviews, user_id, item_id = data[0], data[3], data[4], data[5], data[6]
    shape, n = _flatten_negs(user_id)
    table = params['target.word2vec.weight']
    ud = F.embedding(user_reviews.reshape(n, -1), table)
    idoc = F.embedding(item_reviews.reshape(n, -1), table)
    td = F.embedding(this_reviews.reshape(n, -1), table)
    # Source: two towers -> MLP -> dropout
    zu = textcnn_forward(params, 'source.user_conv', ud, p, train, masks)
    zi = textcnn_forward(params, 'source.item_conv', idoc, p, train, masks)
    h = F.relu(_linear(params, 'source.project.0', torch.cat([zu, zi], -1)))
    src_ir = _dropout(_linear(params, 'source.project.2', h), 'source.dropout', p, train, masks)
    if model_type == 'transnet++':
        u, i = user_id.reshape(-1), item_id.reshape(-1)
        ue = _dropout(params['user_embedding.weight'][u], 'dropout.user', p, train, masks)
        ie = _dropout(params['item_embedding.weight'][i], 'dropout.item', p, train, masks)
        final = torch.cat([ue, ie, src_ir], -1)
    else:
        final = src_ir
    source_out = fm_forward(params, 'source_fm', final)
    # Target: tower on the actual review -> dropout -> FM
    tgt_ir = _dropout(textcnn_forward(params, 'target.conv', td, p, train, masks),
                      'target.dropout', p, train, masks)
    target_out = fm_forward(params, 'target.fm', tgt_ir)
    transform = (src_ir - tgt_ir).pow(2).sum(-1).mean()
    return [source_out.view(shape), target_out.view(shape), transform]


def model_forward(params, data, hyper_params, train=False, masks=None):
    mt = hyper_params['model_type']
    p = float(hyper_params.get('dropout', 0.0))
    if mt in ('bias_only', 'MF_dot', 'MF'):
        return mf_forward(params, data, mt, p, train, masks)
    if mt == 'NeuMF':
        return neumf_forward(params, data, hyper_params.get('neumf_stage', 'NeuMF'), p, train, masks)
    if mt in ('deepconn', 'deepconn++'):
        return deepconn_forward(params, data, mt, p, train, masks)
    if mt == 'NARRE':
        return narre_forward(params, data, p, train, masks)
    if mt in ('transnet', 'transnet++'):
        return transnet_forward(params, data, mt, p, train, masks)
    raise ValueError('unknown model_type %r' % (mt,))


def mse_loss(output, y, return_mean=True):
    se = (output - y).pow(2)
    return se.mean() if return_mean else se


# --------------------------------------------------------------------------
# parameter construction (ctor defaults + xavier_init)
# --------------------------------------------------------------------------
def _xavier(shape, gen):
    # torch.nn.init.xavier_uniform_: fan_in = size(1)*receptive, fan_out = size(0)*receptive
    rec = 1
    for s in shape[2:]:
        rec *= s
    bound = math.sqrt(6.0 / (shape[1] * rec + shape[0] * rec))
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def _bias(n, fan_in, gen):
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(n, generator=gen) * 2 - 1) * bound


def _textcnn_params(params, prefix, E, L, gen):
    params[prefix + '.convs.0.weight'] = _xavier((NUM_FILTERS, 1, WINDOW, E), gen)
    params[prefix + '.convs.0.bias'] = _bias(NUM_FILTERS, WINDOW * E, gen)
    params[prefix + '.fc.weight'] = _xavier((L, NUM_FILTERS), gen)
    params[prefix + '.fc.bias'] = _bias(L, NUM_FILTERS, gen)


def _linear_params(params, prefix, n_in, n_out, gen):
    params[prefix + '.weight'] = _xavier((n_out, n_in), gen)
    params[prefix + '.bias'] = _bias(n_out, n_in, gen)


def _fm_params(params, prefix, n, k, gen):
    params[prefix + '.V'] = _xavier((n, k), gen)
    _linear_params(params, prefix + '.lin', n, 1, gen)


def init_params(hyper_params, vocab_size=None, seed=0):
    """Random parameters with the distributions the reference ends up with
    after ``Model(hp)`` + ``xavier_init`` (every dim>1 tensor xavier-uniform,
    INCLUDING the word table and FM ``V`` -- SURVEY.md fact 2; biases at their
    constructor defaults).  Same keys/shapes as the reference state_dict; the
    values are this oracle's own stream, not torch's module-init stream.
    """
    gen = torch.Generator().manual_seed(seed)
    mt = hyper_params['model_type']
    L = hyper_params['latent_size']
    E = hyper_params.get('word_embed_size', 64)
    U, I = hyper_params['total_users'], hyper_params['total_items']
    P = {}
    if mt in ('bias_only', 'MF_dot', 'MF'):
        P['user_bias'] = torch.full((U + 1,), 0.1)
        P['item_bias'] = torch.full((I + 1,), 0.1)
        P['global_bias'] = torch.full((1,), 4.0)
        if mt != 'bias_only':
            P['user_embedding.weight'] = _xavier((U + 1, L), gen)
            P['item_embedding.weight'] = _xavier((I + 1, L), gen)
        if mt == 'MF':
            _linear_params(P, 'projection.1', 2 * L, L, gen)
            _linear_params(P, 'projection.3', L, L, gen)
            _fm_params(P, 'final', 2 * L, L, gen)
        return P
    if mt == 'NeuMF':
        stage = hyper_params.get('neumf_stage', 'NeuMF')
        P['user_bias'] = torch.full((U + 1,), 0.1)
        P['item_bias'] = torch.full((I + 1,), 0.1)
        P['global_bias'] = torch.full((1,), 4.0)
        tables = ['user_embedding', 'item_embedding'] if stage != 'NeuMF' else \
            ['gmf_user_embedding', 'gmf_item_embedding', 'mlp_user_embedding', 'mlp_item_embedding']
        for t in tables:
            P[t + '.weight'] = _xavier(((U if 'user' in t else I) + 1, L), gen)
        if stage != 'GMF':
            _linear_params(P, 'project.1', 2 * L, L, gen)
            _linear_params(P, 'project.3', L, L, gen)
        _linear_params(P, 'final', 2 * L if stage == 'NeuMF' else L, 1, gen)
        return P
    V = vocab_size
    if mt in ('deepconn', 'deepconn++'):
        P['word2vec.weight'] = _xavier((V, E), gen)
        _textcnn_params(P, 'user_conv', E, L, gen)
        _textcnn_params(P, 'item_conv', E, L, gen)
        _linear_params(P, 'final.0', 2 * L, L, gen)
        _linear_params(P, 'final.3', L, 1, gen)
        P['user_bias'] = torch.full((U + 2,), 0.1)
        P['item_bias'] = torch.full((I + 2,), 0.1)
        P['global_bias'] = torch.full((1,), 4.0)
        _fm_params(P, 'fm', 2 * L, FM_K, gen)
        return P
    if mt == 'NARRE':
        P['word2vec.weight'] = _xavier((V, E), gen)
        P['user_embedding.weight'] = _xavier((U + 2, L), gen)
        P['item_embedding.weight'] = _xavier((I + 2, L), gen)
        _textcnn_params(P, 'user_conv', E, L, gen)
        _textcnn_params(P, 'item_conv', E, L, gen)
        for s in ('attention_scorer_user', 'attention_scorer_item'):
            _linear_params(P, s + '.0', 2 * L, L, gen)
            _linear_params(P, s + '.3', L, 1, gen)
        _linear_params(P, 'final.1', L, L, gen)
        _linear_params(P, 'final.3', L, 1, gen)
        P['user_bias'] = torch.full((U + 2,), 0.1)
        P['item_bias'] = torch.full((I + 2,), 0.1)
        P['global_bias'] = torch.full((1,), 4.0)
        return P
    if mt in ('transnet', 'transnet++'):
        P['target.word2vec.weight'] = _xavier((V, E), gen)
        _textcnn_params(P, 'target.conv', E, L, gen)
        _fm_params(P, 'target.fm', L, FM_K, gen)
        _textcnn_params(P, 'source.user_conv', E, L, gen)
        _textcnn_params(P, 'source.item_conv', E, L, gen)
        _linear_params(P, 'source.project.0', 2 * L, L, gen)
        _linear_params(P, 'source.project.2', L, L, gen)
        if mt == 'transnet++':
            P['user_embedding.weight'] = _xavier((U + 2, TRANSNET_ID_DIM), gen)
            P['item_embedding.weight'] = _xavier((I + 2, TRANSNET_ID_DIM), gen)
            _fm_params(P, 'source_fm', 2 * TRANSNET_ID_DIM + L, FM_K, gen)
        else:
            _fm_params(P, 'source_fm', L, FM_K, gen)
        return P
    raise ValueError('unknown model_type %r' % (mt,))


def trainable_names(params):
    """Names Adam would see: everything except the frozen word table
    (``Embedding.from_pretrained`` freezes it -- SURVEY.md fact 3)."""
    return [k for k in params if not k.endswith('word2vec.weight')]
