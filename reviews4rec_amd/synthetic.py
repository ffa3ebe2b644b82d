"""Seeded, Amazon-shaped synthetic interactions (there is no network for the real
datasets).  Shapes and distributions follow SURVEY.md 8(d): Zipf user / item /
token ids, log-normal document fill zero-padded to T (the reference pads with id
0, data.py:198-199), 5-star-skewed ratings, and a word table with the
distribution utils.xavier_init leaves behind (fact 2)."""
import math
import os

import numpy as np

SEED = 20200725

CONFIGS = {
    # BASELINE.json configs; cardinalities are the public Amazon 5-core figures
    'cfg1_bias_only_musical': dict(model_type='bias_only', total_users=1429, total_items=900, n_train=8208,
                                   latent_size=10, word_embed_size=64, input_length=1000, vocab=0),
    'cfg2_mfdot_electronics': dict(model_type='MF_dot', total_users=192403, total_items=63001, n_train=1351350,
                                   latent_size=64, word_embed_size=64, input_length=1000, vocab=0),
    'cfg3_deepconn_electronics_e300': dict(model_type='deepconn', total_users=192403, total_items=63001,
                                           n_train=1351350, latent_size=10, word_embed_size=300,
                                           input_length=1000, vocab=50002),
    'cfg4_narre_kindle': dict(model_type='NARRE', total_users=68223, total_items=61934, n_train=786095,
                              latent_size=10, word_embed_size=64, input_length=1000, vocab=50002,
                              narre_num_reviews=10, narre_num_words=100),
    'cfg5_transnetpp_synthetic': dict(model_type='transnet++', total_users=10_000_000, total_items=1_000_000,
                                      n_train=10_000_000, latent_size=10, word_embed_size=64,
                                      input_length=1000, vocab=1_000_000),
}


def hyper_params_for(name, **over):
    hp = dict(dataset=name, k_core=5, percent_reviews_to_keep=100, weight_decay=1e-6, lr=0.002, epochs=1,
              batch_size=128, dropout=0.6, narre_num_reviews=10, narre_num_words=100)
    hp.update(CONFIGS[name])
    hp.update(over)
    return hp


def _zipf_sampler(n, alpha, rng, ranked=False):
    """Sampler of ids in [0, n) with P(rank k) ~ 1/(k+1)^alpha, ranks randomly permuted."""
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), alpha)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = rng.permutation(n)
    if ranked:                                               # ids in frequency order (id 0 the most frequent)
        perm = np.arange(n)
    return lambda size: perm[np.searchsorted(cdf, rng.random(size), side='right').clip(max=n - 1)]


def word_table(V, E, rng=None):
    rng = rng or np.random.default_rng(SEED)
    bound = math.sqrt(6.0 / (V + E))
    return ((rng.random((V, E), dtype=np.float32) * 2 - 1) * bound).astype(np.float32)


class Generator:
    """doc_fill 'lognormal' (SURVEY 8d default: documents zero-padded to T) or 'full' (every document
    fills all T positions: no pad run); token_dist 'zipf' (default) or 'uniform' (every word equally
    likely: the most distinct tokens a batch can hold).  The non-default settings are stress points
    for the project-then-gather convolution, whose work follows the batch's DISTINCT tokens."""

    def __init__(self, hp, seed=SEED, doc_fill='lognormal', token_dist='zipf'):
        self.hp = hp
        self.rng = np.random.default_rng(seed)
        self.users = _zipf_sampler(hp['total_users'], 1.1, self.rng)
        self.items = _zipf_sampler(hp['total_items'], 1.1, self.rng)
        if doc_fill not in ('lognormal', 'full') or token_dist not in ('zipf', 'uniform'):
            raise ValueError('doc_fill / token_dist: %r / %r' % (doc_fill, token_dist))
        self.doc_fill = doc_fill
        V = hp.get('vocab', 0)
        self.tokens = None
        if V:
            if token_dist == 'uniform':
                self.tokens = lambda size: self.rng.integers(1, V, size=size)
            else:
                tok = _zipf_sampler(V - 1, 1.0, self.rng, ranked=os.environ.get('R4R_SYNTH_TOKEN_ORDER') == 'rank')
                self.tokens = lambda size: tok(size) + 1      # id 0 is the pad / UNK row

    def _docs(self, lead, T):
        tok = self.tokens(lead + (T,))
        if self.doc_fill == 'full':
            return tok
        fill = np.minimum(T, self.rng.lognormal(math.log(0.4 * T), 1.0, size=lead)).astype(np.int64).clip(min=1)
        return np.where(np.arange(T) < fill[..., None], tok, 0)

    def batch(self, B):
        """-> ([this, users_who, items_reviewed, user_reviews, item_reviews, user_id, item_id], y) numpy
        int64 / float32, the layout data_fast.py:101-109 yields."""
        hp = self.hp
        mt = hp['model_type']
        T = hp['input_length']
        uid, iid = self.users((B,)), self.items((B,))
        y = self.rng.choice(np.array([1, 2, 3, 4, 5], dtype=np.float32), size=B, p=[.05, .05, .10, .22, .58])
        if self.tokens is None:
            z = np.zeros((B, 1), dtype=np.int64)
            return [z, z, z, z, z, uid.astype(np.int64), iid.astype(np.int64)], y.astype(np.float32)
        if mt == 'NARRE':
            R, W = hp['narre_num_reviews'], hp['narre_num_words']
            ur, ir = self._docs((B, R), W), self._docs((B, R), W)
            n_u = self.rng.integers(1, R + 1, size=B)
            n_i = self.rng.integers(1, R + 1, size=B)
            ur[np.arange(R)[None, :] >= n_u[:, None]] = 0       # padded (all-zero) reviews
            ir[np.arange(R)[None, :] >= n_i[:, None]] = 0
            this = np.zeros((B, 1), dtype=np.int64)
        else:
            ur, ir = self._docs((B,), T), self._docs((B,), T)
            this = self._docs((B,), T) if mt.startswith('transnet') else np.zeros((B, 1), dtype=np.int64)
        who = self.users((B, 10))
        rev = self.items((B, 10))
        pad_u = self.rng.random((B, 10)) < 0.3
        pad_i = self.rng.random((B, 10)) < 0.3
        who = np.where(pad_u, hp['total_users'] + 1, who)       # the +1 sentinel (data.py:275-276)
        rev = np.where(pad_i, hp['total_items'] + 1, rev)
        data = [this, who, rev, ur, ir, uid, iid]
        return [np.ascontiguousarray(d.astype(np.int64)) for d in data], y.astype(np.float32)
