"""Pickle-backed data loader (counterpart of the reference's data.py): same constructor, same
``iter`` / ``iter_simple`` / ``iter_review`` / ``iter_negs`` / ``__len__`` surface, same files read by
``load_data`` (data.py:449-485), same batches -- pinned bit for bit by fixtures the reference's own
loader produced (tests/golden/tiny, tests/test_data_loader.py).

The reference rebuilds every batch from Python lists of token lists (data.py:250-333: per rating a
``remove_overlap`` walk, per document a ``pad_and_join`` loop) -- ~1 s per DeepCoNN batch, which is why
it ships a second loader (data_fast.py) that stores all N x 3 x T padded documents up front (32 GB for
Electronics).  Here the reviews are flattened ONCE into token pools:

    tok      int32 [total tokens]      every review of every user (item), in (owner, review) order
    rev_off  int64 [reviews + 1]       token range of a review
    first    int64 [owners + 1]        review range of a user (item)
    nb       int64 [reviews]           the item (user) each review is about: u_to_i_map / i_to_u_map
                                       of data.py:36-63, flattened onto the same review index

and a batch is a pure index computation on them: a document is the owner's contiguous token range
with ONE hole (the review of the rating itself, data.py:212-236), cut / zero-padded to input_length
(data.py:181-210); NARRE's [R, W] layout and the two 10-wide neighbour lists are the same ranges seen
per review.  On a ROCm device the pools are resident in HBM (Electronics: ~1 GB instead of 32 GB of
host RAM) and ONE kernel launch per batch (r4r_batch_build, csrc/batcher.hip) writes the five index
tensors: no host work, no H2D copy per batch.  On the CPU the same ranges are evaluated with numpy --
that is the host-side reader (the reference's loader is host code too), used where no GPU exists
and by tools/make_quick_data.py; the models themselves refuse CPU tensors.
"""
from itertools import chain

import numpy as np
import torch

from .utils import load_obj

NEIGHBOURS = 10                                            # data.py:275-279: hard-coded list width


class _Pool:
    """Token pool of one side (users or items): see the module docstring."""

    def __init__(self, reviews, owners):
        self.owners = owners
        counts = np.zeros(owners, dtype=np.int64)
        for o, revs in reviews.items():
            counts[int(o)] = len(revs)
        self.first = np.zeros(owners + 1, dtype=np.int64)
        np.cumsum(counts, out=self.first[1:])
        ordered = [reviews.get(o, ()) for o in range(owners)]
        lens = np.fromiter((len(r) for revs in ordered for r in revs), dtype=np.int64, count=int(self.first[-1]))
        self.rev_off = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=self.rev_off[1:])
        self.tok = np.fromiter(chain.from_iterable(chain.from_iterable(ordered)), dtype=np.int32,
                               count=int(self.rev_off[-1]))
        self.nb = np.zeros(len(lens), dtype=np.int64)      # data.py:41,56: entries nobody fills stay 0
        self._dev = None

    def device(self, dev):
        if self._dev is None or self._dev[0] != dev:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            # one spare token so an empty pool still has an address
            tok = np.concatenate([self.tok, np.zeros(1, np.int32)])
            self._dev = (dev, t(tok), t(self.rev_off), t(self.first), t(self.nb))
        return self._dev[1:]


class _Held:
    """test_reviews[u][i] -> token list (preprocess_random_split.py:226-238) as a pool keyed by (u, i)."""

    def __init__(self, test_reviews):
        keys, revs = [], []
        for u, per_item in (test_reviews or {}).items():
            for i, rev in per_item.items():
                keys.append((int(u), int(i)))
                revs.append(rev)
        self.index = {k: n for n, k in enumerate(keys)}
        lens = np.fromiter((len(r) for r in revs), dtype=np.int64, count=len(revs))
        self.rev_off = np.zeros(len(revs) + 1, dtype=np.int64)
        np.cumsum(lens, out=self.rev_off[1:])
        self.tok = np.fromiter(chain.from_iterable(revs), dtype=np.int32, count=int(self.rev_off[-1]))
        self._dev = None

    def device(self, dev):
        if self._dev is None or self._dev[0] != dev:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            self._dev = (dev, t(np.concatenate([self.tok, np.zeros(1, np.int32)])), t(self.rev_off))
        return self._dev[1:]


class ReviewStore:
    """Everything the three loaders of one dataset share (data.py:469-483 passes the same dicts to all)."""

    def __init__(self, user_reviews, item_reviews, this_index_user_item, test_reviews, total_users=None,
                 total_items=None):
        nu = max([int(u) for u in user_reviews] + [-1]) + 1
        ni = max([int(i) for i in item_reviews] + [-1]) + 1
        self.users = _Pool(user_reviews, max(nu, int(total_users or 0)))
        self.items = _Pool(item_reviews, max(ni, int(total_items or 0)))
        self.held = _Held(test_reviews)
        self.has_held = test_reviews is not None
        # calculate_reviewed_map (data.py:36-63): review k of user u is about item nb[first[u] + k]
        self.pair = {}                                     # (u, i) -> (index among u's reviews, among i's)
        for u, per_item in (this_index_user_item or {}).items():
            for i, (ku, ki) in per_item.items():
                u_, i_ = int(u), int(i)
                self.pair[(u_, i_)] = (int(ku), int(ki))
                if u_ < self.users.owners and ku < self.users.first[u_ + 1] - self.users.first[u_]:
                    self.users.nb[self.users.first[u_] + ku] = i_
                if i_ < self.items.owners and ki < self.items.first[i_ + 1] - self.items.first[i_]:
                    self.items.nb[self.items.first[i_] + ki] = u_


# ----------------------------------------------------------------------------- host-side evaluation
def _rows(pool_first, ids, skip):
    rb = pool_first[ids]
    return rb, pool_first[ids + 1], skip


def _np_docs(tok, rev_off, rb, re, skip, T):
    """[N, T]: tokens of reviews [rb, re) minus review rb + skip (skip < 0: none), zero padded / cut."""
    start, end = rev_off[rb], rev_off[np.maximum(re, rb)]
    has = (skip >= 0) & (skip < re - rb)        # an index past the owner's list removes nothing (data.py:223-235)
    hb = np.where(has, rev_off[np.where(has, rb + skip, rb)], end)
    hl = np.where(has, rev_off[np.where(has, rb + skip + 1, rb)] - hb, 0)
    src = start[:, None] + np.arange(T, dtype=np.int64)[None, :]
    src = np.where(src >= hb[:, None], src + hl[:, None], src)
    ok = src < end[:, None]
    padded = np.concatenate([tok, np.zeros(1, tok.dtype)])
    return np.where(ok, padded[np.where(ok, src, len(tok))], 0).astype(np.int64)


def _np_reviews(tok, rev_off, rb, re, skip, R, W):
    """[N, R, W] (NARRE, data.py:144-172): review slot r = the r-th remaining review, each cut / padded to W."""
    r = np.arange(R, dtype=np.int64)[None, :]
    skip = np.where(skip >= re - rb, -1, skip)
    k = r + ((skip[:, None] >= 0) & (r >= skip[:, None]))
    rev = rb[:, None] + k
    okr = rev < re[:, None]
    rev = np.where(okr, rev, 0)
    if len(rev_off) < 2:
        return np.zeros((len(rb), R, W), dtype=np.int64)
    rs, rl = rev_off[rev], rev_off[rev + 1] - rev_off[rev]
    w = np.arange(W, dtype=np.int64)[None, None, :]
    ok = okr[:, :, None] & (w < rl[:, :, None])
    padded = np.concatenate([tok, np.zeros(1, tok.dtype)])
    return np.where(ok, padded[np.where(ok, rs[:, :, None] + w, len(tok))], 0).astype(np.int64)


def _np_neighbours(nb, rb, re, skip, pad):
    r = np.arange(NEIGHBOURS, dtype=np.int64)[None, :]
    skip = np.where(skip >= re - rb, -1, skip)
    k = r + ((skip[:, None] >= 0) & (r >= skip[:, None]))
    at = rb[:, None] + k
    ok = at < re[:, None]
    padded = np.concatenate([nb, np.zeros(1, nb.dtype)])
    return np.where(ok, padded[np.where(ok, at, len(nb))], pad).astype(np.int64)


class SpanDescriptor:
    """Host-side view of one loader for r4r_*_span: `words` (ctypes uint64 [R4R_LOADER_WORDS], layout in
    include/r4r.h), `built` (the ring's state, a c_int64 the C side advances), `full_batches`, `batch_size`, and the
    tensors the words point at (kept alive here)."""

    WORDS = 28

    def __init__(self, loader):
        import ctypes
        hp, st, dev = loader.hyper_params, loader.store, loader.device
        B, N = int(hp['batch_size']), len(loader.data)
        self.batch_size, self.n_ratings, self.full_batches = B, N, N // B
        self.review = loader.iter != loader.iter_simple
        w = [0] * self.WORDS
        if self.review:
            narre = hp['model_type'] in ['NARRE']
            T = int(hp['input_length'])
            R, W = (int(hp['narre_num_reviews']), int(hp['narre_num_words'])) if narre else (0, 0)
            doc = R * W if narre else T
            d = loader._device_split()
            self.group = G = max(1, loader.SPAN_RATINGS // B)
            stride = G * B * (3 * doc + 2 * NEIGHBOURS)
            self.ring = torch.empty(2 * stride, dtype=torch.int64, device=dev)
            pools = list(st.users.device(dev)) + list(st.items.device(dev)) + list(st.held.device(dev))
            self._keep = (pools, d, self.ring)
            w[0:10] = [t.data_ptr() for t in pools]
            w[10:16] = [d[k].data_ptr() for k in ('u', 'i', 'ku', 'ki', 'held', 'y')]
            w[16:22] = [int(loader.this_index_user_item is not None), T, R, W, int(hp['total_users']) + 1,
                        int(hp['total_items']) + 1]
            w[22:24] = [self.ring.data_ptr(), stride]
            self.doc_shape = (R, W) if narre else (T,)
        else:
            d = loader._device_split_simple()
            self.group, self.ring, self._keep = 1, None, (d,)
            w[10], w[11], w[15] = d['u'].data_ptr(), d['i'].data_ptr(), d['y'].data_ptr()
            self.doc_shape = None
        w[24:27] = [N, self.group, B]
        self.words = (ctypes.c_uint64 * self.WORDS)(*w)
        self.built = ctypes.c_int64(-1)

    @classmethod
    def resident(cls, batches, review=None):
        """A descriptor over batches that already exist on the device -- ``[(data, y), ...]`` in the 7-slot layout, all
        of one size -- cycled through: batch b of a span is ``batches[b % len(batches)]``, nothing is built.  (What a
        caller holding its epoch in HBM -- bench.py's pool -- gives the span entry points.)"""
        import ctypes
        self = cls.__new__(cls)
        data0, y0 = batches[0]
        B = int(y0.numel())
        table, keep = [], []
        for data, y in batches:
            if int(y.numel()) != B:
                raise ValueError('SpanDescriptor.resident: batches of %d and %d ratings' % (B, int(y.numel())))
            row = []
            for t in list(data) + [y]:
                if t is None:
                    row.append(0)
                    continue
                t = t.contiguous()
                keep.append(t)
                row.append(t.data_ptr())
            table += row
        self.batch_size, self.group, self.ring = B, len(batches), None
        self.n_ratings = (1 << 40) * B
        self.full_batches = 1 << 40
        self.review = (data0[3] is not None) if review is None else bool(review)   # (ids-only batches may carry dummy documents)
        self.doc_shape = None
        if self.review:
            self.doc_shape = tuple(data0[3].shape[-2:]) if data0[3].dim() - data0[5].dim() == 2 else tuple(data0[3].shape[-1:])
        self._table = (ctypes.c_uint64 * len(table))(*table)
        self._keep = keep
        w = [0] * cls.WORDS
        w[24:27] = [self.n_ratings, self.group, B]
        w[27] = ctypes.addressof(self._table)
        self.words = (ctypes.c_uint64 * cls.WORDS)(*w)
        self.built = ctypes.c_int64(-1)
        return self


class DataLoader():
    """data.DataLoader (data.py:11-447).  ``data`` is the rating list ``[[user, item, rating], ...]``;
    the dict arguments are the unpickled files of a dataset directory.  A loader built with
    ``train_loader=`` shares that loader's counts, maps and token pools (data.py:29-32)."""

    def __init__(self, hyper_params, data, user_reviews, item_reviews, negs, this_index_user_item=None,
                 test_reviews=None, train_loader=None, device=None):
        self.data = np.array(data)
        self.hyper_params = hyper_params
        self.user_reviews = user_reviews
        self.item_reviews = item_reviews
        self.this_index_user_item = this_index_user_item
        self.test_reviews = test_reviews
        self.negs = negs
        self.device = torch.device(device) if device is not None else (
            torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu'))
        if train_loader is None:
            self.count_train_counts()
            self.store = ReviewStore(user_reviews or {}, item_reviews or {}, this_index_user_item, test_reviews,
                                     hyper_params.get('total_users'), hyper_params.get('total_items'))
        else:
            self.user_count, self.item_count = train_loader.user_count, train_loader.item_count
            self.store = train_loader.store
            if test_reviews is not None and not self.store.has_held:
                self.store.held, self.store.has_held = _Held(test_reviews), True
        self.takes_batch = True                              # the iterators accept batch= (eval.py)
        if hyper_params['model_type'] in ['bias_only', 'MF', 'MF_dot', 'NeuMF']:
            self.iter = self.iter_simple
        else:
            self.iter = self.iter_review
        # the split as arrays: ids, ratings and -- what remove_overlap looks up per rating
        # (data.py:217-218, 243-245) -- the indices of the rating's own review
        n = len(self.data)
        self._u = self.data[:, 0].astype(np.int64) if n else np.zeros(0, np.int64)
        self._i = self.data[:, 1].astype(np.int64) if n else np.zeros(0, np.int64)
        self._y = self.data[:, 2].astype(np.float32) if n else np.zeros(0, np.float32)
        self._prepared = None
        self._dev_split = None
        self._dev_simple = None
        self._dev_negs = {}
        self._span = None

    # ------------------------------------------------------------------ the reference's small methods
    @property
    def u_to_i_map(self):
        s = self.store.users
        return {u: s.nb[s.first[u]:s.first[u + 1]].tolist() for u in range(s.owners)}

    @property
    def i_to_u_map(self):
        s = self.store.items
        return {i: s.nb[s.first[i]:s.first[i + 1]].tolist() for i in range(s.owners)}

    def __len__(self):
        bsz = int(self.hyper_params['batch_size'])
        return len(self.data) // bsz + int(len(self.data) % bsz > 0)

    def batch_sizes(self):
        """Rows of every batch iter() will yield (data parallelism sums them across ranks once per epoch)."""
        bsz, n = int(self.hyper_params['batch_size']), len(self.data)
        return [min(bsz, n - s) for s in range(0, n, bsz)]

    def count_train_counts(self):
        self.user_count, self.item_count = {}, {}
        if len(self.data) == 0:
            return
        for col, out in ((0, self.user_count), (1, self.item_count)):
            ids, first, cnt = np.unique(self.data[:, col], return_index=True, return_counts=True)
            for j in np.argsort(first, kind='stable'):     # first-appearance order, like the reference's loop
                out[ids[j]] = int(cnt[j])

    def get_count_user(self, user):
        return self.user_count.get(user, 0)

    def get_count_item(self, item):
        return self.item_count.get(item, 0)

    # ------------------------------------------------------------------ per-split index arrays
    def _split_arrays(self):
        """(ku, ki, held): per rating the index of its own review among the user's / the item's reviews
        (train loader: this_index_user_item given, data.py:217) or -1, and the held-out review's index
        (loaders with test_reviews, data.py:243-244) or -1 (data.py:245: ``[0]``, an all-padding document)."""
        if self._prepared is None:
            n = len(self.data)
            ku = np.full(n, -1, np.int64)
            ki = np.full(n, -1, np.int64)
            held = np.full(n, -1, np.int64)
            if self.this_index_user_item is not None:
                pair = self.store.pair
                for r in range(n):
                    ku[r], ki[r] = pair[(int(self._u[r]), int(self._i[r]))]     # KeyError like data.py:217
            elif self.test_reviews is not None:
                index = self.store.held.index
                for r in range(n):
                    held[r] = index[(int(self._u[r]), int(self._i[r]))]         # KeyError like data.py:244
            self._prepared = (ku, ki, held)
        return self._prepared

    def _review_fields_host(self, u, i, ku, ki, held, own_user_pos=None):
        """The five review slots of data.py:282-300 for ratings (u, i) -> numpy arrays."""
        hp, st = self.hyper_params, self.store
        narre = hp['model_type'] in ['NARRE']
        T = int(hp['input_length'])
        R, W = int(hp['narre_num_reviews']), int(hp['narre_num_words'])
        us, it = st.users, st.items
        urb, ure, _ = _rows(us.first, u, ku)
        irb, ire, _ = _rows(it.first, i, ki)
        # neighbour lists follow the rating's OWN pair even where the item documents do not (iter_negs
        # passes the positive item to remove_overlap for all six candidates, data.py:398)
        nb_i = i if own_user_pos is None else own_user_pos
        nrb, nre, _ = _rows(it.first, nb_i, ki)
        who = _np_neighbours(it.nb, nrb, nre, ki, int(hp['total_users']) + 1)
        what = _np_neighbours(us.nb, urb, ure, ku, int(hp['total_items']) + 1)
        # the rating's own review: train -> review ku of the user; held-out -> the held pool; else [0]
        if self.this_index_user_item is not None:
            tb = urb + ku
            te = tb + 1
            ttok, toff = us.tok, us.rev_off
        else:
            tb = np.where(held >= 0, held, 0)
            te = np.where(held >= 0, held + 1, 0)
            ttok, toff = st.held.tok, st.held.rev_off
        none = np.full(len(u), -1, np.int64)
        if narre:
            this = _np_reviews(ttok, toff, tb, te, none, R, W)
            udoc = _np_reviews(us.tok, us.rev_off, urb, ure, ku, R, W)
            idoc = _np_reviews(it.tok, it.rev_off, irb, ire, ki, R, W)
        else:
            this = _np_docs(ttok, toff, tb, te, none, T)
            udoc = _np_docs(us.tok, us.rev_off, urb, ure, ku, T)
            idoc = _np_docs(it.tok, it.rev_off, irb, ire, ki, T)
        return this, who, what, udoc, idoc

    # ------------------------------------------------------------------ device side
    def _device_split(self):
        if self._dev_split is None:
            ku, ki, held = self._split_arrays()
            self._check_owners(self._u, self._i)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            self._dev_split = dict(u=t(self._u), i=t(self._i), y=t(self._y), ku=t(ku), ki=t(ki), held=t(held))
        return self._dev_split

    def _check_owners(self, u, i):
        """The device batcher indexes the pools' ``first`` arrays by id: an id outside the pools is the
        reference's KeyError (``self.user_reviews[u]``, data.py:283-284), raised once on the host."""
        st = self.store
        if len(u) and (int(np.max(u)) >= st.users.owners or int(np.min(u)) < 0):
            raise KeyError('user id outside the review pools: {}'.format(int(np.max(u))))
        if len(i) and (int(np.max(i)) >= st.items.owners or int(np.min(i)) < 0):
            raise KeyError('item id outside the review pools: {}'.format(int(np.max(i))))

    def _review_fields_device(self, u, i, ku, ki, held, lead, nb_item=None, rep_user=1):
        """Same five slots, built by ONE launch of r4r_batch_build from the HBM-resident pools.
        ``u`` .. ``held`` are int64 device tensors of N ratings; ``lead`` is the leading shape of the
        outputs ([B] or [B, 6])."""
        from . import _lib
        from ._lib import ptr
        hp, st = self.hyper_params, self.store
        narre = hp['model_type'] in ['NARRE']
        T = int(hp['input_length'])
        R, W = (int(hp['narre_num_reviews']), int(hp['narre_num_words'])) if narre else (0, 0)
        doc = R * W if narre else T
        n = int(u.numel())
        utok, uoff, ufirst, unb = st.users.device(self.device)
        itok, ioff, ifirst, inb = st.items.device(self.device)
        htok, hoff = st.held.device(self.device)
        block = torch.empty(n * (3 * doc + 2 * NEIGHBOURS), dtype=torch.int64, device=self.device)
        train = self.this_index_user_item is not None
        rc = _lib.lib().r4r_batch_build(
            ptr(utok), ptr(uoff), ptr(ufirst), ptr(unb), ptr(itok), ptr(ioff), ptr(ifirst), ptr(inb),
            ptr(htok), ptr(hoff), ptr(u), ptr(i), ptr(nb_item if nb_item is not None else i), ptr(ku), ptr(ki),
            ptr(held), int(train), ptr(block), n, T, R, W, int(hp['total_users']) + 1, int(hp['total_items']) + 1,
            _lib.current_stream())
        _lib.check(rc, 'r4r_batch_build')
        lead = tuple(lead)
        dshape = lead + ((R, W) if narre else (T,))
        at = 0
        out = []
        for cnt, shape in ((n * doc, dshape), (n * NEIGHBOURS, lead + (NEIGHBOURS,)),
                           (n * NEIGHBOURS, lead + (NEIGHBOURS,)), (n * doc, dshape), (n * doc, dshape)):
            out.append(block[at:at + cnt].view(shape))
            at += cnt
        return out

    def _on_device(self):
        return self.device.type == 'cuda'

    def _device_batch(self, s, e):
        """Ratings [s, e) of the split as one batch built on the device."""
        d = self._device_split()
        u, i = d['u'][s:e], d['i'][s:e]
        f = self._review_fields_device(u, i, d['ku'][s:e], d['ki'][s:e], d['held'][s:e], (e - s,))
        return f + [u, i], d['y'][s:e]

    def batch(self, b):
        """Batch number b of iter() by itself (device loaders): what a caller that runs most of an epoch through
        span_descriptor() uses for the steps it keeps to itself (a ragged tail, a step it wants to look at)."""
        bsz, n = int(self.hyper_params['batch_size']), len(self.data)
        s, e = b * bsz, min(n, (b + 1) * bsz)
        if self.iter == self.iter_simple:
            d = self._device_split_simple()
            return [None, None, None, None, None, d['u'][s:e], d['i'][s:e]], d['y'][s:e]
        return self._device_batch(s, e)

    # ------------------------------------------------------------------ spans (include/r4r.h: r4r_*_span)
    SPAN_RATINGS = 1024          # ratings per r4r_batch_build launch of a span (a group of batches)

    def span_descriptor(self):
        """The epoch as the native K-steps-per-call entry points read it (include/r4r.h, "Spans"): the host array of
        R4R_LOADER_WORDS words over this loader's HBM-resident pools and split arrays, plus the ring the batch groups
        are built into.  None on the host (no device, nothing to enqueue).  The ring's state (`built`) restarts with
        every call: one descriptor per epoch, like one `iter()` per epoch."""
        if not self._on_device():
            return None
        if self._span is None:
            self._span = SpanDescriptor(self)
        self._span.built.value = -1
        return self._span

    # ------------------------------------------------------------------ iterators
    def iter_review(self, eval=False, simple=False, batch=None):
        """data.py:250-333: contiguous slices of batch_size ratings (ragged tail), the 7-slot list.
        ``batch``: another slice length (eval.py's validation passes launch larger slices of the same
        stream: a rating's score does not depend on what shares its launch)."""
        bsz = int(batch or self.hyper_params['batch_size'])
        n = len(self.data)
        if self._on_device() and not simple:
            for s in range(0, n, bsz):
                yield self._device_batch(s, min(n, s + bsz))
            return
        ku, ki, held = self._split_arrays()
        for s in range(0, n, bsz):
            e = min(n, s + bsz)
            u, i = self._u[s:e], self._i[s:e]
            f = list(self._review_fields_host(u, i, ku[s:e], ki[s:e], held[s:e]))
            if simple:                                       # make_quick_data.py:35 consumes plain arrays
                yield f + [u, i], self.data[s:e, 2]
            else:
                yield [torch.from_numpy(a) for a in f + [u, i]], torch.from_numpy(self._y[s:e])

    def iter_simple(self, eval=False, batch=None):
        """data.py:336-372: ids only; the five review slots are None."""
        bsz = int(batch or self.hyper_params['batch_size'])
        n = len(self.data)
        if self._on_device():
            d = self._device_split_simple()
            for s in range(0, n, bsz):
                yield [None, None, None, None, None, d['u'][s:s + bsz], d['i'][s:s + bsz]], d['y'][s:s + bsz]
            return
        for s in range(0, n, bsz):
            yield [None, None, None, None, None, torch.from_numpy(self._u[s:s + bsz]),
                   torch.from_numpy(self._i[s:s + bsz])], torch.from_numpy(self._y[s:s + bsz])

    def _device_split_simple(self):
        if self._dev_simple is None:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            self._dev_simple = dict(u=t(self._u), i=t(self._i), y=t(self._y))
        return self._dev_simple

    def _negs_arrays(self, review):
        """iter_negs' rows (data.py:379-401): per ranking user (dict order) the positive item followed by
        its five negatives; every candidate is paired with the POSITIVE item's overlap indices."""
        key = bool(review)
        if key not in self._dev_negs:
            users = [u for u in self.negs]
            n = len(users)
            cand = np.zeros((n, 6), np.int64)
            for r, u in enumerate(users):
                cand[r] = [self.negs[u][0][0]] + list(self.negs[u][1])
            uu = np.asarray([int(u) for u in users], np.int64).reshape(n)
            ku = np.full(n, -1, np.int64)
            ki = np.full(n, -1, np.int64)
            held = np.full(n, -1, np.int64)
            if review:
                for r in range(n):
                    k = (int(uu[r]), int(cand[r, 0]))
                    if self.this_index_user_item is not None:
                        ku[r], ki[r] = self.store.pair[k]                      # KeyError like data.py:217
                    elif self.test_reviews is not None:
                        held[r] = self.store.held.index[k]                     # KeyError like data.py:244
                self._check_owners(uu, cand.reshape(-1))
            self._dev_negs[key] = dict(u=uu, cand=cand, ku=ku, ki=ki, held=held)
        return self._dev_negs[key]

    def iter_negs(self, review, batch=None):
        """data.py:375-447 -> ([this [B,6,..], who [B,6,10], what [B,6,10], user docs, item docs,
        user [B,6], item [B,6]], zeros [B]).  With review == False the five review slots are EMPTY
        tensors, as ``LongTensor([])`` of the reference's unfilled lists.  ``batch``: rows per slice."""
        bsz = int(batch or self.hyper_params['batch_size'])
        a = self._negs_arrays(review)
        n = len(a['u'])
        dev = self._on_device()
        for s in range(0, n, bsz):
            e = min(n, s + bsz)
            b = e - s
            u6 = np.repeat(a['u'][s:e, None], 6, axis=1)
            i6 = a['cand'][s:e]
            if not review:
                data = [torch.zeros(0, dtype=torch.int64) for _ in range(5)] + \
                    [torch.from_numpy(np.ascontiguousarray(u6)), torch.from_numpy(np.ascontiguousarray(i6))]
                y = torch.zeros(b, dtype=torch.float32)
                if dev:
                    data, y = [d.to(self.device) for d in data], y.to(self.device)
                yield data, y
                continue
            rep = lambda x: np.repeat(x[s:e], 6)
            pos6 = np.repeat(i6[:, 0], 6)
            if dev:
                t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
                ud, idv = t(u6.reshape(-1)), t(i6.reshape(-1))
                f = self._review_fields_device(ud, idv, t(rep(a['ku'])), t(rep(a['ki'])), t(rep(a['held'])),
                                               (b, 6), nb_item=t(pos6))
                yield f + [ud.view(b, 6), idv.view(b, 6)], torch.zeros(b, dtype=torch.float32, device=self.device)
                continue
            f = self._review_fields_host(u6.reshape(-1), i6.reshape(-1), rep(a['ku']), rep(a['ki']),
                                         rep(a['held']), own_user_pos=pos6)
            f = [x.reshape((b, 6) + x.shape[1:]) for x in f]
            yield [torch.from_numpy(np.ascontiguousarray(x)) for x in f + [u6, i6]], \
                torch.zeros(b, dtype=torch.float32)


def load_data(hyper_params, load_negs=True, device=None):
    """data.py:449-485: reads the pickles of ``hyper_params['data_dir']`` and returns
    (train_loader, test_loader, val_loader, hyper_params)."""
    print('Loading data...')
    d = hyper_params['data_dir']
    train, test, val = load_obj(d + 'train'), load_obj(d + 'test'), load_obj(d + 'val')
    user_reviews, item_reviews = load_obj(d + 'user_reviews'), load_obj(d + 'item_reviews')
    negs = load_obj(d + 'negs') if load_negs else None
    this_index_user_item = load_obj(d + 'this_index_user_item')
    test_reviews = load_obj(d + 'test_reviews')
    num_users, num_items, num_words = load_obj(d + 'num_users_items')
    hyper_params['total_users'] = num_users
    hyper_params['total_items'] = num_items
    hyper_params['total_words'] = num_words
    train_loader = DataLoader(hyper_params, train, user_reviews, item_reviews, negs,
                              this_index_user_item=this_index_user_item, device=device)
    return train_loader, \
        DataLoader(hyper_params, test, user_reviews, item_reviews, negs, test_reviews=test_reviews,
                   train_loader=train_loader, device=device), \
        DataLoader(hyper_params, val, user_reviews, item_reviews, negs, test_reviews=test_reviews,
                   train_loader=train_loader, device=device), \
        hyper_params
