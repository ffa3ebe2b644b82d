"""Fast batcher (counterpart of the reference's data_fast.py).

Same contract as ``data_fast.DataLoader`` (data_fast.py:14-123): the whole split is
resident in host RAM as 8 arrays ``a..h`` (this_reviews, users_who_reviewed,
reviewed_items, user_reviews, item_reviews, user, item, rating -- the datasets
make_quick_data.py:23-32 writes), ``len(loader)`` is the number of batches, and
``iter()`` yields contiguous slices of ``batch_size`` rows -- no shuffling, ragged
last batch -- as ``([7 int64 device tensors], float32 device tensor)``.

On-disk format: the reference's own gzip HDF5 epoch files (``train.hdf5`` / ``val.hdf5`` /
``test.hdf5`` under ``quick_data_*/<data_dir>``, written by data_scripts/make_quick_data.py:21-44)
are read as they are -- through h5py where it exists, otherwise through ``hdf5_lite``, this
package's own reader of that container format (the MI355X image has no h5py / libhdf5).  Where
no ``.hdf5`` file exists the same 8 datasets are taken from one ``.npz`` per split (``train.npz``
..., i8 / f8 like the HDF5 writer; ``save_split`` / tools/make_quick_data.py write them).
``from_arrays`` builds a loader from in-memory arrays (synthetic data).

Host->device: the reference builds 8 tensors per batch from pageable numpy slices
(3 MB of int64 per DeepCoNN batch, synchronous, on the compute stream).  Here the arrays
are pinned once and batch k+1 is copied on a dedicated COPY stream while batch k is being
trained on (double buffering): 2 MB of indices is ~33 us at PCIe Gen5 x16, a quarter of a
0.137 ms DeepCoNN step if it were left on the compute stream.  The consumer's stream waits
on the copy's event before using a batch, and the copy stream waits for the consumer to be
done with a buffer before refilling it (record_stream).
"""
import os

import numpy as np
import torch

from .utils import load_obj

KEYS = 'abcdefgh'


class DataLoader():
    def __init__(self, hyper_params, file_name=None, arrays=None, device=None):
        self.hyper_params = hyper_params
        self.bsz = int(hyper_params['batch_size'])
        self.file_name = file_name
        self.device = device if device is not None else (
            torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu'))
        if arrays is None:
            init_path = '/'.join(hyper_params['data_dir'].split('/')[1:])
            root = 'quick_data_narre/' if hyper_params['model_type'] in ['NARRE'] else 'quick_data_deepconn/'
            arrays = read_split(root + init_path + file_name)
        self.total = len(arrays['a'])
        self._copy_stream = None
        pin = torch.cuda.is_available()
        self._t = {}
        for k in KEYS:
            dt = np.float32 if k == 'h' else np.int64
            self._t[k] = torch.from_numpy(np.ascontiguousarray(arrays[k]).astype(dt, copy=False))
        self._tail = {k: tuple(self._t[k].shape[1:]) for k in KEYS}      # per-row shape of every field
        arrays = None
        # Batches are fixed contiguous slices (the reference never shuffles, data_fast.py:99), so each
        # batch is packed ONCE into one contiguous pinned block [a | b | ... | g | h-as-int64-bits]:
        # one H2D copy per batch instead of eight, and the device-side fields are contiguous views.
        # H2D_GROUP consecutive batches share one pinned block and travel in ONE copy: a copy's
        # event wait on the compute stream costs a few microseconds of queue serialisation, which at
        # 0.11 ms per step is worth amortising (16 MB per copy for DeepCoNN batches of 128).
        self.group = max(1, int(hyper_params.get('h2d_group', 8)))
        self._packed, self._layout = [], []
        if pin:
            def block(index):
                sl = slice(index, index + self.bsz)
                parts = [self._t[k][sl].reshape(-1) for k in KEYS[:7]]
                yb = self._t['h'][sl]
                ypad = torch.zeros(yb.numel() + (yb.numel() & 1), dtype=torch.float32)
                ypad[:yb.numel()] = yb
                parts.append(ypad.view(torch.int64))
                return torch.cat(parts)
            starts = list(range(0, self.total, self.bsz))
            for g0 in range(0, len(starts), self.group):       # one group at a time: never a second full copy
                grp = [block(i) for i in starts[g0:g0 + self.group]]
                self._packed.append(torch.cat(grp).pin_memory())
                offs, at = [], 0
                for blk in grp:
                    offs.append((at, blk.numel()))
                    at += blk.numel()
                self._layout.append(offs)
            # the split now lives ONCE, in the pinned blocks (Electronics-sized DeepCoNN data is tens of GB):
            # the unpacked arrays are dropped, host-side iteration re-slices the blocks
            self._t = None

    @classmethod
    def from_arrays(cls, hyper_params, data, y, device=None):
        arrays = dict(zip(KEYS[:7], data))
        arrays['h'] = y
        return cls(hyper_params, arrays=arrays, device=device)

    def __len__(self):
        return int(self.total // self.bsz) + int(self.total % self.bsz > 0)

    def batch_sizes(self):
        """Rows of every batch iter() will yield (data parallelism sums them across ranks once per epoch)."""
        return [min(self.bsz, self.total - i) for i in range(0, self.total, self.bsz)]

    def _stage(self, grp, stream):
        """Enqueue the single H2D copy of batch group `grp` on `stream`; returns the per-batch
        device views + the event + the device block."""
        import torch as _torch
        with _torch.cuda.stream(stream):
            dev = self._packed[grp].to(self.device, non_blocking=True)
            ev = _torch.cuda.Event()
            ev.record(stream)
        out = []
        for j, (off, cnt_blk) in enumerate(self._layout[grp]):
            index = (grp * self.group + j) * self.bsz
            n = min(self.bsz, self.total - index)
            blk = dev[off:off + cnt_blk]
            data, at = [], 0
            for k in KEYS[:7]:
                shape = (n,) + self._tail[k]
                cnt = int(np.prod(shape))
                data.append(blk[at:at + cnt].view(shape))
                at += cnt
            out.append((data, blk[at:].view(_torch.float32)[:n]))
        return out, ev, dev

    def _host_batches(self):
        """(fields, ratings) of every batch as host tensors: slices of the arrays, or -- once they were
        packed into pinned blocks and dropped -- views of those blocks."""
        import torch as _torch
        if self._t is not None:
            for index in range(0, self.total, self.bsz):
                sl = slice(index, index + self.bsz)
                yield [self._t[k][sl] for k in KEYS[:7]], self._t['h'][sl]
            return
        for grp, offs in enumerate(self._layout):
            for j, (off, cnt_blk) in enumerate(offs):
                n = min(self.bsz, self.total - (grp * self.group + j) * self.bsz)
                blk = self._packed[grp][off:off + cnt_blk]
                data, at = [], 0
                for k in KEYS[:7]:
                    shape = (n,) + self._tail[k]
                    cnt = int(np.prod(shape))
                    data.append(blk[at:at + cnt].view(shape))
                    at += cnt
                yield data, blk[at:].view(_torch.float32)[:n]

    def iter(self, eval=False, torch=True):
        import torch as _torch
        if not torch:
            for data, y in self._host_batches():
                yield [d.numpy() for d in data], y.numpy()
            return
        if self.device.type != 'cuda':
            for data, y in self._host_batches():
                yield [d.to(self.device) for d in data], y.to(self.device)
            return
        if self._copy_stream is None:
            self._copy_stream = _torch.cuda.Stream(device=self.device)
        ng = len(self._packed)
        nxt = self._stage(0, self._copy_stream) if ng else None
        for g in range(ng):
            batches, ev, dev = nxt
            nxt = self._stage(g + 1, self._copy_stream) if g + 1 < ng else None   # prefetch the next group
            cur = _torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                              # the group has landed
            dev.record_stream(cur)                          # the allocator must not reuse it under the trainer
            for item in batches:
                yield item


def read_split(path):
    """The 8 arrays of one split: the reference's HDF5 file (data_fast.py:31-45 reads it whole, and so does this),
    or the ``.npz`` beside it when there is no such file."""
    if path.endswith('.hdf5') and not os.path.exists(path) and os.path.exists(path[:-5] + '.npz'):
        path = path[:-5] + '.npz'
    if path.endswith('.npz'):
        with np.load(path) as z:
            return {k: z[k] for k in KEYS}
    try:
        import h5py
        opener = h5py.File
    except ImportError:
        from . import hdf5_lite
        opener = hdf5_lite.File
    with opener(path, 'r') as f:
        return {k: f[k][:] for k in KEYS}


def save_split(path, data, y):
    """Write one split (i8 ids, f8 ratings): ``*.hdf5`` in the reference's own format -- the eight gzip datasets
    make_quick_data.py:21-32 creates, readable by the reference's h5py loader -- or ``*.npz``."""
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    arrays = {k: np.asarray(d, dtype=np.int64) for k, d in zip(KEYS[:7], data)}
    arrays['h'] = np.asarray(y, dtype=np.float64)
    if path.endswith('.hdf5'):
        from . import hdf5_lite
        hdf5_lite.write_file(path, arrays, compression='gzip')
    else:
        np.savez(path, **arrays)


def load_data_fast(hyper_params):
    print('Loading data...')
    num_users, num_items, num_words = load_obj(hyper_params['data_dir'] + 'num_users_items')
    hyper_params['total_users'] = num_users
    hyper_params['total_items'] = num_items
    hyper_params['total_words'] = num_words
    train_loader = DataLoader(hyper_params, 'train.hdf5')
    test_loader = DataLoader(hyper_params, 'test.hdf5')
    val_loader = DataLoader(hyper_params, 'val.hdf5')
    return train_loader, test_loader, val_loader, hyper_params
