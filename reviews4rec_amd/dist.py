"""Data parallelism, one process per GPU, over RCCL (torch.distributed 'nccl' on
ROCm) -- new work: the reference is single-process, single-device
(main.py:407).

Every example's forward/backward is independent given the weights and the loss
is a mean over the batch (main.py:58), so a global batch is split contiguously
across ranks and the only exchange is the gradient sum:

  C1  ONE all-reduce per step over ONE flat fp32 bucket holding every DENSE-layer
      gradient that exists (DeepCoNN @E=300: 182,402 floats = 0.73 MB).  On the
      xGMI full mesh a sub-MB payload is latency-bound, so a single bucket -- not
      per-tensor calls, not ring-sized buckets -- is the right shape.
  C2  ID-embedding tables and bias vectors (MF_dot on Electronics: 16.6 M floats =
      66 MB if reduced densely) never cross xGMI as dense gradients.  Their backward
      records compact (row-id, gradient-row) lists (ops.SparseGradCapture); each rank
      pads its list to the step's common length, ONE all_gather per table moves
      B_global x (D + 2) floats, and every rank rebuilds the identical dense gradient
      with a fixed summation order (r4r_embed_scatter_add_ordered), then runs the
      same dense Adam (untouched rows still move by weight decay: SURVEY.md fact 4).
  Parameters with no gradient (DeepCoNN's unused `final` MLP and biases in
  'deepconn' mode) are left out of the bucket; the active set
  is agreed once across ranks (a rank whose shard is empty contributes zeros).

Weights are replicated; after an identical all-reduced gradient every rank runs
the identical dense Adam sweep, so replicas stay bit-identical without any
parameter traffic.  The loss must be scaled by 1/B_global (``loss_scale``), not
1/B_local, so ragged shards weigh correctly.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the job torchrun / torch.distributed.run started (RANK, WORLD_SIZE, LOCAL_RANK,
    MASTER_ADDR, MASTER_PORT).  Returns (rank, world, local_rank); a no-op at WORLD_SIZE=1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        # R4R_DIST_BACKEND=gloo lets several ranks share one GPU (test rigs with a single device):
        # the rank -> device map wraps around instead of failing
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if (world > 1 or os.environ.get('R4R_DP_SINGLE') == '1') and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('R4R_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class AgreedFailure(RuntimeError):
    """A set-up failure EVERY rank of the group raises together (the ranks agreed on it over the process group): the
    caller may return without another collective.  Any other exception out of StreamRccl() is one rank's own."""


class StreamRccl:
    """A RCCL communicator of this package's own, called through ctypes ON THE CALLER'S STREAM.

    torch.distributed issues a collective on its process group's internal stream and brackets it with two event
    hand-overs (compute -> collective stream -> compute); at the sizes of this path (DeepCoNN: one 0.73 MB
    bucket per step on a 0.1 ms step) those hand-overs and the extra dependent launch cost more than the
    collective (measured at one rank: 17-24 us per step).  ncclAllReduce / ncclAllGather take the
    stream as an argument: issued on the step's own stream the collective is one more launch in the chain
    -- no events, and the optimiser launch behind it needs no synchronisation either.  The unique id travels
    over the torch.distributed group that already exists (any backend).  RCCL is the library torch itself loaded
    (torch/lib/librccl.so): no second copy in the process."""
    FLOAT32, SUM = 7, 0                                    # ncclDataType_t / ncclRedOp_t (rccl.h)

    class _UniqueId(__import__('ctypes').Structure):
        _fields_ = [('internal', __import__('ctypes').c_byte * 128)]

    INIT_TIMEOUT_S = 60.0                                  # ncclCommInitRank is bounded by this (R4R_RCCL_INIT_TIMEOUT)

    def __init__(self, group=None):
        """Collective over `group`; every step that can fail on ONE rank only is followed by an agreement over the
        existing process group, so that a local failure makes every rank raise here together instead of leaving the
        others inside a broadcast or inside RCCL's bootstrap:
          1. load the library and resolve its symbols          -> agree
          2. rank 0 draws the unique id; it broadcasts None when that failed (it always joins the broadcast)
          3. ncclCommInitRank in a helper thread, joined for at most INIT_TIMEOUT_S -> agree
        A rank whose init never returns leaves a parked daemon thread behind and reports failure like the others."""
        import ctypes
        import threading
        self.ct = ctypes
        self.comm = None
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        err = None
        try:
            path = os.environ.get('R4R_RCCL_LIBRARY') or os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            lib = self.lib = ctypes.CDLL(path)
            lib.ncclGetErrorString.restype = ctypes.c_char_p
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._UniqueId, ctypes.c_int]
            lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
            lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p]
            lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            lib.ncclCommAbort.argtypes = [ctypes.c_void_p]
            lib.ncclGetUniqueId                               # (resolved now: a missing symbol is a stage-1 failure)
        except Exception as e:                               # noqa: BLE001
            err = 'loading RCCL: %s: %s' % (type(e).__name__, e)
        self._agree(group, err)
        uid = self._UniqueId()
        box = [None]
        if self.rank == 0:
            rc = lib.ncclGetUniqueId(ctypes.byref(uid))
            box = [bytes(bytearray(uid.internal)) if rc == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None:                                   # (every rank sees the same box: a collective decision)
            raise AgreedFailure('RCCL ncclGetUniqueId failed on rank 0')
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        comm, done = ctypes.c_void_p(), {}
        dev = torch.cuda.current_device()

        def init():
            try:
                torch.cuda.set_device(dev)                   # (a new thread starts on device 0)
                done['rc'] = lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank)
            except Exception as e:                           # noqa: BLE001
                done['exc'] = e

        th = threading.Thread(target=init, name='r4r-rccl-init', daemon=True)
        th.start()
        th.join(float(os.environ.get('R4R_RCCL_INIT_TIMEOUT', self.INIT_TIMEOUT_S)))
        if th.is_alive():
            err = 'ncclCommInitRank did not return within its time limit'
        elif 'exc' in done:
            err = 'ncclCommInitRank raised %r' % (done['exc'],)
        elif done.get('rc', 1) != 0:
            err = 'ncclCommInitRank failed: %s' % lib.ncclGetErrorString(done['rc']).decode()
        else:
            self.comm = comm
        try:
            self._agree(group, err)
        except AgreedFailure:
            self.close()
            raise

    def _agree(self, group, err):
        """Every rank passes its local error (or None); all raise if any rank has one."""
        dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
        flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if err:
            raise AgreedFailure('StreamRccl: ' + err)
        if int(flag.item()) == 0:
            raise AgreedFailure('StreamRccl: another rank failed to set the communicator up')

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('RCCL %s failed: %s' % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_reduce(self, t):
        """In-place fp32 sum of `t` over the ranks, enqueued on torch's current stream."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self.FLOAT32, self.SUM, self.comm,
                                           torch.cuda.current_stream(t.device).cuda_stream), 'ncclAllReduce')

    def all_gather(self, out, t):
        """Rank r's `t` into out[r * n : (r + 1) * n], enqueued on torch's current stream."""
        nbytes = t.numel() * t.element_size()
        assert t.dtype == out.dtype and t.is_contiguous() and out.is_contiguous() and out.numel() == self.world * t.numel()
        assert nbytes % 4 == 0                               # (moved as 4-byte words: an all_gather does no arithmetic)
        self._check(self.lib.ncclAllGather(t.data_ptr(), out.data_ptr(), nbytes // 4, self.FLOAT32,
                                           self.comm, torch.cuda.current_stream(t.device).cuda_stream), 'ncclAllGather')

    def close(self, abort=False):
        """Destroy the communicator (abort=True: ncclCommAbort -- a communicator with a collective that never ends)."""
        if getattr(self, 'comm', None):
            (self.lib.ncclCommAbort if abort else self.lib.ncclCommDestroy)(self.comm)
            self.comm = None


class PeerExchange:
    """The all_gather of the ranks' flat gradient buffers done by ONE kernel per rank over peer-mapped memory
    (csrc/peer.hip).  Every rank owns one fine-grained segment [flag array | gathered buffer, even steps | odd steps]
    (r4r_peer_segment_create), ships its 64-byte IPC handle through the process group and maps everybody else's
    (r4r_peer_segment_open).  `exchange` is one launch: r4r_peer_push writes this rank's gradient into its slot on
    every rank, raises a flag there, and waits (bounded) for every rank's flag in its own array; r4r_adam_gathered
    then sums the slots in rank order.  No collective library call in the step: on the xGMI mesh the exchange is one
    store stream per link plus a flag.  Functionally testable on one GPU (two processes map each other's segments on
    the same device: tests/test_gpu_dist.py); `R4R_DP_EXCHANGE=peer` selects it in the DeepCoNN engine.

    Two gathered buffers alternate by step parity (csrc/peer.hip explains why that is enough)."""
    TIMEOUT_S = 20.0
    FLAG_BYTES = 4096

    def __init__(self, numel, device, group=None):
        import ctypes
        from . import _lib
        lib = _lib.lib()
        self.group, self.dev = group, torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.numel = int(numel)
        half = -(-self.world * self.numel * 4 // 4096) * 4096
        self.bytes = self.FLAG_BYTES + 2 * half
        self.local = torch.zeros(16, dtype=torch.int32, device=device)          # [0] arrival counter, [1] timeout word
        handle = (ctypes.c_uint8 * 64)()
        mine = ctypes.c_void_p()
        with torch.cuda.device(self.dev):
            _lib.check(lib.r4r_peer_segment_create(self.bytes, ctypes.byref(mine), handle), 'r4r_peer_segment_create')
        self._mine, self._mapped = mine.value, []
        everyone = [None] * self.world
        dist.all_gather_object(everyone, bytes(handle), group=group)
        base = [0] * self.world
        for r, h in enumerate(everyone):
            if r == self.rank:
                base[r] = self._mine
                continue
            p = ctypes.c_void_p()
            with torch.cuda.device(self.dev):
                _lib.check(lib.r4r_peer_segment_open((ctypes.c_uint8 * 64).from_buffer_copy(h), ctypes.byref(p)),
                           'r4r_peer_segment_open')
            self._mapped.append(p.value)
            base[r] = p.value
        u64 = lambda v: torch.tensor(v + [0] * (16 - len(v)), dtype=torch.int64)
        # host arrays (the launcher copies them into the kernel arguments)
        self._dst = [u64([b + self.FLAG_BYTES + par * half for b in base]) for par in (0, 1)]
        self._flg = u64(base)
        self.gathered = [self._mine + self.FLAG_BYTES + par * half for par in (0, 1)]   # device addresses
        dist.barrier(group=group)                            # nobody pushes before everybody has mapped

    def exchange(self, flat, epoch):
        """Push `flat` for step `epoch` (1-based) and wait for every rank's push: -> the device address of this
        step's gathered buffer [world][numel]."""
        from . import _lib
        par = int(epoch) & 1
        _lib.check(_lib.lib().r4r_peer_push(flat.data_ptr(), self.numel, self._dst[par].data_ptr(), self._flg.data_ptr(),
                                            self.local.data_ptr(), self.rank, self.world, int(epoch) & 0x7fffffff,
                                            self._mine, self.local.data_ptr() + 4, self.TIMEOUT_S, _lib.current_stream()),
                   'r4r_peer_push')
        return self.gathered[par]

    def check(self):
        """Raise if a wait timed out (a host sync: call it at epoch ends, not per step)."""
        word = int(self.local[1].item())
        if word:
            # (the device side skips every later wait while the word is set -- csrc/peer_device.h -- so it is cleared
            # with the report: a caller that catches this and goes on gets its waits back, not a whole epoch of
            # exchanges over incomplete buffers)
            self.local[1].zero_()
            raise RuntimeError('PeerExchange: rank %d never raised its flag (waited %.0f s)' % (word - 1, self.TIMEOUT_S))

    def close(self):
        """Unmap the peers' segments and free this rank's (after a barrier: nobody still writes into it)."""
        from . import _lib
        if self._mine is None:
            return
        torch.cuda.synchronize(self.dev)
        dist.barrier(group=self.group)
        for p in self._mapped:
            _lib.check(_lib.lib().r4r_peer_segment_close(p), 'r4r_peer_segment_close')
        dist.barrier(group=self.group)
        _lib.check(_lib.lib().r4r_peer_segment_destroy(self._mine), 'r4r_peer_segment_destroy')
        self._mine, self._mapped = None, []


def shard_bounds(n, rank, world):
    """Contiguous split of n rows: the first n % world ranks get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(data, y, rank, world):
    """This rank's contiguous slice of a global batch (the 7-slot list + ratings)."""
    lo, hi = shard_bounds(y.shape[0], rank, world)
    return [None if d is None else d[lo:hi] for d in data], y[lo:hi]


class DataParallel:
    def __init__(self, model, group=None, sparse_tables=True, rebuild_fn=None):
        """sparse_tables: exchange ID-table / bias gradients as compact lists (C2) instead of
        putting them into the dense bucket.  rebuild_fn(idx, g, R, D, out) -> dense [R, D]
        gradient (default: the HIP ordered scatter; tests on the CPU inject their own)."""
        self.model = model
        self.group = group
        # (R4R_DP_SINGLE=1: a ONE-rank job keeps the data-parallel path on -- the collectives move nothing, but
        # every one of them is issued, which is how a single-GPU box exercises RCCL: tests/test_gpu_dist.py)
        self.on = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get('R4R_DP_SINGLE') == '1')
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._active = None          # indices into self.params that carry gradients
        self._bucket = None
        self.sparse = bool(sparse_tables) and self.on
        self._rebuild = rebuild_fn
        self._sparse_set = None      # indices into self.params exchanged as compact lists
        self._dense_grads = {}
        if self.sparse:
            from . import ops
            ops.SparseGradCapture.active = True
            ops.SparseGradCapture.clear()
        # the fused engines' exchanges go through a communicator of our own on the compute stream where the job
        # runs over RCCL (R4R_DP_RCCL=0: torch.distributed's collectives, as before)
        self.stream_rccl = None
        if self.on and os.environ.get('R4R_DP_RCCL', '1') != '0' and dist.get_backend(group) == 'nccl':
            self.stream_rccl = self._checked_stream_rccl(group)

    @staticmethod
    def _checked_stream_rccl(group):
        """StreamRccl, proven on this job's fabric before anything depends on it: one all-reduce and one all-gather
        with known answers; every rank must see both right, or ALL ranks fall back to torch.distributed's
        collectives together (a warning, not an error: the exchange is an optimisation of the same sums)."""
        import time
        import warnings
        comm, ok, why, hung = None, 1, '', False
        try:
            comm = StreamRccl(group)
            dev = torch.device('cuda', torch.cuda.current_device())
            # on a stream of its own, waited for with a deadline: a collective that never ends must not sit in the
            # compute stream (everything enqueued behind it would wait for ever) nor block this thread in a synchronize
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                t = torch.full((1024,), float(comm.rank + 1), device=dev)
                comm.all_reduce(t)
                g = torch.empty(comm.world * 256, device=dev)
                comm.all_gather(g, torch.full((256,), float(comm.rank), device=dev))
                ev = torch.cuda.Event()
                ev.record(side)
            deadline = time.monotonic() + float(os.environ.get('R4R_RCCL_CHECK_TIMEOUT', 30.0))
            while not ev.query():
                if time.monotonic() > deadline:
                    hung = True
                    raise RuntimeError('the known-answer collectives did not finish within their time limit')
                time.sleep(0.001)
            torch.cuda.current_stream(dev).wait_stream(side)
            want = torch.arange(comm.world, device=dev, dtype=torch.float32).repeat_interleave(256)
            if not (bool((t == comm.world * (comm.world + 1) / 2).all()) and torch.equal(g, want)):
                ok, why = 0, 'a collective returned wrong values'
        except AgreedFailure as e:                           # every rank raised this together: no further agreement needed
            warnings.warn('reviews4rec_amd.dist: the on-stream RCCL communicator is not usable here (%s); using '
                          "torch.distributed's collectives" % e, RuntimeWarning)
            return None
        except Exception as e:                               # noqa: BLE001 -- this rank's own failure: every rank must hear of it
            ok, why = 0, '%s: %s' % (type(e).__name__, e)
        flag = torch.tensor([ok], device='cuda', dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            return comm
        if comm is not None:
            comm.close(abort=hung)
        warnings.warn('reviews4rec_amd.dist: the on-stream RCCL communicator is not usable here (%s); using '
                      "torch.distributed's collectives" % (why or 'another rank reported a failure'), RuntimeWarning)
        return None

    def close(self):
        """Destroy the package's own RCCL communicator (collective in effect: every rank calls it, with its device idle --
        a communicator left alive at interpreter exit is torn down in whatever order the process dies in)."""
        comm, self.stream_rccl = self.stream_rccl, None
        if comm is not None:
            torch.cuda.synchronize()
            if dist.is_initialized():
                dist.barrier(group=self.group)
            comm.close()

    def rebind(self, model):
        """Point the exchange at `model` (a new training stage, a freshly built model): its parameters
        become the exchanged set, every cached decision about the previous model -- which parameters
        carry gradients, the flat bucket, the compact-list tables, their dense buffers -- is dropped,
        stale compact-gradient records are cleared, and rank 0's weights are broadcast so the replicas
        start identical (each rank ran its own xavier_init)."""
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._active = self._bucket = self._sparse_set = None
        self._dense_grads = {}
        if self.sparse:
            from . import ops
            ops.SparseGradCapture.clear()
        self.broadcast_parameters()

    def barrier(self):
        if self.on:
            dist.barrier(group=self.group)

    def gather_ints(self, values):
        """[[rank 0's values], [rank 1's], ...] of a short list of host integers."""
        if not self.on:
            return [list(values)]
        dev = self.params[0].device if self.params else torch.device('cpu')
        mine = torch.tensor(list(values), dtype=torch.int64, device=dev)
        out = torch.empty(self.world * mine.numel(), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        return out.view(self.world, -1).tolist()

    def epoch_counts(self, reader):
        """Global size of every batch of the coming epoch, from the ranks' own batch sizes: one
        collective per epoch.  None when the reader cannot tell its batch sizes up front.  Ranks must
        hold the same NUMBER of batches (possibly empty ones) -- anything else would hang the first
        collective of the shorter rank's missing step, so it is an error here."""
        if not self.on:
            return None
        sizes = reader.batch_sizes() if hasattr(reader, 'batch_sizes') else None
        dev = self.params[0].device if self.params else torch.device('cpu')
        n = torch.tensor([-1 if sizes is None else len(sizes)], dtype=torch.int64, device=dev)
        lo, hi = n.clone(), n.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if int(lo) < 0:
            return None
        if int(lo) != int(hi):
            raise RuntimeError('data parallel: ranks hold %d..%d batches this epoch; shard every global batch '
                               '(dist.shard_batch) so that all ranks step together' % (int(lo), int(hi)))
        t = torch.tensor(sizes, dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.tolist()

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank `src`'s weights (buffers included)."""
        if not self.on:
            return
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)

    def loss_scale(self, n_local, n_global):
        """Factor turning sum-of-local-SE into this rank's share of mean(SE) over the GLOBAL batch."""
        return 1.0 / float(n_global)

    def global_count(self, n_local, device):
        if not self.on:
            return int(n_local)
        t = torch.tensor([float(n_local)], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def _agree_active_set(self, device, contributions):
        have = torch.tensor([1.0 if p.grad is not None else 0.0 for p in self.params], device=device)
        sp = torch.tensor([1.0 if p.data_ptr() in contributions else 0.0 for p in self.params], device=device)
        if self.on:
            dist.all_reduce(have, op=dist.ReduceOp.MAX, group=self.group)
            dist.all_reduce(sp, op=dist.ReduceOp.MAX, group=self.group)
        self._sparse_set = [i for i, h in enumerate(sp.tolist()) if h > 0]
        self._active = [i for i, h in enumerate(have.tolist()) if h > 0 and i not in set(self._sparse_set)]
        total = sum(self.params[i].numel() for i in self._active)
        self._bucket = torch.zeros(total, dtype=torch.float32, device=device)

    @torch.no_grad()
    def _exchange_sparse(self, device, contributions):
        """C2: all_gather the compact (row-id, grad-row) lists and rebuild dense gradients."""
        if not self._sparse_set:
            return
        rebuild = self._rebuild
        if rebuild is None:
            from . import ops
            rebuild = ops.rebuild_dense
        locals_ = []
        for i in self._sparse_set:
            p = self.params[i]
            D = 1 if p.dim() == 1 else p.shape[1]
            parts = contributions.get(p.data_ptr(), [])
            if parts:
                idx = torch.cat([a.reshape(-1) for a, _ in parts])
                g = torch.cat([b.reshape(-1, D) for _, b in parts])
            else:
                idx = torch.empty(0, dtype=torch.int64, device=device)
                g = torch.empty((0, D), dtype=torch.float32, device=device)
            locals_.append((p, D, idx, g))
        counts = torch.tensor([float(idx.numel()) for _, _, idx, _ in locals_], device=device)
        dist.all_reduce(counts, op=dist.ReduceOp.MAX, group=self.group)       # common padded length per table
        for (p, D, idx, g), nmax in zip(locals_, counts.tolist()):
            nmax = int(nmax)
            pad_idx = torch.full((nmax,), -1, dtype=torch.int64, device=device)
            pad_g = torch.zeros((nmax, D), dtype=torch.float32, device=device)
            pad_idx[:idx.numel()] = idx
            pad_g[:idx.numel()] = g
            all_idx = torch.empty((self.world * nmax,), dtype=torch.int64, device=device)
            all_g = torch.empty((self.world * nmax, D), dtype=torch.float32, device=device)
            dist.all_gather_into_tensor(all_idx, pad_idx, group=self.group)
            dist.all_gather_into_tensor(all_g, pad_g, group=self.group)
            R = p.shape[0]
            out = self._dense_grads.get(id(p))
            if out is None:
                out = self._dense_grads[id(p)] = torch.empty((R, D), dtype=torch.float32, device=device)
            p.grad = rebuild(all_idx, all_g, R, D, out).view_as(p)

    @torch.no_grad()
    def allreduce_grads(self):
        """Sum gradients across ranks through one flat bucket; afterwards every active
        parameter's .grad is a view into the reduced bucket."""
        if not self.on:
            return
        device = self.params[0].device
        contributions = {}
        if self.sparse:
            from . import ops
            contributions = ops.SparseGradCapture.contributions
        if self._active is None:
            self._agree_active_set(device, contributions)
        if self.sparse:
            self._exchange_sparse(device, contributions)
            ops.SparseGradCapture.clear()
        bucket = self._bucket
        off = 0
        views = []
        for i in self._active:
            p = self.params[i]
            n = p.numel()
            v = bucket[off:off + n].view_as(p)
            if p.grad is None:
                v.zero_()                                   # empty shard on this rank
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            views.append((p, v))
            off += n
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
        for p, v in views:
            p.grad = v

    @torch.no_grad()
    def allreduce_flat(self, flat):
        """The fused engine's gradients already live in one flat buffer: one all-reduce, no copies."""
        if self.on:
            if self.stream_rccl is not None and flat.dtype == torch.float32:
                self.stream_rccl.all_reduce(flat)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, out, t):
        """Rank r's contiguous `t` into out[r * n : (r + 1) * n] (flat views; any 4- or 8-byte dtype), on the
        compute stream through the package's own communicator where the job runs over RCCL."""
        if self.stream_rccl is not None and (t.numel() * t.element_size()) % 4 == 0 and t.is_cuda:
            self.stream_rccl.all_gather(out, t)
        else:
            dist.all_gather_into_tensor(out, t, group=self.group)

    def gather_flat(self, flat, out):
        """All ranks' flat gradient buffers back to back in `out` [world * n] (rank order): the
        one-phase alternative to the all-reduce; the sum happens in the optimiser kernel
        (r4r_adam_gathered), in rank order, so every rank computes the same bits."""
        if self.on:
            if self.stream_rccl is not None and (flat.numel() * flat.element_size()) % 4 == 0:
                self.stream_rccl.all_gather(out, flat)
            else:
                dist.all_gather_into_tensor(out, flat, group=self.group)
        else:
            out[:flat.numel()].copy_(flat)

    @torch.no_grad()
    def sum_scalar(self, t):
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t
