"""Stage timeline of the one-workgroup-per-rating head kernels (NARRE, TransNet) on one MI355X.

`make -C reviews4rec_amd/csrc trace` first (s_memrealtime stamps after every barrier of the head
under -DR4R_TRACE; never loaded by the product path), then
`python tools/head_trace.py cfg4_narre_kindle|cfg5_transnetpp_synthetic` prints, per stage, the
median and the maximum over workgroups of the time since the previous stamp."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['R4R_LIBRARY'] = os.path.join(ROOT, 'reviews4rec_amd/csrc', os.environ.get('TRACE_SO', 'libr4r_hip_trace.so'))
import torch
import reviews4rec_amd
from reviews4rec_amd import synthetic
from reviews4rec_amd.utils import xavier_init

workload = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 'cfg4_narre_kindle'
hp = synthetic.hyper_params_for(workload, dropout=0.6)
B = hp['batch_size']
hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
gen = synthetic.Generator(hp, seed=5, doc_fill='lognormal', token_dist='zipf')
pool = []
for _ in range(4):
    data, y = gen.batch(B)
    pool.append(([torch.from_numpy(d).cuda() for d in data], torch.from_numpy(y).cuda()))
torch.manual_seed(0)
m = reviews4rec_amd.get_model_class(hp['model_type'])(hp)
xavier_init(m)
from reviews4rec_amd import engine as E
kw = dict(lr=hp['lr'], weight_decay=hp['weight_decay'], seed=4321)
eng = {'NARRE': E.NarreEngine, 'deepconn': E.DeepCoNNEngine}.get(hp['model_type'], E.TransNetEngine)(m.cuda().train(), **kw)
lib = ctypes.CDLL(os.environ['R4R_LIBRARY'])
setter = {'NARRE': lib.r4r_debug_narre_head_trace, 'deepconn': lib.r4r_debug_dc_head_trace}.get(hp['model_type'], lib.r4r_debug_tn_head_trace)
setter.argtypes = [ctypes.c_void_p]
backward = '--backward' in sys.argv
gather = '--gather' in sys.argv                           # stage timeline of proj_gather_max_kernel's workgroups instead
if gather:
    setter = lib.r4r_debug_gather_trace
    setter.argtypes = [ctypes.c_void_p]
if backward:
    # (narre_backward_kernel<0> is instantiated by the NARRE, DeepCoNN++ AND TransNet translation units; which copy a
    # launch runs is the linker's choice, and each reads its own unit's trace pointer: set both)
    names = ['r4r_debug_dc_bwd_trace'] if hp['model_type'] == 'deepconn' else [
        'r4r_debug_narre_bwd_trace', 'r4r_debug_tn_bwd_trace', 'r4r_debug_dcpp_bwd_trace']
    fns = [getattr(lib, n) for n in names]
    for f in fns:
        f.argtypes = [ctypes.c_void_p]
    setter = lambda p: max(f(p) for f in fns)
trace = torch.zeros(max(B * 32, 65536 * 4), dtype=torch.int64, device='cuda')
step = lambda i: eng.train_step(*pool[i % 4], next_data=pool[(i + 1) % 4][0])
for i in range(20):
    step(i)
torch.cuda.synchronize()
assert setter(ctypes.c_void_p(trace.data_ptr())) == 0
for i in range(20, 24):                                    # (the loop announces the next batch, as bench.py and main.train do)
    trace.zero_()
    step(i)
torch.cuda.synchronize()
if backward:
    tr = trace.cpu().numpy().reshape(-1, 4)
    tr[:, 3] = np.arange(len(tr))                            # linear workgroup index (x fastest, then y, then z)
    tr = tr[tr[:, 0] > 0]
    t0 = tr[:, 0].min()
    print('%s backward launch: %d workgroups, first start -> last end %.2f us' % (workload, len(tr), (tr[:, 1].max() - t0) / 100.0))
    for z in sorted(set(tr[:, 2].tolist())):
        r = tr[tr[:, 2] == z]
        d = (r[:, 1] - r[:, 0]) / 100.0
        print('z-slice %d: %4d workgroups, start %5.2f .. %5.2f us, duration med %5.2f max %5.2f, last end %5.2f us'
              % (z - 1, len(r), (r[:, 0].min() - t0) / 100.0, (r[:, 0].max() - t0) / 100.0, np.median(d), d.max(),
                 (r[:, 1].max() - t0) / 100.0))
        per = len(r)
        print('           longest: workgroup %d of the slice (%.2f us)' % (int(r[np.argmax(d), 3]) % per if per else -1, d.max()))
        busy = np.sort(d[d > 2.0])
        if len(busy) and z == 1 and hp['model_type'] == 'NARRE':   # the ID-table role: most workgroups exit at once
            print('           %d busy workgroups: p50 %.2f p90 %.2f p99 %.2f max %.2f us' % (
                len(busy), busy[len(busy) // 2], busy[int(len(busy) * 0.9)], busy[int(len(busy) * 0.99)], busy[-1]))
    sys.exit(0)
tr = trace.cpu().numpy().reshape(-1, 32)[:(8192 if gather else B)]
tr = tr[tr[:, 0] > 0]                                      # (DeepCoNN: one workgroup per 4 ratings)
n = int((tr[0, :20] > 0).sum())
t0 = tr[:, 0].min()
print('%s: B = %d,, %d stamps; workgroup starts spread %.2f us' % (workload, B, n, (tr[:, 0].max() - t0) / 100.0))
for k in range(1, n):
    d = (tr[:, k] - tr[:, k - 1]) / 100.0
    print('stage %2d: med %6.2f  max %6.2f us   (cum. med %6.2f)' % (k, np.median(d), d.max(), np.median(tr[:, k] - tr[:, 0]) / 100.0))
for k in range(20, 32):
    if tr[0, k] > 0:
        print('extra stamp %d: med %.2f us after start' % (k, np.median(tr[:, k] - tr[:, 0]) / 100.0))
print('whole: first start -> last end %.2f us' % ((tr[:, n - 1].max() - t0) / 100.0))
