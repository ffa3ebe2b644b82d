"""Per-workgroup timeline of the projection GEMM on one MI355X.

Build the instrumented library first: `make -C reviews4rec_amd/csrc trace` (adds s_memrealtime
stamps + HW_ID per workgroup under -DR4R_TRACE; never loaded by the product path), then
`python tools/gemm_trace.py` prints start / operands-staged / loop-done / end per workgroup, the
workgroups-per-CU histogram and the busy span of each CU.  DESIGN.md section 4.1b quotes it."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['R4R_LIBRARY'] = os.path.join(ROOT, 'reviews4rec_amd/csrc', os.environ.get('TRACE_SO', 'libr4r_hip_trace.so'))
import torch
import reviews4rec_amd
from reviews4rec_amd import synthetic
from reviews4rec_amd.engine import DeepCoNNEngine
from reviews4rec_amd.utils import xavier_init

B = int(os.environ.get('B', 128))
hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', dropout=0.6)
hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
gen = synthetic.Generator(hp, seed=5)
pool = []
for _ in range(4):
    data, y = gen.batch(B)
    pool.append(([torch.from_numpy(d).cuda() for d in data], torch.from_numpy(y).cuda()))
torch.manual_seed(0)
m = reviews4rec_amd.get_model_class('deepconn')(hp)
xavier_init(m)
eng = DeepCoNNEngine(m.cuda().train(), conv_algo=2)
lib = ctypes.CDLL(os.environ['R4R_LIBRARY'])
NWG = 4096
trace = torch.zeros(NWG * 8, dtype=torch.int64, device='cuda')
for i in range(20):
    eng.train_step(*pool[i % 4])
torch.cuda.synchronize()
lib.r4r_debug_trace.argtypes = [ctypes.c_void_p]
assert lib.r4r_debug_trace(ctypes.c_void_p(trace.data_ptr())) == 0
for i in range(6):
    trace.zero_()
    eng.train_step(*pool[i % 4])
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(NWG, 8)
launched = tr[:, 0] > 0
act = launched & ((tr[:, 6] & 1) == 1)
t0 = tr[launched, 0].min()
us = lambda x: (x - t0) / 100.0
print('workgroups launched %d, active %d' % (launched.sum(), act.sum()))
print('all WG starts: first 0, last %.2f us' % us(tr[launched, 0].max()))
a = tr[act]
st, staged, loop, end = us(a[:, 0]), us(a[:, 1]), us(a[:, 2]), us(a[:, 3])
print('active start: min %.2f med %.2f max %.2f' % (st.min(), np.median(st), st.max()))
print('prologue (start->staged): med %.2f max %.2f' % (np.median(staged - st), (staged - st).max()))
print('loop: min %.2f med %.2f max %.2f' % ((loop - staged).min(), np.median(loop - staged), (loop - staged).max()))
print('epilogue: med %.2f max %.2f' % (np.median(end - loop), (end - loop).max()))
print('active end: min %.2f med %.2f max %.2f' % (end.min(), np.median(end), end.max()))
w7 = a[:, 7]
if (a[:, 6] >> 1).any():                                   # A-resident form: word 7 = cycles of pass 1, word 6 >> 1 = of pass 2
    m1, m2 = w7 / (staged - st), (a[:, 6] >> 1) / (loop - staged)
    print('A-resident form: pass 1 (start->stamp 1) clock med %.0f MHz, cycles med %d; pass 2 clock med %.0f MHz, cycles med %d'
          % (np.median(m1), np.median(w7), np.median(m2), np.median(a[:, 6] >> 1)))
elif w7.any():
    mhz = w7 / (loop - staged)
    print('shader clock over the loop: med %.0f MHz (min %.0f, max %.0f); cycles med %d' % (np.median(mhz), mhz.min(), mhz.max(), np.median(w7)))
hw = a[:, 4]; xcc = a[:, 5] & 0xf
if (a[:, 5] >> 8).any():
    pro = ((a[:, 5] >> 8) - t0) / 100.0 - st
    print('A-resident form: prologue of pass 1 (start -> first operands read): med %.2f max %.2f us' % (np.median(pro), pro.max()))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
ids, cnt = np.unique(cuid, return_counts=True)
print('distinct CUs used by active WGs: %d; WGs per CU histogram:' % len(ids), np.bincount(cnt))
print('per XCC active WGs:', np.bincount(xcc.astype(int), minlength=8))
print('per XCC end (med / max):', ' '.join('%.1f/%.1f' % (np.median(end[xcc == x]), end[xcc == x].max()) for x in range(8) if (xcc == x).any()))
print('per XCC loop (med):', ' '.join('%.1f' % np.median((loop - staged)[xcc == x]) for x in range(8) if (xcc == x).any()))
# concurrency per CU over time: how many active WGs share a CU at each WG's midpoint
order = np.argsort(st)
for k in list(range(0, len(order), max(1, len(order) // 24))):
    i = order[k]
    print('  wg#%4d xcc %d cu %5d start %6.2f staged %6.2f loopdone %6.2f end %6.2f' % (i, xcc[i], cuid[i], st[i], staged[i], loop[i], end[i]))
# per-CU busy union
span = []
for c in ids:
    sel = cuid == c
    span.append((st[sel].min(), end[sel].max(), sel.sum()))
span = np.array(span)
print('per-CU: first start med %.2f, last end med %.2f / max %.2f' % (np.median(span[:, 0]), np.median(span[:, 1]), span[:, 1].max()))
for n in sorted(set(cnt)):
    s = span[span[:, 2] == n]
    print('   CUs with %d WGs: %d, busy span med %.2f us, last end med %.2f' % (n, len(s), np.median(s[:, 1] - s[:, 0]), np.median(s[:, 1])))
