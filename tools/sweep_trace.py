"""Role timeline of mf_adam_kernel (the ID-table Adam launch of MF_dot / TransNet++ / MF / NeuMF) on one MI355X.

`make -C reviews4rec_amd/csrc trace` first (s_memrealtime stamps at the start and end of every workgroup under
-DR4R_TRACE; never loaded by the product path), then `python tools/sweep_trace.py cfg2_mfdot_electronics` prints, per
role (table chunks, bias-vector chunks, the global-bias workgroup, entry waves), when its workgroups start and how
long they live."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['R4R_LIBRARY'] = os.path.join(ROOT, 'reviews4rec_amd/csrc', os.environ.get('TRACE_SO', 'libr4r_hip_trace.so'))
import torch
import reviews4rec_amd
from reviews4rec_amd import synthetic
from reviews4rec_amd.utils import xavier_init
from reviews4rec_amd import main as M

workload = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_mfdot_electronics'
hp = synthetic.hyper_params_for(workload, dropout=0.6)
B = hp['batch_size']
if 'vocab' in hp and hp.get('word_embed_size'):
    hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
gen = synthetic.Generator(hp, seed=5)
pool = []
for _ in range(8):
    data, y = gen.batch(B)
    pool.append(([None if d is None else torch.from_numpy(d).cuda() for d in data], torch.from_numpy(y).cuda()))
torch.manual_seed(0)
m = reviews4rec_amd.get_model_class(hp['model_type'])(hp)
xavier_init(m)
eng = M.make_engine(hp, m.cuda().train())
lib = ctypes.CDLL(os.environ['R4R_LIBRARY'])
lib.r4r_debug_mf_adam_trace.argtypes = [ctypes.c_void_p]
trace = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
for i in range(40):
    eng.train_step(*pool[i % 8], defer_sweep=True)
torch.cuda.synchronize()
assert lib.r4r_debug_mf_adam_trace(ctypes.c_void_p(trace.data_ptr())) == 0
names = {1: 'table chunks', 2: 'bias-vector chunks', 3: 'global bias', 4: 'entry waves'}
for i in range(3):
    trace.zero_()
    eng.train_step(*pool[i % 8], defer_sweep=True)
    torch.cuda.synchronize()
    tr = trace.cpu().numpy().reshape(-1, 4)
    tr = tr[tr[:, 0] > 0]
    t0 = tr[:, 0].min()
    print('%s step %d: %d workgroups, first start -> last end %.2f us' % (workload, i, len(tr), (tr[:, 1].max() - t0) / 100.0))
    # how many workgroups are resident at once (every 2 us of the launch)
    st_all, en_all = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0
    print('  resident workgroups at t = 1, 3, 5, ... us: ' + ' '.join(
        '%d' % int(((st_all <= t) & (en_all > t)).sum()) for t in np.arange(1.0, en_all.max(), 2.0)))
    for z in sorted(set(tr[:, 2].tolist())):
        r = tr[tr[:, 2] == z]
        d = (r[:, 1] - r[:, 0]) / 100.0
        st = (r[:, 0] - t0) / 100.0
        print('  %-20s %5d workgroups, starts %5.2f .. %5.2f us (median %5.2f), life med %5.2f p90 %5.2f max %5.2f, last end %5.2f us'
              % (names.get(int(z) - 1, str(z)), len(r), st.min(), st.max(), np.median(st), np.median(d),
                 np.sort(d)[int(len(d) * 0.9)], d.max(), (r[:, 1].max() - t0) / 100.0))
