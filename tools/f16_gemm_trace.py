"""Per-workgroup timeline of the opt-in fp16-split projection GEMM (csrc/project_f16.hip) on one MI355X:
`make -C reviews4rec_amd/csrc trace`, then `python tools/f16_gemm_trace.py` prints prologue / K loop /
epilogue per workgroup (s_memrealtime stamps under -DR4R_TRACE; never loaded by the product path)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['R4R_LIBRARY'] = os.path.join(ROOT, 'reviews4rec_amd/csrc', os.environ.get('TRACE_SO', 'libr4r_hip_trace.so'))
os.environ['R4R_GEMM_MATH'] = 'f16x2'
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
trace = torch.zeros(256 * 32, dtype=torch.int64, device='cuda')
for i in range(20):
    eng.train_step(*pool[i % 4])
torch.cuda.synchronize()
lib.r4r_debug_f16_gemm_trace.argtypes = [ctypes.c_void_p]
assert lib.r4r_debug_f16_gemm_trace(ctypes.c_void_p(trace.data_ptr())) == 0
for i in range(4):
    trace.zero_()
    eng.train_step(*pool[i % 4])
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(256, 32)
tr = tr[tr[:, 0] > 0]
t0 = tr[:, 0].min()
us = lambda x: (x - t0) / 100.0
print('%d workgroups; starts spread %.2f us' % (len(tr), us(tr[:, 0].max())))
for name, a, b in (('prologue', 0, 1), ('K loop', 1, 2), ('epilogue', 2, 3)):
    d = (tr[:, b] - tr[:, a]) / 100.0
    print('%-9s med %6.2f  min %6.2f  max %6.2f us' % (name, np.median(d), d.min(), d.max()))
print('last workgroup done at %.2f us' % us(tr[:, 3].max()))
