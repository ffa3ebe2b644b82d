import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reviews4rec_amd import ops, synthetic, _lib
B = int(os.environ.get('B', 128)); T = 1000; E = int(os.environ.get('E', 300)); V = 50002
hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', word_embed_size=E)
gen = synthetic.Generator(hp)
data, y = gen.batch(B)
idx = torch.from_numpy(data[3]).cuda()
table = torch.from_numpy(synthetic.word_table(V, E)).cuda()
torch.manual_seed(0)
w = (torch.rand((100, 1, 3, E), device='cuda') - 0.5) * 0.1
b = (torch.rand(100, device='cuda') - 0.5) * 0.1
pooled, arg = ops.textcnn_fwd_raw(idx, table, w, b)
torch.cuda.synchronize()
lib = _lib.lib()
for rep in range(3):
    lib.r4r_timing_enable(0xff)
    for _ in range(20):
        ops.textcnn_fwd_raw(idx, table, w, b)
    torch.cuda.synchronize()
    lib.r4r_timing_enable(0)
    tot, cnt = ctypes.c_double(), ctypes.c_int64()
    lib.r4r_timing_read(0, ctypes.byref(tot), ctypes.byref(cnt), 1)
    ms = tot.value / cnt.value
    fl = B * (T + 2) * 100 * 3 * E * 2
    print('variant', os.environ.get('R4R_TEXTCNN_FWD', 'default'), 'B', B, 'E', E, 'ms %.4f' % ms, 'TF %.2f' % (fl / ms / 1e9), 'frac %.3f' % (fl / ms / 1e9 / 157.3))
print('checksum', float(pooled.sum()), int(arg.sum()))
