"""VERDICT r5 next #6 -- "pipeline the towers: GEMM(user) -> [gather(user) || GEMM(item)] -> gather(item)": what the two
halves of that pipeline cost when each tower's projection GEMM and gather are a launch of their own, against the one
launch over both towers the step runs today.  The headline's plan (cfg3, B = 128, the bench's own pool batches, the
padded E = 304 table), timed with the library's HIP-event brackets around the GEMM and the gather kernel.

    python tools/r06_tower_split.py        (on the MI355X box; prints one JSON line)"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['R4R_CONV_ALGO'] = 'project'

from reviews4rec_amd import _lib, synthetic           # noqa: E402
from reviews4rec_amd.engine import padded_word_table  # noqa: E402


def main():
    dev = torch.device('cuda')
    lib = _lib.lib()
    hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300')
    table = padded_word_table(torch.from_numpy(synthetic.word_table(hp['vocab'], hp['word_embed_size'])).to(dev))
    V, E = table.shape
    gen = synthetic.Generator(hp, seed=synthetic.SEED)
    pool = [gen.batch(128)[0] for _ in range(8)]
    docs = [(torch.from_numpy(d[3]).to(dev), torch.from_numpy(d[4]).to(dev)) for d in pool]
    both = [torch.cat(p) for p in docs]                     # [256, T]: the two towers' documents as ONE launch's rows
    g = torch.Generator(device='cpu').manual_seed(0)
    conv_w = (torch.rand(100, 3, E, generator=g) * 0.1 - 0.05).to(dev)
    conv_b = torch.zeros(100, device=dev)
    T = docs[0][0].shape[1]

    def run(idx_list, reps=40):
        N = idx_list[0].shape[0]
        ws = torch.zeros(lib.r4r_textcnn_ws_bytes(N, T, E, 100, V), dtype=torch.uint8, device=dev)
        pooled = torch.empty(N, 100, device=dev)
        argmax = torch.empty(N, 100, dtype=torch.int32, device=dev)

        def once(idx):
            _lib.check(lib.r4r_textcnn_fwd(table.data_ptr(), V, idx.data_ptr(), conv_w.data_ptr(), conv_b.data_ptr(),
                                           pooled.data_ptr(), argmax.data_ptr(), ws.data_ptr(), ws.numel(), N, T, E, 100,
                                           _lib.current_stream()), 'r4r_textcnn_fwd')
        for k in range(20):
            once(idx_list[k % len(idx_list)])
        torch.cuda.synchronize()
        lib.r4r_timing_read(0, ctypes.byref(ctypes.c_double()), ctypes.byref(ctypes.c_int64()), 1)
        lib.r4r_timing_enable((1 << 3) | (1 << 4))
        for k in range(reps):
            once(idx_list[k % len(idx_list)])
        torch.cuda.synchronize()
        lib.r4r_timing_enable(0)
        out = {}
        for name, slot in (('gemm_us', 3), ('gather_us', 4)):
            tot, cnt = ctypes.c_double(), ctypes.c_int64()
            lib.r4r_timing_read(slot, ctypes.byref(tot), ctypes.byref(cnt), 0)
            out[name] = round(1000.0 * tot.value / max(1, cnt.value), 2)
        lib.r4r_timing_read(0, ctypes.byref(ctypes.c_double()), ctypes.byref(ctypes.c_int64()), 1)
        out['rows'] = int(np.mean([int(i.unique().numel()) for i in idx_list]))
        return out

    res = {'user_tower_alone': run([u for u, _ in docs]), 'item_tower_alone': run([i for _, i in docs]),
           'both_towers_one_launch_as_one_row_set': run(both)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
