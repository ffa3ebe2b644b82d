"""Yardstick for the projection GEMM's shape: what the vendor's tuned fp32 GEMM (rocBLAS / hipBLASLt behind
torch.mm) sustains at M = distinct rows of a cfg3 batch, N = 304 | 300, K = 300 on this MI355X -- a dense GEMM on
contiguous operands, i.e. without the row gather the product kernel fuses.  Not part of the product path.
    python tools/microbench/sgemm_yardstick.py            (run on the GPU box)"""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = 'cuda'


def run(M, N, K, iters=300):
    a = (torch.rand(M, K, device=dev) - 0.5) * 0.02
    w = (torch.rand(N, K, device=dev) - 0.5) * 0.2
    out = torch.empty(M, N, device=dev)
    for _ in range(30):
        torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            torch.mm(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    fl = 2.0 * M * N * K
    print('sgemm M %6d N %3d K %3d: %7.2f us  %6.1f TFLOP/s (%.3f of 157.3)' % (M, N, K, best * 1000, fl / best / 1e9, fl / best / 1e9 / 157.3))


for M in (29547, 29568, 32768, 14784):
    for N in (304, 300, 320):
        run(M, N, 300)
run(29547, 304, 304)
run(29568, 320, 320)
