"""Accuracy of the opt-in fp16-split projection GEMM (csrc/project_f16.hip) against a float64 convolution, next to
the default fp32 GEMM's, through r4r_textcnn_fwd (r4r_gemm_math mode 2).  DESIGN.md 4.1d quotes it."""
import os, sys, math
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, torch.nn.functional as F
from reviews4rec_amd import _lib, ops
lib = _lib.lib()
torch.manual_seed(0)
for (N, T, E, V, tscale) in [(36, 1000, 300, 1500, 0.011), (64, 1000, 300, 30000, 0.011), (700, 100, 64, 2000, 1.0), (64, 1000, 300, 20000, 3.0)]:
    g = torch.Generator().manual_seed(N + T)
    table = (torch.rand((V, E), generator=g) * 2 - 1) * tscale
    table[5] *= 1e-4                                      # a row far below the maximum
    w = (torch.rand((100, 1, 3, E), generator=g) - 0.5) * (2 * math.sqrt(6.0 / (3 * E + 300 * E)))
    b = (torch.rand(100, generator=g) - 0.5) * 0.1
    zipf = torch.distributions.Categorical(probs=1.0 / torch.arange(1, V + 1).float())
    idx = zipf.sample((N, T))
    x = F.embedding(idx, table.double()).unsqueeze(1)
    y = F.relu(F.conv2d(x, w.double(), b.double(), padding=(2, 0))).squeeze(-1)
    ref = y.max(dim=2).values                              # float64 reference
    args = (idx.cuda(), table.cuda(), w.cuda(), b.cuda())
    lib.r4r_gemm_math(0, 0.0, 0.0)
    p32, a32 = ops.textcnn_fwd_raw(*args); p32 = p32.cpu().double(); a32 = a32.cpu()
    lib.r4r_gemm_math(2, float(table.abs().max()), float(w.abs().max()))
    p16, a16 = ops.textcnn_fwd_raw(*args); p16 = p16.cpu().double(); a16 = a16.cpu()
    lib.r4r_gemm_math(0, 0.0, 0.0)
    scale = ref.abs().max()
    e32 = (p32 - ref).abs().max() / scale; e16 = (p16 - ref).abs().max() / scale
    print('N %d T %d E %d V %d |table|<=%.3g: max err / max|ref|: fp32 %.2e  f16x2 %.2e  (f16x2 vs fp32 %.2e)  argmax equal %.5f'
          % (N, T, E, V, tscale, e32, e16, (p16 - p32).abs().max() / scale, (a16 == a32).float().mean()))
