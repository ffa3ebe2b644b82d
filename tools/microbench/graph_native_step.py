"""Does a hipGraph of the fused native step beat launching its five kernels eagerly?  cfg3, B = 128, dropout 0 (the
step passes its dropout offset by value, so a captured step would repeat its masks): eager 105.0 us/step, graph replay
(two steps per graph) 106.9 us/step on one MI355X -- no: the launch queue already runs ahead of the GPU, and the
gaps between dependent kernels are the hardware's, not the host's.  DESIGN.md section 7."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, reviews4rec_amd
from reviews4rec_amd import synthetic
from reviews4rec_amd.engine import DeepCoNNEngine
from reviews4rec_amd.utils import xavier_init
hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', dropout=0.0)
hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
gen = synthetic.Generator(hp, seed=5)
data, y = gen.batch(128)
data = [torch.from_numpy(d).cuda() for d in data]; y = torch.from_numpy(y).cuda()
torch.manual_seed(0)
m = reviews4rec_amd.get_model_class('deepconn')(hp); xavier_init(m)
eng = DeepCoNNEngine(m.cuda().train(), conv_algo=2)
for _ in range(20): eng.train_step(data, y, next_data=data)
torch.cuda.synchronize()
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('eager  %.1f us/step' % timeit(lambda: eng.train_step(data, y, next_data=data)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): eng.train_step(data, y, next_data=data)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    eng.train_step(data, y, next_data=data)
    eng.train_step(data, y, next_data=data)
torch.cuda.synchronize()
print('graph  %.1f us/step (2 steps per replay)' % (timeit(g.replay) / 2))
