import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import torch
import reviews4rec_amd
from reviews4rec_amd import synthetic
from reviews4rec_amd.engine import DeepCoNNEngine
from reviews4rec_amd.utils import xavier_init
hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', dropout=0.6)
hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
gen = synthetic.Generator(hp, seed=5)
pool = []
for _ in range(8):
    data, y = gen.batch(128)
    pool.append(([torch.from_numpy(d).cuda() for d in data], torch.from_numpy(y).cuda()))
torch.manual_seed(0)
m = reviews4rec_amd.get_model_class('deepconn')(hp); xavier_init(m)
eng = DeepCoNNEngine(m.cuda().train())
def step(i):
    d, y = pool[i % 8]; nd = pool[(i + 1) % 8]
    eng.train_step(d, y, next_data=nd[0])
for i in range(40): step(i)
for trial, gap in enumerate([0.0, 0.0, 0.001, 0.01]):
    torch.cuda.synchronize(); time.sleep(gap)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(40):
        step(40 + i); evs[i + 1].record()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    d = [evs[i].elapsed_time(evs[i + 1]) * 1000 for i in range(40)]
    print('gap %.3fs host %.1f us/step; per-step us:' % (gap, (t1 - t0) / 40 * 1e6), ' '.join('%.0f' % x for x in d))
