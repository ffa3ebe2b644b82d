"""Host-side cost per step of the training loop's pieces (enqueue time, no device sync inside the loops): the loader's
batch iterator, the native engine's train_step, both through main.train -- per batch (hyper_params['spans'] = False) and
through the span entry points (the default).  python tools/host_probe.py [--model-type deepconn] [--embed 300]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
ap = argparse.ArgumentParser(); ap.add_argument('--model-type', default='deepconn'); ap.add_argument('--ratings', type=int, default=100000); ap.add_argument('--embed', type=int, default=300)
args = ap.parse_args()
import reviews4rec_amd
from reviews4rec_amd import main as M, synthetic
from reviews4rec_amd.data import DataLoader
from reviews4rec_amd.utils import xavier_init
from synth_reviews import synthesize
d = synthesize(args.ratings, 40000, 15000, 50002, test=1000)
hp = dict(model_type=args.model_type, batch_size=128, input_length=1000, narre_num_reviews=10, narre_num_words=100,
          total_users=40000, total_items=15000, latent_size=10, word_embed_size=args.embed, dropout=0.6, lr=0.002,
          weight_decay=1e-6, vocab=50002, total_words=50002, engine='native')
hp['word_vectors'] = synthetic.word_table(50002, args.embed)
train = DataLoader(hp, d['train'], d['user_reviews'], d['item_reviews'], None,
                   this_index_user_item=d['this_index_user_item'], device='cuda')
model = reviews4rec_amd.get_model_class(args.model_type)(hp); xavier_init(model); model = model.cuda().train()
engine = M.make_engine(hp, model)
import gc; gc.collect(); gc.freeze()
nb = len(train)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batches = [b for b in train.iter()]
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('loader only: host %.1f us per batch, with drain %.1f' % ((t1 - t0) * 1e6 / nb, (t2 - t0) * 1e6 / nb))
    keep = batches[:64]; del batches
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(nb):
        data, y = keep[i % 64]
        engine.train_step(data, y, n_global=128, next_data=keep[(i + 1) % 64][0])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('engine only: host %.1f us per step, with drain %.1f' % ((t1 - t0) * 1e6 / nb, (t2 - t0) * 1e6 / nb))
    for spans in (False, True):
        hp['spans'] = spans
        torch.cuda.synchronize(); t0 = time.perf_counter()
        M.train(model, None, None, train, hp, engine=engine)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print('main.train, %s: host %.1f us per step, with drain %.1f' % ('spans (K steps per host call)' if spans else 'per-batch loop',
                                                                           (t1 - t0) * 1e6 / nb, (t2 - t0) * 1e6 / nb))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
M.train(model, None, None, train, hp, engine=engine)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
