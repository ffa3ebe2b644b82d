"""One training epoch + its validation pass THROUGH THE HOST LOOP (main.train, eval.evaluate) on one MI355X:
ratings stream out of the reference-schema loader (data.DataLoader: batches built on the device from the token
pools), the native engine trains on them.  What bench.py's device-resident pool leaves out -- the Python
iterator, the batch-build launch, the epoch-end reads -- is in this number.

    python tools/bench_epoch.py [--model-type deepconn] [--ratings 200000] [--embed 300]

Prints one JSON line: train ratings/s over the second epoch (the first warms workspaces), the validation
pass beside it."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ratings', type=int, default=200000)
    ap.add_argument('--test', type=int, default=20000)
    ap.add_argument('--users', type=int, default=40000)
    ap.add_argument('--items', type=int, default=15000)
    ap.add_argument('--vocab', type=int, default=50002)
    ap.add_argument('--embed', type=int, default=300)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--model-type', default='deepconn')
    ap.add_argument('--spans', type=int, default=1, help='0: main.train iterates the loader and steps per batch')
    ap.add_argument('--epochs', type=int, default=3)
    args = ap.parse_args()
    import reviews4rec_amd
    from reviews4rec_amd import main as M, synthetic
    from reviews4rec_amd.data import DataLoader
    from reviews4rec_amd.eval import evaluate
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.utils import xavier_init
    from synth_reviews import synthesize
    d = synthesize(args.ratings, args.users, args.items, args.vocab, test=args.test)
    hp = dict(model_type=args.model_type, batch_size=args.batch, input_length=1000, narre_num_reviews=10,
              narre_num_words=100, total_users=args.users, total_items=args.items, latent_size=10,
              word_embed_size=args.embed, dropout=0.6, lr=0.002, weight_decay=1e-6, vocab=args.vocab,
              total_words=args.vocab, engine='native', spans=bool(args.spans))
    hp['word_vectors'] = synthetic.word_table(args.vocab, args.embed)
    train = DataLoader(hp, d['train'], d['user_reviews'], d['item_reviews'], None,
                       this_index_user_item=d['this_index_user_item'], device='cuda')
    val = DataLoader(hp, d['test'], d['user_reviews'], d['item_reviews'], None, test_reviews=d['test_reviews'],
                     train_loader=train, device='cuda')
    torch.manual_seed(0)
    model = reviews4rec_amd.get_model_class(args.model_type)(hp)
    xavier_init(model)
    model = model.cuda()
    engine = M.make_engine(hp, model)
    criterion = MSELoss(hp)
    review = args.model_type not in ('bias_only', 'MF', 'MF_dot', 'NeuMF')
    import gc
    gc.collect()
    gc.freeze()                                              # as main.train_complete does before its loop
    out = {'model_type': args.model_type, 'train_ratings': len(d['train']), 'val_ratings': len(d['test']),
           'batch': args.batch, 'embed': args.embed, 'spans': bool(args.spans)}
    for epoch in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        metrics = M.train(model, criterion, None, train, hp, engine=engine)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        vm, _, _ = evaluate(model, criterion, val, hp, dict(train.user_count), dict(train.item_count), review,
                            engine=engine)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out['epoch%d' % epoch] = {'train_s': round(t1 - t0, 3), 'train_ratings_per_s': round(len(d['train']) / (t1 - t0)),
                                  'us_per_step': round((t1 - t0) * 1e6 / len(train), 1), 'train_MSE': metrics['MSE'],
                                  'val_s': round(t2 - t1, 3), 'val_ratings_per_s': round(len(d['test']) / (t2 - t1)),
                                  'val_MSE': vm['MSE']}
    # the same engine on the same loader's batches held RESIDENT (what bench.py times): the loop's ceiling on this data
    keep = [b for _, b in zip(range(64), train.iter())]
    keep = [b for b in keep if b[1].shape[0] == args.batch]
    kw = {'defer_sweep': True} if getattr(engine, 'TEMPORAL_SWEEP', False) else {}
    model.train()
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nsteps = 4 * len(keep)
        for k in range(nsteps):
            data, y = keep[k % len(keep)]
            engine.train_step(data, y, n_global=args.batch, next_data=keep[(k + 1) % len(keep)][0], **kw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    out['resident'] = {'us_per_step': round((t1 - t0) * 1e6 / nsteps, 1),
                       'train_ratings_per_s': round(nsteps * args.batch / (t1 - t0))}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
