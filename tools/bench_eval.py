"""Validation throughput on one MI355X: `eval.evaluate` (MSE + the two count -> [SE] maps)
over a held-out split and `eval.eval_ranking` (HR@1 over [B, 6] candidate rows), batches built on the device
from the token pools, scored by the fused eval forward of the native engine -- and, beside it, by the
op-by-op HIP module path.

    python tools/bench_eval.py [--model-type deepconn|NARRE|MF_dot|...] [--test 20000] [--rank-users 2000]

Prints one JSON line: ratings/s of evaluate(), candidate rows/s of eval_ranking(), and the seconds the
reference-shaped host work (the maps, the top-1) takes inside them."""
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
    ap.add_argument('--rank-users', type=int, default=2000)
    ap.add_argument('--users', type=int, default=40000)
    ap.add_argument('--items', type=int, default=15000)
    ap.add_argument('--vocab', type=int, default=50002)
    ap.add_argument('--embed', type=int, default=300)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--model-type', default='deepconn')
    args = ap.parse_args()
    import reviews4rec_amd
    from reviews4rec_amd import synthetic
    from reviews4rec_amd.data import DataLoader
    from reviews4rec_amd.eval import evaluate, eval_ranking
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.main import make_engine
    from reviews4rec_amd.utils import xavier_init
    from synth_reviews import synthesize
    d = synthesize(args.ratings, args.users, args.items, args.vocab, test=args.test, rank_users=args.rank_users)
    hp = dict(model_type=args.model_type, batch_size=args.batch, input_length=1000, narre_num_reviews=10,
              narre_num_words=100, total_users=args.users, total_items=args.items, latent_size=10,
              word_embed_size=args.embed, dropout=0.6, lr=0.002, weight_decay=1e-6, vocab=args.vocab,
              total_words=args.vocab, engine='native')
    hp['word_vectors'] = synthetic.word_table(args.vocab, args.embed)
    train = DataLoader(hp, d['train'], d['user_reviews'], d['item_reviews'], d['negs'],
                       this_index_user_item=d['this_index_user_item'], device='cuda')
    test = DataLoader(hp, d['test'], d['user_reviews'], d['item_reviews'], d['negs'], test_reviews=d['test_reviews'],
                      train_loader=train, device='cuda')
    torch.manual_seed(0)
    model = reviews4rec_amd.get_model_class(args.model_type)(hp)
    xavier_init(model)
    model = model.cuda()
    engine = make_engine(hp, model)
    criterion = MSELoss(hp)
    review = args.model_type not in ('bias_only', 'MF', 'MF_dot', 'NeuMF')
    out = {'model_type': args.model_type, 'test_ratings': len(d['test']), 'ranking_rows': len(d['negs']),
           'batch': args.batch, 'embed': args.embed}

    def timed(fn):
        fn()                                                 # warm: workspaces, token pools on the device
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    for label, eng in (('native', engine), ('module', None)):
        (m, umap, imap), t = timed(lambda: evaluate(model, criterion, test, hp, dict(train.user_count),
                                                    dict(train.item_count), review, engine=eng))
        hr, t2 = timed(lambda: eval_ranking(model, test, hp, review=review, engine=eng))
        out[label] = {'evaluate_ratings_per_s': round(len(d['test']) / t), 'evaluate_s': round(t, 3), 'MSE': m['MSE'],
                      'eval_ranking_candidates_per_s': round(6 * len(d['negs']) / t2), 'eval_ranking_s': round(t2, 3),
                      'HR@1': hr['HR@1']}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
