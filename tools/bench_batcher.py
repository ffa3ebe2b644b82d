"""Device-side batch construction (r4r_batch_build, reviews4rec_amd/data.py) on one MI355X: time per batch
and bytes moved, Amazon-shaped synthetic reviews (log-normal review lengths, Zipf users / items / words).

    python tools/bench_batcher.py [--ratings 300000] [--model-type deepconn|NARRE]

Prints one JSON line: microseconds per batch of 128 ratings (HIP events over 200 batches), the int64 bytes
a batch writes, the token bytes it reads, and the resulting GB/s -- the loader-side number DESIGN.md quotes
next to the preprocessed-epoch loader's PCIe-bound rate."""
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
    ap.add_argument('--ratings', type=int, default=300000)
    ap.add_argument('--users', type=int, default=40000)
    ap.add_argument('--items', type=int, default=15000)
    ap.add_argument('--vocab', type=int, default=50002)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--model-type', default='deepconn')
    args = ap.parse_args()
    from reviews4rec_amd.data import DataLoader
    from synth_reviews import synthesize
    t0 = time.time()
    d = synthesize(args.ratings, args.users, args.items, args.vocab)
    train, user_reviews, item_reviews, tiui = d['train'], d['user_reviews'], d['item_reviews'], d['this_index_user_item']
    hp = dict(model_type=args.model_type, batch_size=args.batch, input_length=1000, narre_num_reviews=10,
              narre_num_words=100, total_users=args.users, total_items=args.items)
    t1 = time.time()
    loader = DataLoader(hp, train, user_reviews, item_reviews, None, this_index_user_item=tiui, device='cuda')
    t2 = time.time()
    it = loader.iter()
    for _ in range(20):
        data, y = next(it)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    a.record()
    for _ in range(n):
        data, y = next(it)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000.0 / n
    out_bytes = sum(d.numel() * 8 for d in data[:5])
    tok_bytes = int(sum((d != 0).sum().item() for d in (data[0], data[3], data[4])) * 4)
    pool_mb = (loader.store.users.tok.nbytes + loader.store.items.tok.nbytes) / 1e6
    print(json.dumps({'kernel': 'batch_build_kernel', 'model_type': args.model_type, 'batch': args.batch,
                      'us_per_batch': round(us, 2), 'int64_bytes_written': out_bytes, 'token_bytes_read': tok_bytes,
                      'GBs': round((out_bytes + tok_bytes) / us / 1e3, 1), 'ratings': len(train),
                      'token_pools_MB': round(pool_mb, 1),
                      'padded_epoch_arrays_MB': round(len(train) * out_bytes / args.batch / 1e6, 1),
                      'host_s': {'synthesize': round(t1 - t0, 1), 'flatten_into_pools': round(t2 - t1, 1)}}))


if __name__ == '__main__':
    main()
