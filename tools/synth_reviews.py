"""An Amazon-shaped synthetic dataset in the reference's pickled schema (data.load_data's inputs):
Zipf users / items / words, log-normal review lengths.  Shared by tools/bench_batcher.py and
tools/bench_eval.py; nothing in the product imports it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthesize(ratings=300000, users=40000, items=15000, vocab=50002, test=0, rank_users=0, seed=7):
    """-> dict(train, test, user_reviews, item_reviews, this_index_user_item, test_reviews, negs).
    `test` held-out ratings (their reviews in test_reviews, as preprocess_random_split.py:226-238 leaves
    them); `rank_users` users with one held-out positive and five negatives each (negs[u] = [[pos], [5]])."""
    from reviews4rec_amd import synthetic
    rng = np.random.default_rng(seed)
    draw_u = synthetic._zipf_sampler(users, 1.1, rng)
    draw_i = synthetic._zipf_sampler(items, 1.1, rng)
    draw_w = synthetic._zipf_sampler(vocab - 1, 1.0, rng)
    want = ratings + test
    seen, pairs = set(), []
    while len(pairs) < want:                                 # (Zipf draws repeat: keep drawing)
        for u, i in zip(draw_u((want,)).tolist(), draw_i((want,)).tolist()):
            if (u, i) not in seen:
                seen.add((u, i))
                pairs.append([u, i, float(rng.integers(1, 6))])
                if len(pairs) == want:
                    break
    train, held = pairs[:ratings], pairs[ratings:]
    lens = np.minimum(400, rng.lognormal(np.log(60), 0.9, size=len(pairs))).astype(np.int64).clip(min=1)
    toks = (draw_w((int(lens.sum()),)) + 1).astype(np.int64)
    cuts = np.concatenate([[0], np.cumsum(lens)])
    user_reviews = {u: [] for u in range(users)}
    item_reviews = {i: [] for i in range(items)}
    tiui, test_reviews = {}, {}
    for n, (u, i, r) in enumerate(train):
        rev = toks[cuts[n]:cuts[n + 1]].tolist()
        tiui.setdefault(u, {})[i] = [len(user_reviews[u]), len(item_reviews[i])]
        user_reviews[u].append(rev)
        item_reviews[i].append(rev)
    for n, (u, i, r) in enumerate(held, start=len(train)):
        test_reviews.setdefault(u, {})[i] = toks[cuts[n]:cuts[n + 1]].tolist()
    negs = {}
    for u, i, r in held:
        if len(negs) == rank_users:
            break
        if u not in negs:
            negs[u] = [[i], rng.integers(0, items, size=5).tolist()]
    return dict(train=train, test=held, user_reviews=user_reviews, item_reviews=item_reviews,
                this_index_user_item=tiui, test_reviews=test_reviews, negs=negs)
