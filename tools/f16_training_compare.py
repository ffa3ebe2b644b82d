"""Does the opt-in fp16-split projection GEMM change TRAINING?  cfg3 shapes, dropout 0, the same initial weights
and the same 2,000 batches through three arithmetic variants of the same native step:

    project   fp32 projection GEMM (the default)
    direct    the gather-fused direct conv (fp32 MFMA, another summation order: the yardstick for fp32 noise)
    f16x2     fp16-split projection GEMM (R4R_GEMM_MATH=f16x2)

Prints the running train MSE per window of 250 steps for each, and the relative distance of the final weights
of `direct` and `f16x2` from `project`'s: if the opt-in arithmetic is fp32-grade, it sits at the distance the
fp32 reordering does.  DESIGN.md 4.1d quotes the output (profiles/r02k_f16_training.txt)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS, WINDOW, B = int(os.environ.get('STEPS', 2000)), 250, 128


def run(variant):
    os.environ['R4R_GEMM_MATH'] = 'f16x2' if variant == 'f16x2' else 'f32'
    import reviews4rec_amd
    from reviews4rec_amd import engine as E, synthetic
    from reviews4rec_amd.utils import xavier_init
    E._MATH_OWNER[0] = None
    hp = synthetic.hyper_params_for('cfg3_deepconn_electronics_e300', dropout=0.0)
    hp['word_vectors'] = synthetic.word_table(hp['vocab'], hp['word_embed_size'])
    gen = synthetic.Generator(hp, seed=11)
    torch.manual_seed(99)
    model = reviews4rec_amd.get_model_class('deepconn')(hp)
    xavier_init(model)
    eng = E.DeepCoNNEngine(model.cuda().train(), lr=hp['lr'], weight_decay=hp['weight_decay'],
                           conv_algo=1 if variant == 'direct' else 2)
    windows, prev = [], 0.0
    for step in range(STEPS):
        data, y = gen.batch(B)
        eng.train_step([torch.from_numpy(d).cuda() for d in data], torch.from_numpy(y).cuda())
        if (step + 1) % WINDOW == 0:
            tot = float(eng.sse[0].item())
            windows.append((tot - prev) / (WINDOW * B))
            prev = tot
    E.apply_gemm_math(eng.table, eng._conv_weights()) if variant == 'f16x2' else None
    return windows, eng.flat_p.detach().double().cpu()


def main():
    out = {v: run(v) for v in ('project', 'direct', 'f16x2')}
    os.environ['R4R_GEMM_MATH'] = 'f32'
    print('train MSE per window of %d steps (B = %d, dropout 0, same init, same batches):' % (WINDOW, B))
    for v, (w, _) in out.items():
        print('%-8s %s' % (v, ' '.join('%.5f' % x for x in w)))
    ref = out['project'][1]
    for v in ('direct', 'f16x2'):
        d = out[v][1] - ref
        print('%-8s final dense weights vs project: |d|_2 / |w|_2 = %.3e, max |d| = %.3e'
              % (v, float(d.norm() / ref.norm()), float(d.abs().max())))


if __name__ == '__main__':
    main()
