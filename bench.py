#!/usr/bin/env python3
"""Headline benchmark: train ratings/s, DeepCoNN (word2vec-300, 100 conv filters) on
Amazon-Electronics-shaped synthetic interactions (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches this file under torch.distributed.run, one rank per
GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env); ranks run
data-parallel with one RCCL all-reduce of the dense gradients per step.

A "step" is one full training step -- zero_grad, forward (word gather + TextCNN x2
+ FC + dropout + FM), per-example SE, mean, backward, fused Adam -- over one batch
of `--batch-per-gpu` ratings per GPU whose index tensors are already resident in
HBM.  Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the
default step -- proj_gemm_kernel, the fp32-MFMA projection of the batch's distinct
tokens (project-then-gather): useful flops of the batches actually
run / the kernel's mean duration, measured with HIP events on the launch stream
inside the timed region (textcnn_fwd_kernel takes its place with --conv-algo direct).
`roofline_gather` is the second leg (proj_gather_max_kernel; its operands are L2 /
Infinity-Cache resident, so it is reported against that, not against HBM).
`cpu_baseline` times the plain-PyTorch CPU oracle (same graph as the reference) on a
bounded sample of the same workload, on rank 0 at N=1 only.

Scaling: the default is WEAK (`--batch-per-gpu` ratings per rank, the reference's
batch_size 128, so N GPUs train 128 N ratings per step).  `--scaling strong
--global-batch G` shards a fixed global batch of G ratings instead.  A weak run at
N > 1 appends a second, separately timed STRONG measurement at G = 1024
(`"strong": {...}`) so one SCALE sweep yields both curves, and every N > 1 run ends
with a replica check: the ranks' parameter buffers must be bit-identical.

Data dependence: project-then-gather's work follows the DISTINCT tokens of a batch.
`--doc-fill full` (no zero padding) and `--token-dist uniform` (no Zipf head) are the
unfriendly points; the value line always uses the SURVEY 8d defaults.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA (no sparsity)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured achievable)
PEAK_L2_GBS = 34500.0             # MI355X_MICROARCH.md: L2, 4 MiB per XCD, aggregate over the 8 XCDs
WORKLOAD = 'cfg3_deepconn_electronics_e300'


# rocprofv3 --pmc summaries (profiles/, made by tools/r06_profiles.sh: separate counter passes of the same bench command)
# by (workload, ratings per rank, doc_fill, token_dist)
PMC_SUMMARIES = {
    (WORKLOAD, 128, 'lognormal', 'zipf'): 'r06_bench_pmc_summary.json',
    ('cfg1_bias_only_musical', 128, 'lognormal', 'zipf'): 'r06_bench_cfg1_pmc_summary.json',
    ('cfg2_mfdot_electronics', 128, 'lognormal', 'zipf'): 'r06_bench_cfg2_pmc_summary.json',
    ('cfg2_mfdot_electronics', 8192, 'lognormal', 'zipf'): 'r06_bench_cfg2_b8192_pmc_summary.json',
    ('cfg4_narre_kindle', 128, 'lognormal', 'zipf'): 'r06_bench_cfg4_pmc_summary.json',
    ('cfg5_transnetpp_synthetic', 128, 'lognormal', 'zipf'): 'r06_bench_cfg5_pmc_summary.json',
    (WORKLOAD, 128, 'full', 'uniform'): 'r06_bench_cfg3_fullunif_pmc_summary.json',
    # the one true HBM gather: full-length documents of uniformly drawn words at a 1 M-word vocabulary, projection pinned
    ('cfg5_transnetpp_synthetic', 128, 'full', 'uniform'): 'r06_cfg5_fullunif_pmc_summary.json',
}


def measured_traffic(kernel, args, live_launch_s=None):
    """(HBM-side bytes per launch of `kernel`, the file they come from) out of the committed rocprofv3
    --pmc summary (profiles/: separate counter passes of this same command, corrected as
    MI355X_MICROARCH.md prescribes) -- only for the configuration those passes ran, and only while the
    summary still describes this build: if the kernel duration recorded with the counters is more than
    35 % away from the one measured live in this run, the summary is stale and no traffic is reported."""
    name = PMC_SUMMARIES.get((args.workload, args.batch_per_gpu, args.doc_fill, args.token_dist))
    path = os.path.join(ROOT, 'profiles', name) if name else None
    pinned_ok = args.conv_algo == 'auto' or (args.conv_algo == 'project' and args.doc_fill == 'full'
                                             and args.workload == 'cfg5_transnetpp_synthetic')
    if not (path and os.path.exists(path) and args.engine == 'native' and pinned_ok and not args.model_type
            and not args.embed and args.scaling == 'weak' and args.gemm_math == 'f32'):
        return None, None
    kernels = json.load(open(path))['kernels']
    # (template instantiations carry a <..> suffix, and a run may hold two of them -- a flush launch of another form:
    # the one whose duration under the counters is closest to the live one is this launch)
    same = [v for k, v in kernels.items() if k.startswith('r4r::' + kernel) and v.get('avg_duration_us_under_pmc')]
    if not same:
        return None, None
    # ... the step's own launch is the instantiation launched most often (the one-off all-chunks flush of a blocked sweep
    # runs once per counter pass); among equally frequent ones, the closest in duration
    most = max(v.get('launches_under_pmc', 0) for v in same)
    same = [v for v in same if v.get('launches_under_pmc', 0) * 2 >= most]
    v = min(same, key=lambda v: abs(v['avg_duration_us_under_pmc'] * 1e-6 - (live_launch_s or 0.0)))
    then = v['avg_duration_us_under_pmc']
    # (a kernel runs 5-20 % longer under the counter passes than in the timed region: the band is wide enough for that)
    if live_launch_s and abs(then * 1e-6 - live_launch_s) > 0.35 * live_launch_s:
        return None, 'profiles/%s is stale for %s (%.1f us then, %.1f us now)' % (name, kernel, then, live_launch_s * 1e6)
    return v.get('hbm_bytes_per_launch'), 'profiles/' + name


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch-per-gpu', type=int, default=128, help='hyper_params batch_size per rank')
    ap.add_argument('--workload', default=WORKLOAD)
    ap.add_argument('--dropout', type=float, default=0.6, help='reference default (hyper_params.py:66)')
    ap.add_argument('--pool', type=int, default=8, help='distinct resident batches cycled through')
    ap.add_argument('--from-host', action='store_true',
                    help='feed the steps from pinned HOST arrays through data_fast.DataLoader (double-buffered H2D on a '
                         'copy stream) instead of HBM-resident batches: the PCIe-inclusive rate, for DESIGN.md only')
    ap.add_argument('--model-type', default=None, help="override the workload's recommender (e.g. deepconn++ on the "
                    "cfg3 shapes)")
    ap.add_argument('--embed', type=int, default=None, help='override word_embed_size (crossover experiments)')
    ap.add_argument('--latent', type=int, default=None, help='override latent_size')
    ap.add_argument('--neumf-stage', default=None, help="with --model-type NeuMF: GMF / MLP / NeuMF")
    ap.add_argument('--cpu-seconds', type=float, default=12.0,
                    help='budget of each half (thread calibration, measurement) of the cpu_baseline leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--step-events', action='store_true', help='diagnosis: a HIP event after every timed step, spans on stderr')
    ap.add_argument('--no-config-legs', action='store_true',
                    help='N = 1 default line: skip the separately timed legs of the other BASELINE.json configurations '
                         '(--no-cpu-baseline, the A/B scripts\' quick mode, skips them too unless --config-legs is given)')
    ap.add_argument('--config-legs', action='store_true', help='time the configuration legs even with --no-cpu-baseline')
    ap.add_argument('--leg-steps', type=int, default=200, help='timed steps of each configuration leg')
    ap.add_argument('--no-steady-leg', action='store_true',
                    help='short regions (--steps < 50): skip the appended 200-step steady leg')
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='leave the HIP-event kernel timing off (to measure what the instrumentation costs)')
    ap.add_argument('--conv-algo', choices=['auto', 'direct', 'project'], default='auto',
                    help='native engine: direct gather-fused MFMA conv, or projection GEMM over the distinct '
                         'tokens + gather-add-max (include/r4r.h R4R_CONV_*)')
    ap.add_argument('--token-prefetch', choices=['fused', 'side-stream', 'off'], default='fused',
                    help='native engine: token marks / compaction of batch k+1 prepared during step k -- on step '
                         'k\'s backward and gradient-reduce launches (fused), on a side stream, or not at all')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='weak: --batch-per-gpu ratings per rank (default); strong: --global-batch ratings per step '
                         'sharded over the ranks')
    ap.add_argument('--global-batch', type=int, default=1024, help='--scaling strong: ratings per step over all ranks')
    ap.add_argument('--strong-leg', default='1024,8192,32768',
                    help='weak runs at N > 1 also time strong-scaling legs at these global batch sizes '
                         "(comma list; '' skips): the fixed global batch is sharded over the ranks")
    ap.add_argument('--doc-fill', choices=['lognormal', 'full'], default='lognormal',
                    help='lognormal: documents zero-padded to T (SURVEY 8d, default); full: no padding')
    ap.add_argument('--token-dist', choices=['zipf', 'uniform'], default='zipf',
                    help='zipf (SURVEY 8d, default) or uniform word ids (most distinct tokens per batch)')
    ap.add_argument('--ramp', type=int, default=300,
                    help='untimed steps before the requested warm-up when --warmup is shorter than this: clocks '
                         'and launch queue reach steady state whatever --warmup says (reported as warmup_effective)')
    ap.add_argument('--opt-in-leg', action='store_true',
                    help='time the opt-in fp16-split-GEMM leg also where the configuration legs are off (it rides on the '
                         'default N = 1 line; never `value`)')
    ap.add_argument('--no-opt-in-leg', action='store_true', help='skip the opt-in arithmetic leg of the default N = 1 line')
    ap.add_argument('--gemm-math', choices=['f32', 'f16x2'], default='f32',
                    help='arithmetic of the projection GEMM: f32 (default, the headline: fp32 operands on the fp32 MFMA) or '
                         'the opt-in fp16-split form (every operand = hi + lo fp16, three f16-MFMA products per fp32 '
                         'product, fp32 accumulation; csrc/project_f16.hip) -- reported under its own dtype')
    ap.add_argument('--per-step', action='store_true',
                    help='native engine: one train_step call per step from Python (default: the steps are enqueued K per host '
                         'call through the span entry points, r4r_*_span, as main.train enqueues them)')
    ap.add_argument('--engine', choices=['native', 'module', 'graph'], default='native',
                    help="native: fused r4r_deepconn_step where the model has one (else module); module: op-by-op "
                         "autograd path; graph: the module path captured into one hipGraph per step")
    return ap.parse_args()


def blocked_sweep_leg(leg, engine, traffic, avg_s):
    """The sweep leg under temporal blocking: `achieved` stays the ALGORITHMIC rate (24 B per parameter
    and step), which exceeds the HBM peak once a visit applies several steps' updates per byte moved -- the same
    arithmetic, fewer bytes; `frac` is then what actually crossed HBM (PMC passes) against the peak."""
    if not (getattr(engine, 'TEMPORAL_SWEEP', False) and getattr(engine, '_tb_used', False)):
        return
    leg['algorithmic_frac'] = leg['frac']
    leg['hbm_side_GBs'] = None if traffic is None else round(traffic / avg_s / 1e9, 1)
    leg['frac'] = None if traffic is None else round(traffic / avg_s / 1e9 / PEAK_HBM_GBS, 4)
    leg['blocked_sweep_period'] = engine.sweep_period        # (frac = counter bytes per launch / launch time / HBM peak: DESIGN 5)


def tower_flops_per_doc(hp):
    """Algorithmic forward flops of ONE TextCNN tower on one document:
    P positions x F filters x 3E window x 2."""
    P = hp['input_length'] + 2
    return P * 100 * 3 * hp['word_embed_size'] * 2


def walked_positions(tokens):
    """(positions the gather-add-max launch walks, positions in all) for the documents `tokens` [..., T]: the launch
    cuts the T + 2 conv positions of a document into segments of 128 and every segment into four EQUAL slices (32
    positions; ceil(len / 4) in a shorter segment: NARRE's 102-position review, a 1000-word document's last 106); a slice
    whose tokens p_lo - 2 .. p_hi - 1 all name the same row (out-of-range tokens count as a row of their own) is decided
    by its first position (csrc/project.hip, proj_gather_max_kernel: `slen`, `same`, `npos_eff`)."""
    tok = np.asarray(tokens).reshape(-1, np.asarray(tokens).shape[-1])
    n, T = tok.shape
    P = T + 2
    ext = np.full((n, T + 4), -1, dtype=np.int64)           # ext[:, t + 2] = token t; t = -2, -1, T, T + 1 are out of range
    ext[:, 2:T + 2] = tok
    walked = 0
    for seg_lo in range(0, P, 128):
        seg_hi = min(P, seg_lo + 128)
        slen = 32 if seg_hi - seg_lo == 128 else -(-(seg_hi - seg_lo) // 4)
        for p_lo in range(seg_lo, seg_hi, slen):
            p_hi = min(seg_hi, p_lo + slen)
            sl = ext[:, p_lo:p_hi + 2]                       # tokens p_lo - 2 .. p_hi - 1
            same = (sl == sl[:, :1]).all(axis=1)
            walked += int(np.where(same, 1, p_hi - p_lo).sum())
    return walked, n * P


# The other BASELINE.json configurations and the data the projection conv does not like, as separately timed legs of
# the default N = 1 line (never `value`): (label, bench arguments that differ from the headline's)
CONFIG_LEGS = [
    ('cfg1_bias_only_musical', dict(workload='cfg1_bias_only_musical')),
    ('cfg2_mfdot_electronics', dict(workload='cfg2_mfdot_electronics')),
    ('cfg2_mfdot_electronics_b8192', dict(workload='cfg2_mfdot_electronics', batch_per_gpu=8192)),
    ('cfg4_narre_kindle', dict(workload='cfg4_narre_kindle')),
    ('cfg5_transnetpp_synthetic', dict(workload='cfg5_transnetpp_synthetic')),
    ('cfg3_full_uniform', dict(workload=WORKLOAD, doc_fill='full', token_dist='uniform')),
    # (the one true HBM gather: 1 M-word vocabulary, full-length documents of uniformly drawn words -- the projected rows
    # outgrow the Infinity Cache; the engines' measured rule would pick the direct conv here, so the leg pins the projection)
    ('cfg5_full_uniform_hbm_gather', dict(workload='cfg5_transnetpp_synthetic', doc_fill='full', token_dist='uniform',
                                          conv_algo='project')),
]


def config_legs_wanted(args, dp_job):
    """The legs ride on the default single-GPU line only: the headline workload at its default shape and engine."""
    return (not dp_job and not args.no_config_legs and (args.config_legs or not args.no_cpu_baseline)
            and args.workload == WORKLOAD and args.engine == 'native'
            and args.batch_per_gpu == 128 and args.scaling == 'weak' and not args.model_type and not args.embed
            and not args.latent and not args.from_host and args.doc_fill == 'lognormal' and args.token_dist == 'zipf'
            and args.gemm_math == 'f32' and args.conv_algo == 'auto')


def config_legs(args, env):
    """Each leg: its own model, engine and resident batch pool, --leg-steps timed steps between the same fences as the
    headline region, kernel timing sampled the same way.  A leg that fails reports its error and the line survives."""
    import copy
    import gc
    legs = []
    for label, over in CONFIG_LEGS:
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        a.steps, a.warmup = args.leg_steps, 20
        t0 = time.perf_counter()
        try:
            r = run(a, env, is_leg=True)
        except (Exception, SystemExit) as e:                 # noqa: BLE001  (reported, not raised: the headline line stands)
            legs.append({'leg': label, 'error': '%s: %s' % (type(e).__name__, str(e)[:300])})
            continue
        finally:
            gc.collect()
            torch.cuda.empty_cache()
        out = {'leg': label, 'workload': a.workload, 'batch': r['config']['batch_per_gpu'], 'doc_fill': a.doc_fill,
               'token_dist': a.token_dist, 'recommender': r['config']['shape']['recommender'],
               'engine': r['config']['engine'], 'ratings_per_s': r['value'], 'ms_per_step': r['ms_per_step'],
               'gpu_ms_per_step': r['gpu_ms_per_step'], 'steps': r['steps'], 'warmup': r['warmup'],
               'kernel_ms': r.get('kernel_ms'), 'leg_wall_s': round(time.perf_counter() - t0, 1)}
        if 'table_sweep' in r['config']:
            out['table_sweep'] = r['config']['table_sweep']
        for k in ('roofline', 'roofline_gather', 'roofline_gemm'):
            if k in r:
                out[k] = r[k]
        if 'roofline' in r:
            out['dominant_kernel'] = r['roofline']['kernel']
        legs.append(out)
    return legs


LEG_FIELDS = ['ratings_per_s', 'ms_per_step', 'dominant_kernel', 'kernel_ms', 'bound', 'frac', 'hbm_bytes_per_launch(pmc)']


def compact_leg(leg):
    """One `configs` leg as a short array in LEG_FIELDS order (+ the gather kernel's [ms, bound, frac, hbm_side_frac]
    where the leg has one)."""
    if 'error' in leg:
        return {'error': leg['error'][:120]}
    r = leg.get('roofline') or {}
    out = [leg['ratings_per_s'], leg['ms_per_step'], r.get('kernel'), r.get('avg_launch_ms'), r.get('bound'), r.get('frac'),
           r.get('traffic')]
    g = leg.get('roofline_gather')
    if g:
        out.append([g.get('avg_launch_ms'), g.get('bound'), g.get('frac'), g.get('hbm_side_frac')])
    return out


def emit(result):
    """The ONE JSON line, short enough to survive in the last 8 KB of stdout whole: the contract's fields, the roofline
    legs, the CPU baseline, and -- last -- every configuration leg as a short array (`legs`, LEG_FIELDS order), the
    opt-in arithmetic leg and the host-loop legs.  The full record (every leg's own roofline dictionaries) goes to
    stderr and to gpurun_out/bench_line_full.json."""
    full = json.dumps(result)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_line_full.json'), 'w') as f:
            f.write(full + '\n')
    except OSError:
        pass
    print('bench.py full record: ' + full, file=sys.stderr, flush=True)
    line = dict(result)
    legs = line.pop('configs', None)
    opt_in = line.pop('opt_in_f16_split', None)
    host = line.pop('host_loop', None)
    for key in ('roofline', 'roofline_gather', 'roofline_gemm'):
        if key in line:
            line[key] = {k: v for k, v in line[key].items() if k not in ('traffic_source', 'launches_in_timed_region',
                         'positions_per_launch', 'walked_positions_per_launch', 'projected_rows_bytes',
                         'distinct_token_rows_per_launch', 'algorithmic_GBs', 'steady_launches')}
    if 'steady' in line:
        line['steady'] = {k: line['steady'][k] for k in ('steps', 'ratings_per_s', 'ms_per_step') if k in line['steady']}
    line.pop('conv_equivalent', None)
    if legs is not None:
        line['leg_fields'] = LEG_FIELDS
        line['legs'] = {leg['leg']: compact_leg(leg) for leg in legs}
    if opt_in is not None:
        line['legs_opt_in_f16_split'] = [opt_in['ratings_per_s'], opt_in['ms_per_step']]
    if host is not None:
        line['host_loop_fields'] = ['main.train ratings_per_s', 'same engine, batches resident', 'train MSE']
        line['host_loop'] = host
    print(json.dumps(line), flush=True)


def host_loop_legs(env, ratings=20000):
    """The PRODUCT loop beside the bench's resident-pool loop: reviews4rec_amd.main.train over data.DataLoader (every
    batch built on the device from HBM-resident token pools, data.py:250-372's slices) driving the native engine through
    the span entry points, one epoch of `ratings` Amazon-shaped synthetic ratings in the reference's pickled schema
    (tools/synth_reviews.py), second and third epoch timed (the first warms workspaces); and the same engine on the
    loader's own batches held resident.  -> {family: [loop ratings/s, resident ratings/s, train MSE]}"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import reviews4rec_amd
    from reviews4rec_amd import main as M, synthetic
    from reviews4rec_amd.data import DataLoader
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.utils import xavier_init
    from synth_reviews import synthesize
    users, items, vocab, B = 8000, 3000, 50002, 128
    d = synthesize(ratings, users, items, vocab)
    out = {}
    for label, mt, E in (('deepconn_e300', 'deepconn', 300), ('narre_e64', 'NARRE', 64), ('mf_dot', 'MF_dot', 64)):
        hp = dict(model_type=mt, batch_size=B, input_length=1000, narre_num_reviews=10, narre_num_words=100,
                  total_users=users, total_items=items, latent_size=10 if mt != 'MF_dot' else 64, word_embed_size=E,
                  dropout=0.6, lr=0.002, weight_decay=1e-6, vocab=vocab, total_words=vocab, engine='native')
        hp['word_vectors'] = synthetic.word_table(vocab, E)
        train = DataLoader(hp, d['train'], d['user_reviews'], d['item_reviews'], None,
                           this_index_user_item=d['this_index_user_item'], device=env['dev'])
        torch.manual_seed(0)
        model = reviews4rec_amd.get_model_class(mt)(hp)
        xavier_init(model)
        model = model.to(env['dev'])
        engine = M.make_engine(hp, model)
        best, mse = None, None
        for epoch in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mse = M.train(model, MSELoss(hp), None, train, hp, engine=engine)['MSE']
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if epoch:
                best = dt if best is None else min(best, dt)
        keep = [b for _, b in zip(range(32), train.iter())]
        keep = [b for b in keep if b[1].shape[0] == B]
        kw = {'defer_sweep': True} if getattr(engine, 'TEMPORAL_SWEEP', False) else {}
        from reviews4rec_amd.data import SpanDescriptor
        desc = SpanDescriptor.resident(keep, review=mt != 'MF_dot')
        model.train()
        res = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            while n < 4 * len(keep):
                k = min(engine._span_limit(desc), engine.SPAN_STEPS)
                if k <= 0:
                    engine.train_step(*keep[n % len(keep)], n_global=B, **kw)
                    k = 1
                else:
                    engine._span(desc, n, k, True)
                n += k
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep:
                res = n * B / dt if res is None else max(res, n * B / dt)
        out[label] = [round(len(d['train']) / best), round(res), mse]
        del engine, model, train
        torch.cuda.empty_cache()
    return out


def cpu_baseline(hp, table, batches_np, budget_s):
    """The CPU oracle (stock ATen ops, same graph as the reference) on the host cores.
    ATen's intra-op pool is calibrated first: on a many-core host, os.cpu_count() threads
    on a batch-128 step is far slower than a moderate count, so one step is timed at a few
    thread counts and the fastest is used for the measured sample (`cores` = that count)."""
    import oracle
    ncpu = os.cpu_count() or 1
    P0 = oracle.init_params(hp, vocab_size=None if table is None else table.shape[0], seed=0)
    key = 'target.word2vec.weight' if hp['model_type'].startswith('transnet') else 'word2vec.weight'
    if key in P0:
        P0[key] = torch.from_numpy(table.copy())
    batches = [([torch.from_numpy(d) for d in data], torch.from_numpy(y)) for data, y in batches_np]

    tn = hp['model_type'].startswith('transnet')            # the three-optimiser step of main.py:35-53

    def one_step(P, state, i):
        data, y = batches[i % len(batches)]
        t = time.perf_counter()
        if tn:
            oracle.transnet_train_step(P, data, y, hp, state)
        else:
            oracle.train_step(P, data, y, hp, state)
        return time.perf_counter() - t

    def fresh_state():
        return dict(source=oracle.AdamState(), source_fm=oracle.AdamState(), target=oracle.AdamState()) if tn else oracle.AdamState()

    best, best_t = None, None
    t_cal = time.perf_counter()
    for threads in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(threads)
        P, state = {k: v.clone() for k, v in P0.items()}, fresh_state()
        one_step(P, state, 0)                               # warm-up at this thread count
        dt = one_step(P, state, 1)
        if best_t is None or dt < best_t:
            best, best_t = threads, dt
        if time.perf_counter() - t_cal > budget_s:          # bounded calibration
            break
    torch.set_num_threads(best)
    P, state = {k: v.clone() for k, v in P0.items()}, fresh_state()
    one_step(P, state, 0)
    n, t0 = 0, time.perf_counter()
    while True:
        one_step(P, state, n)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    B = batches[0][1].shape[0]
    return {'value': round(n * B / el, 2), 'unit': 'ratings/s', 'cores': best, 'kind': 'port',
            'sample': '%d full training steps of batch %d on the same synthetic %s batches (plain-PyTorch CPU '
                      'oracle, dropout %.1f; %d logical CPUs on the host, ATen threads calibrated to %d)'
                      % (n, B, hp['dataset'], hp['dropout'], ncpu, best)}


def make_engine(args, hp, model, dp, rank, world, B):
    """The fused native step of the workload's recommender (reviews4rec_amd.main.make_engine's
    selection, with the bench's conv-algo / seed knobs); None -> the op-by-op module path."""
    if args.engine != 'native':
        return None
    from reviews4rec_amd import engine as E
    from reviews4rec_amd.main import native_step_limits
    why_not = native_step_limits(dict(hp, batch_size=B), world)
    if why_not is not None:
        print('bench: no native step for this configuration (%s): module path' % why_not, file=sys.stderr)
        return None
    kw = dict(lr=hp['lr'], weight_decay=hp['weight_decay'], seed=4321, rank=rank, dp=dp)
    algo = {'auto': 0, 'direct': 1, 'project': 2}[args.conv_algo]
    mt = hp['model_type']
    if mt in ('MF_dot', 'bias_only'):
        return E.MFEngine(model, **kw)
    if mt in ('MF', 'NeuMF'):
        return E.IdNetEngine(model, **kw)
    if mt == 'NARRE':
        return E.NarreEngine(model, conv_algo=algo, **kw)
    if mt == 'deepconn++':
        return E.DeepCoNNPPEngine(model, conv_algo=algo, **kw)
    if mt in ('transnet', 'transnet++'):
        return E.TransNetEngine(model, conv_algo=algo, **kw)
    return E.DeepCoNNEngine(model, conv_algo=algo, **kw)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (no RANK / WORLD_SIZE in the environment): replace this
    process by the N-rank job the driver would have started -- torch.distributed.run, one rank per GPU, static
    rendezvous on 127.0.0.1 -- so that the bare command is a complete N-GPU run (rank 0 prints the one JSON line).
    The reference side of this is the single-process loop main.py:401-431; data parallelism is new work."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return
    check_device_count(args.gpus)
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what the host driver supports
    os.environ.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def check_device_count(n):
    """One rank per GPU over RCCL needs N visible devices; say so in one line BEFORE any process group exists (ranks that
    wrap around onto one device hang in ncclCommInitRank instead).  R4R_DIST_BACKEND=gloo is the test rig's several
    ranks on one GPU and is exempt."""
    if os.environ.get('R4R_DIST_BACKEND', 'nccl') != 'nccl':
        return
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n > have:
        raise SystemExit('bench.py: --gpus %d but %d GPU(s) visible: one rank per GPU over RCCL needs %d devices '
                         '(R4R_DIST_BACKEND=gloo lets test rigs share one)' % (n, have, n))


def main():
    args = parse()
    self_launch(args)
    from reviews4rec_amd import _lib, dist as r4dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if world_env != args.gpus:
        raise SystemExit('launched with WORLD_SIZE=%d but --gpus %d' % (world_env, args.gpus))
    if world_env > 1:
        check_device_count(world_env)
    rank, world, local = r4dist.init_from_env()
    # a data-parallel job: N > 1 -- or ONE rank with R4R_DP_SINGLE=1 (tests: the whole N > 1 path over RCCL on one GPU)
    dp_job = world > 1 or os.environ.get('R4R_DP_SINGLE') == '1'
    env = dict(rank=rank, world=world, local=local, dp_job=dp_job, dev=torch.device('cuda', local), lib=_lib.lib())
    result = run(args, env)
    if rank == 0:
        if config_legs_wanted(args, dp_job):
            result['configs'] = config_legs(args, env)
            try:
                result['host_loop'] = host_loop_legs(env)
            except (Exception, SystemExit) as e:             # noqa: BLE001  (reported, not raised: the headline line stands)
                result['host_loop'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        if not dp_job and not args.no_cpu_baseline:
            result['cpu_baseline'] = result.pop('_cpu_baseline_thunk')()
        result.pop('_cpu_baseline_thunk', None)
        emit(result)
    if dp_job:
        if env.get('dp') is not None:
            env['dp'].close()                                # the on-stream communicator, before the process group it was built over
        torch.distributed.destroy_process_group()


def run(args, env, is_leg=False):
    """One workload, timed: the headline line (is_leg=False) or one of the `configs` legs.  -> rank 0's result dict."""
    from reviews4rec_amd import synthetic
    import reviews4rec_amd
    from reviews4rec_amd.main import native_step_limits
    from reviews4rec_amd.loss import MSELoss
    from reviews4rec_amd.optim import Adam
    from reviews4rec_amd.ops import DropoutState
    from reviews4rec_amd.utils import xavier_init

    os.environ['R4R_GEMM_MATH'] = args.gemm_math             # read by the engines when they are built
    rank, world, dp_job, dev, lib = env['rank'], env['world'], env['dp_job'], env['dev'], env['lib']
    from reviews4rec_amd import dist as r4dist

    strong = args.scaling == 'strong'
    if strong:
        if args.global_batch % world:
            raise SystemExit('--global-batch %d is not divisible by %d ranks' % (args.global_batch, world))
        B = args.global_batch // world
    else:
        B = args.batch_per_gpu
    hp = synthetic.hyper_params_for(args.workload, batch_size=B, dropout=args.dropout)
    if args.model_type:
        hp['model_type'] = args.model_type
    if args.embed:
        hp['word_embed_size'] = args.embed
    if args.latent:
        hp['latent_size'] = args.latent
    if args.neumf_stage:
        hp['neumf_stage'] = args.neumf_stage
    table = synthetic.word_table(hp['vocab'], hp['word_embed_size']) if hp.get('vocab') else None
    if table is not None:
        hp['word_vectors'] = table

    torch.manual_seed(1234)                                  # same init on every rank
    model = reviews4rec_amd.get_model_class(hp['model_type'])(hp)
    xavier_init(model)                                       # main.py:377
    model = model.to(dev)
    model.train()
    dp = r4dist.DataParallel(model)
    env['dp'] = dp
    dp.broadcast_parameters()
    DropoutState.manual_seed(4321, rank)
    criterion = MSELoss(hp)
    is_tn = hp['model_type'] in ('transnet', 'transnet++')
    if is_tn:
        from reviews4rec_amd.utils import init_transnet_optim
        optimizer = init_transnet_optim(hp, model)           # utils.py:70-92: [source, source_fm, target, all]
    else:
        optimizer = Adam(model.parameters(), lr=hp['lr'], weight_decay=hp['weight_decay'])

    # each rank owns its shard of the stream (a contiguous shard of every global batch: the rank's
    # generator IS its shard -- ratings are i.i.d. draws)
    gen = synthetic.Generator(hp, seed=synthetic.SEED + rank, doc_fill=args.doc_fill, token_dist=args.token_dist)

    def make_pool(b, n=None):
        np_batches = [gen.batch(b) for _ in range(n or args.pool)]
        return np_batches, [([torch.from_numpy(d).to(dev) for d in data], torch.from_numpy(y).to(dev))
                            for data, y in np_batches]

    batches_np, pool = make_pool(B)
    metric_sum = torch.zeros((), device=dev)                 # sum of SE stays on the device (no per-step sync)
    B_global = B * world

    engine = make_engine(args, hp, model, dp, rank, world, B)
    if engine is None and is_tn and dp_job:
        raise SystemExit('TransNet under data parallelism needs the native step (see reviews4rec_amd.main.train)')
    # N > 1: measure the two gradient-exchange forms on this node's fabric once, before any timed
    # step, and keep the faster (R4R_DP_EXCHANGE=allreduce|gather pins one)
    exchange_ms = engine.autotune_exchange() if (engine is not None and dp_job
                                                 and hasattr(engine, 'autotune_exchange')) else {}

    graphed = None
    if args.engine == 'graph':
        if dp_job:
            raise SystemExit('--engine graph is single-GPU for now (the all-reduce is not captured)')
        from reviews4rec_amd.graph import GraphedStep
        graphed = GraphedStep(model, criterion, optimizer, *pool[0])   # TransNet: the optimiser list

    def make_step(pool, b, b_global):
        def step(i):
            data, y = pool[i % len(pool)]
            if graphed is not None:
                graphed(data, y)
                return
            if engine is not None:
                # forward + loss + backward + all-reduce + Adam; the next batch's token compaction overlaps it
                nxt = pool[(i + 1) % len(pool)][0]
                if getattr(engine, 'TEMPORAL_SWEEP', False) and args.token_prefetch == 'fused':
                    # (TransNet++: the ID-vector sweep temporally blocked, as main.train runs it; what it leaves
                    # pending is applied by engine.flush() INSIDE the timed region, before the closing fence)
                    engine.train_step(data, y, n_global=b_global, next_data=nxt, defer_sweep=True)
                else:
                    engine.train_step(data, y, n_global=b_global,
                                      next_data=nxt if args.token_prefetch == 'fused' else None)
                if args.token_prefetch == 'side-stream':
                    engine.prefetch_tokens(nxt)
                return
            model.zero_grad()
            if is_tn:                                        # the 3-optimiser step of main.py:35-53
                for o in optimizer:
                    o.zero_grad()
                out = model(data)
                criterion(out[1], y).backward(retain_graph=True)
                optimizer[2].step()
                out[2].backward(retain_graph=True)
                optimizer[0].step()
                se = criterion(out[0], y, return_mean=False)
                metric_sum.add_(se.detach().sum())
                torch.mean(se).backward()
                optimizer[1].step()
                return
            optimizer.zero_grad()
            out = model(data)
            se = criterion(out, y, return_mean=False)
            metric_sum.add_(se.detach().sum())
            (se.sum() * dp.loss_scale(b, b_global)).backward()
            dp.allreduce_grads()
            optimizer.step()
        return step

    step = make_step(pool, B, B_global)

    # The product loop (main.train over a device-side loader) enqueues the steps K per host call (engine._Spans ->
    # r4r_*_span: the same kernels and arguments as train_step, bit for bit); so does the bench, over its resident pool.
    span_desc = None
    # (data parallel: only where the engine's exchange can be issued from C too -- DeepCoNN over the on-stream RCCL
    # communicator, r4r_deepconn_span_dp; the other families' data-parallel steps stay one Python call per step)
    if (engine is not None and (not dp_job or (hasattr(engine, '_span_dp_ok') and engine._span_dp_ok()))
            and not args.from_host and graphed is None and not args.per_step
            and args.token_prefetch == 'fused' and hasattr(engine, '_span')):
        from reviews4rec_amd.data import SpanDescriptor
        d = SpanDescriptor.resident(pool, review=bool(hp.get('vocab')))
        if engine._span_ok(d):
            span_desc = d

    span_of = {}                                             # step function -> the resident descriptor of ITS pool

    def run_steps(step_fn, first, count, mask=0, sample=None):
        """Steps first .. first + count - 1; the sampled ones (kernel timing on) by themselves."""
        span_desc = span_of.get(step_fn)
        if span_desc is None:
            for i in range(first, first + count):
                lib.r4r_timing_enable(mask if (sample and sample(i - first)) else 0)
                step_fn(i)
            return
        i, end = first, first + count
        while i < end:
            if sample and sample(i - first):
                lib.r4r_timing_enable(mask)
                k = 1
            else:
                lib.r4r_timing_enable(0)
                k = 1
                while i + k < end and k < engine.SPAN_STEPS and not (sample and sample(i + k - first)):
                    k += 1
            k = min(k, engine._span_limit(span_desc))
            if k <= 0:                                       # (the conv rule's probe step reads a counter back: by itself)
                step_fn(i)
                i += 1
                continue
            engine._span(span_desc, i, k, True)
            i += k

    if span_desc is not None:
        span_of[step] = span_desc

    if args.from_host:
        from reviews4rec_amd import data_fast
        cat = [np.concatenate([b[0][k] for b in batches_np]) for k in range(7)]
        ycat = np.concatenate([b[1] for b in batches_np])
        loader = data_fast.DataLoader.from_arrays(dict(hp, batch_size=B), cat, ycat, device=dev)

        def host_batches():
            while True:
                for item in loader.iter():
                    yield item
        feed = host_batches()
        resident_step = step

        pool[:] = [next(feed), next(feed)]                   # [current, upcoming]: the host loop's lookahead

        def step(i):                                         # noqa: F811  (same step, batches arrive over PCIe)
            resident_step(0)                                 # trains pool[0], prepares the tokens of pool[1]
            pool[0], pool[1] = pool[1], next(feed)

    def fence(ev=None):
        # barrier + torch.cuda.synchronize() on both sides of the timed region, as the contract says.  The host
        # first SPINS on an event behind the queued work (`ev`: one already recorded there, else a new one):
        # synchronize() on an already idle device returns at once, while a blocked synchronize() is woken some tens of
        # microseconds after the GPU went idle -- 3-5 % of a 20-step (2 ms) region that are the host's wake-up
        # latency, not the steps'.
        if ev is None:
            ev = torch.cuda.Event()
            ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize()
        if dp_job:                                           # (one process: nobody to meet, and the device is already idle)
            torch.distributed.barrier()
            torch.cuda.synchronize()

    gpu_span_ms = [0.0]

    def timed_region(step_fn, steps, first, mask):
        """EXACTLY `steps` steps between two fences; -> max-over-ranks wall seconds."""
        # (everything the region needs is made BEFORE the fence: between the fence and the first launch the device is idle,
        # and that idle time is inside the region -- 35-75 us of a 2 ms one, tools: bench.py --step-events)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        ev0.record()                                         # (torch's current stream = the one every launch goes to; the device
        t0 = time.perf_counter()                             # is idle behind the fence: recorded before the clock starts, 4 us of host time)
        # sampled kernel timing: every 20th step (an instrumented step costs ~30 us more: its HIP events
        # carry release fences, and without them the spans stop agreeing with rocprofv3's kernel
        # durations -- measured: 56.0 vs 59.4 us for the GEMM), or -- short regions, where two
        # instrumented steps are already 1.5 % of the window -- one step in the middle
        sample = (lambda i: i % 20 == 10) if steps >= 50 else (lambda i: i == steps // 2)
        # (a short region's one instrumented step times the DOMINANT kernel only -- the `roofline` leg; every further
        # pair of events is another ~10 us of a 2 ms window, and the other legs have the steady leg's samples)
        m = mask if steps >= 50 else (mask & short_mask)
        marks = []
        if args.step_events:                                 # (diagnosis only: an event per step costs the region ~3 us each)
            for i in range(steps):
                lib.r4r_timing_enable(m if sample(i) else 0)
                step_fn(first + i)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
        else:
            run_steps(step_fn, first, steps, m, sample)
        if hasattr(engine, 'flush'):                         # every optimizer update of the K steps lands inside the region
            engine.flush(check=False)
        ev1.record()
        fence(ev1)
        elapsed = time.perf_counter() - t0
        # first launch's start -> last launch's end on the device: what the steps cost the GPU, without the two
        # host fences and the launch-queue fill the host clock also sees (value stays host-clock)
        gpu_span_ms[0] = ev0.elapsed_time(ev1)
        if marks:
            ts = [ev0.elapsed_time(e) for e in marks]
            print('step ends (us since the region\'s first event): ' + ' '.join('%.0f' % (1000 * t) for t in ts), file=sys.stderr)
            print('step spans (us): ' + ' '.join('%.0f' % (1000 * (b - a)) for a, b in zip([0.0] + ts[:-1], ts)), file=sys.stderr)
        lib.r4r_timing_enable(0)
        el = torch.tensor([elapsed], device=dev)
        if dp_job:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        return float(el.item())

    # The synthetic generators hold millions of Python objects; a generation-2 collection in the
    # middle of the timed loop is a 30-60 ms host pause, longer than the launch queue is deep
    # (seen as one idle GPU gap on the slow-step workloads).  Collect now and freeze the survivors.
    import gc
    gc.collect()
    gc.freeze()
    # Untimed ramp: a 20-step timed region is 2 ms of GPU time, and a GPU that was idle a moment ago
    # has neither its clocks nor its launch queue in steady state after a 5-step warm-up (that run read
    # 6 % low).  Whatever --warmup says, at least --ramp untimed steps precede the timed region.  How many it takes:
    # profiles/r03e_ramp_ab.txt -- the driver's 20-step command after 30 / 100 / 300 untimed steps reads 1.13 / 1.15 /
    # 1.18 M ratings/s (its sampled GEMM launch 59-61 / 57 / 55 us): the clocks settle over the first ~30 ms of work.
    ramp = max(0, args.ramp - args.warmup)
    run_steps(step, 0, ramp + args.warmup)
    # Kernel timing for the roofline legs happens INSIDE the timed region but is sampled: only every
    # 20th step is instrumented, and only the kernels the legs need (direct conv: slot 0; projection
    # GEMM + gather: slots 3, 4).  A HIP event record serialises the queue for ~3 us; instrumenting
    # every launch of every step cost 20 % of a 0.13 ms step, sampling costs ~1 %.
    mask = 0 if args.no_kernel_timing else (1 << 0) | (1 << 3) | (1 << 4) | \
        ((1 << 2) if hp['model_type'] in ('MF_dot', 'bias_only', 'transnet++') else 0)   # the Adam sweep is the leg
    # the dominant kernel's slot: the Adam sweep of the ID-table families, the projection GEMM (or the direct conv) otherwise
    short_mask = (1 << 2) if hp['model_type'] in ('MF_dot', 'bias_only', 'transnet++') else ((1 << 0) | (1 << 3))
    elapsed = timed_region(step, args.steps, ramp + args.warmup, mask)
    if hasattr(engine, 'check_announcements'):               # (outside the region: one int read back from the device)
        engine.check_announcements()
    gpu_ms_per_step = gpu_span_ms[0] / args.steps
    steps_run = ramp + args.warmup + args.steps
    slots = {'textcnn_fwd_kernel': 0, 'textcnn_wgrad_kernel': 1, 'adam_multi_kernel': 2,
             'proj_gemm_kernel': 3, 'proj_gather_max_kernel': 4}

    def read_slots():
        got = {}
        for name, slot in slots.items():
            tot, cnt = ctypes.c_double(), ctypes.c_int64()
            lib.r4r_timing_read(slot, ctypes.byref(tot), ctypes.byref(cnt), 0)
            if cnt.value:
                got[name] = (tot.value, cnt.value)           # (total ms, launches)
        lib.r4r_timing_read(0, ctypes.byref(ctypes.c_double()), ctypes.byref(ctypes.c_int64()), 1)   # reset ALL slots
        return got

    region_slots = read_slots()
    region_counts = {k: v[1] for k, v in region_slots.items()}          # sampled launches inside the requested region
    # A SHORT timed region (the driver's --steps 20: 2 ms of GPU time) carries ONE instrumented step -- more would
    # be a visible share of the window (an instrumented step costs ~30 us) -- so its roofline would rest on one
    # launch, and the region itself on a clock that has not settled.  Such a run appends a separately timed STEADY
    # leg: the same step function for 200 more steps between the same fences, every 20th step instrumented.  `value`
    # and the roofline legs stay the requested region's; the steady leg's launches are reported beside them.
    steady = None
    if (args.steps < 50 and mask and not args.from_host and graphed is None and engine is not None
            and not args.no_steady_leg and not is_leg):                     # (every rank takes the same branch: the legs' steps are collective)
        STEADY = 200
        el_s = timed_region(step, STEADY, steps_run, mask)
        if hasattr(engine, 'check_announcements'):
            engine.check_announcements()
        steady = {'steps': STEADY, 'ratings_per_s': round(STEADY * B_global / el_s, 1),
                  'ms_per_step': round(1000.0 * el_s / STEADY, 4), 'gpu_ms_per_step': round(gpu_span_ms[0] / STEADY, 4),
                  }   # (the same step function, 200 more steps between the same fences; never `value`: DESIGN 5)
        steps_run += STEADY
        leg_slots = read_slots()
        steady['kernel_ms'] = {k: round(v[0] / v[1], 4) for k, v in leg_slots.items()}
        steady['kernel_launches_sampled'] = {k: v[1] for k, v in leg_slots.items()}
        steady_slots = leg_slots
    else:
        steady_slots = {}
    # The roofline legs rest on the launches sampled INSIDE the requested region (the region `value` comes from); the
    # steady leg's launches are reported beside them (`steady_avg_launch_ms`, `steady_frac`), never pooled in.  A kernel
    # the region did not sample at all (it cannot happen with the sampling above) falls back to the steady leg's.
    timed = {k: (tot / cnt, cnt) for k, (tot, cnt) in {**steady_slots, **region_slots}.items()}   # (avg ms per launch, launches)

    def steady_of(kernel, leg_dict, scale_key='frac'):
        """Attach the steady leg's reading of the same kernel to a roofline leg."""
        if kernel in steady_slots and kernel in region_slots:
            tot, cnt = steady_slots[kernel]
            leg_dict['steady_avg_launch_ms'] = round(tot / cnt, 4)
            leg_dict['steady_launches'] = cnt
            if leg_dict.get(scale_key) is not None and leg_dict.get('avg_launch_ms'):
                leg_dict['steady_' + scale_key] = round(leg_dict[scale_key] * leg_dict['avg_launch_ms'] / (tot / cnt), 4)
    run_sse = float(engine.sse[0].item()) if engine is not None else (
        float(graphed.sse.item()) if graphed is not None else float(metric_sum.item()))

    # N > 1, weak: separately timed strong-scaling legs, each at a fixed global batch sharded over the ranks
    strong_legs = []
    if dp_job and not strong and not args.from_host and graphed is None and not is_leg:
        for G in [int(x) for x in str(args.strong_leg).split(',') if x.strip()]:
            if G % world:
                continue
            bs = G // world
            why = native_step_limits(dict(hp, batch_size=bs), world) if engine is not None else None
            if why:                                          # said, not skipped silently: the leg did not run on the native step
                strong_legs.append({'global_batch': G, 'batch_per_gpu': bs, 'skipped': 'native step limit: ' + why})
                continue
            _, pool_s = make_pool(bs, n=max(2, min(args.pool, 8192 // bs)))      # (large shards: two resident batches)
            step_s = make_step(pool_s, bs, G)
            if span_desc is not None:                        # (the leg's steps are enqueued like the weak region's)
                from reviews4rec_amd.data import SpanDescriptor as _SD
                d_s = _SD.resident(pool_s, review=bool(hp.get('vocab')))
                if engine._span_ok(d_s):
                    span_of[step_s] = d_s
            # (the ID-table engines pad every rank's shard to hyper_params['batch_size']: the leg's own)
            ehp = getattr(engine, 'hp', None)
            saved_bs = ehp.get('batch_size') if isinstance(ehp, dict) else None
            if isinstance(ehp, dict):
                ehp['batch_size'] = bs
            try:
                run_steps(step_s, 0, 10)
                el_s = timed_region(step_s, args.steps, 10, 0)
            finally:
                span_of.pop(step_s, None)
                if isinstance(ehp, dict):
                    ehp['batch_size'] = saved_bs
            strong_legs.append({'global_batch': G, 'batch_per_gpu': bs,
                                'ratings_per_s': round(args.steps * G / el_s, 1),
                                'ms_per_step': round(1000.0 * el_s / args.steps, 4), 'steps': args.steps})
            del pool_s

    # N = 1: a separately timed leg with the OPT-IN arithmetic of the projection GEMM (fp16-split operands,
    # fp32 accumulation: DESIGN.md 4.1d).  Reported beside the fp32 line, never as `value`.
    opt_in = None
    if (not dp_job and args.gemm_math == 'f32' and getattr(engine, 'gemm_math', None) == 'f32'
            and 'proj_gemm_kernel' in timed and not args.from_host and not args.no_opt_in_leg
            and (args.opt_in_leg or (not is_leg and config_legs_wanted(args, dp_job)))):
        from reviews4rec_amd import engine as E
        os.environ['R4R_GEMM_MATH'] = 'f16x2'
        engine.gemm_math = 'f16x2'
        E._MATH_OWNER[0] = None                              # the next step reads the scales and switches the GEMM
        try:
            for i in range(10):
                step(steps_run + i)
            el_o = timed_region(step, args.steps, steps_run + 10, 0)
            opt_in = {'gemm_math': 'f16x2', 'ratings_per_s': round(args.steps * B_global / el_o, 1),
                      'ms_per_step': round(1000.0 * el_o / args.steps, 4), 'steps': args.steps,
                      'note': 'opt-in (R4R_GEMM_MATH=f16x2): projection GEMM on fp16-split operands, 3 f16-MFMA '
                              'products per fp32 product, fp32 accumulation; error vs float64 at or below the '
                              'fp32 GEMM\'s (tests/test_gpu_kernels.py); NOT the headline'}
        finally:
            os.environ['R4R_GEMM_MATH'] = 'f32'
            engine.gemm_math = 'f32'
            E.apply_gemm_math(engine.table, engine._conv_weights())
            lib.r4r_timing_read(0, ctypes.byref(ctypes.c_double()), ctypes.byref(ctypes.c_int64()), 1)

    # replica check: after identical gradient sums and the identical dense Adam every rank must hold
    # the same bits; a broken exchange shows up here, not in a throughput number
    replicas = None
    if dp_job:
        flat = torch.cat([p.detach().reshape(-1).view(torch.int32).to(torch.int64) for p in model.parameters()
                          if p.dtype == torch.float32])
        digest = torch.stack([flat.sum(), (flat * (torch.arange(flat.numel(), device=dev) % 8191 + 1)).sum()])
        allsum = [torch.empty_like(digest) for _ in range(world)]
        torch.distributed.all_gather(allsum, digest)
        replicas = all(bool((d == allsum[0]).all()) for d in allsum)
        if not replicas:
            raise SystemExit('bench.py: replicas diverged (parameter digests differ across ranks) -- the '
                             'data-parallel exchange is broken; no number is reported')

    if rank == 0:
        value = args.steps * B_global / elapsed
        result = {
            'metric': 'train ratings/sec', 'value': round(value, 1), 'unit': 'ratings/s',
            # `warmup` is what was asked; `warmup_effective` what RAN untimed before the region (raised to --ramp steps: clock ramp)
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'warmup_effective': ramp + args.warmup,
            'warmup_requested': args.warmup,
            'ms_per_step': round(1000.0 * elapsed / args.steps, 4),
            'gpu_ms_per_step': round(gpu_ms_per_step, 4),     # HIP events around the same steps (rank 0's device)
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.gemm_math == 'f32' else 'f32 with an fp16-split projection GEMM (3 f16-MFMA products per '
                                                            'fp32 product, fp32 accumulate; opt-in, not the headline)',
            'data': 'synthetic' + (' (streamed from pinned host memory)' if args.from_host else
                                   ' (HBM-resident: a pool of %d batches on the device before the timed region)' % args.pool),
            'config': {'workload': args.workload, 'ratings_per_step': B_global, 'batch_per_gpu': B,
                       'parallelism': 'dp%d' % world,
                       'engine': 'native' if engine is not None else ('graph' if graphed is not None else 'module'),
                       'enqueue': 'spans (K steps per host call)' if span_desc is not None else 'per step',
                       'conv_algo': args.conv_algo, 'gemm_math': args.gemm_math, 'doc_fill': args.doc_fill,
                       'token_dist': args.token_dist,
                       **({'table_sweep': 'blocked, period %d, flushed inside the timed region' % engine.sweep_period}
                          if getattr(engine, 'TEMPORAL_SWEEP', False) and not dp_job
                          and (getattr(engine, 'plus', 0) or getattr(engine, 'has_tables', False))
                          and engine.sweep_period > 1 and args.token_prefetch == 'fused' else {}),
                       **({'rccl_ranks': world, 'dist_backend': torch.distributed.get_backend(),
                           # True: the collectives ran on the step's own stream through the package's communicator
                           'collectives_on_compute_stream': bool(dp is not None and getattr(dp, 'stream_rccl', None) is not None),
                           'replicas_identical': replicas} if dp_job else {}),
                       **({'dp_exchange': engine.exchange,
                           'dp_exchange_ms': {k: round(v, 4) for k, v in exchange_ms.items()}} if exchange_ms else {}),
                       'shape': {'recommender': hp['model_type'], 'word_embed_size': hp['word_embed_size'],
                                 'input_length': hp['input_length'], 'conv_filters': 100,
                                 'latent_size': hp['latent_size'], 'vocab': hp.get('vocab', 0),
                                 'users': hp['total_users'], 'items': hp['total_items'],
                                 'dropout': hp['dropout']}},
        }
        if opt_in:
            result['opt_in_f16_split'] = opt_in
        if steady:
            result['steady'] = steady
        if strong_legs:
            ran = [l for l in strong_legs if 'skipped' not in l]
            if ran:
                result['strong'] = ran[0]                     # G = 1024 (2.70 M ratings/s at N = 1)
            result['strong_legs'] = strong_legs
        result['kernel_ms'] = {k: round(v[0], 4) for k, v in timed.items()}
        towers = (3 if is_tn else 2) if engine is not None else 1    # the native step runs all towers per launch
        if 'proj_gather_max_kernel' in timed:
            # project-then-gather.  Dominant kernel: the projection GEMM over the
            # batch's DISTINCT tokens (fp32 MFMA).  Its useful flops are data dependent, so they are
            # counted on the host from the very batches that were run: rows x E x 300 x 2 per tower.
            per_side = [np.mean([len(np.unique(d[k])) for d, _ in batches_np]) for k in ((3, 4, 0) if is_tn else (3, 4))]
            rows = float(sum(per_side))
            g_s = timed['proj_gemm_kernel'][0] / 1000.0
            gflops = rows * hp['word_embed_size'] * 300 * 2
            ach = gflops / g_s / 1e12
            # (tables of E <= 64 run the weight-resident form of the GEMM, csrc/project.hip 2d, unless R4R_GEMM pins one)
            wres = -(-hp['word_embed_size'] // 16) <= 4 and os.environ.get('R4R_GEMM', 'r')[0] == 'r'
            gemm_name = 'proj_gemm_wres_kernel' if wres else 'proj_gemm_kernel'
            traffic, src = measured_traffic(gemm_name, args, g_s)
            f16 = args.gemm_math == 'f16x2'
            # fp16-split form: three f16 MFMA products per useful fp32 product -> a third of the dense f16 peak
            peak = PEAK_F16_MFMA_TFLOPS / 3.0 if f16 else PEAK_FP32_MFMA_TFLOPS
            result['roofline'] = {'kernel': 'proj_gemm_f16_kernel (+ its weight-pack launch)' if f16 else gemm_name,
                                  'bound': 'mfma', 'achieved': round(ach, 2),
                                  'peak': round(peak, 1), 'unit': 'TFLOP/s',
                                  'frac': round(ach / peak, 4),
                                  'traffic': traffic, 'traffic_source': src,
                                  'launches': timed['proj_gemm_kernel'][1],
                                  'launches_in_timed_region': region_counts.get('proj_gemm_kernel', 0),
                                  'avg_launch_ms': round(1000 * g_s, 4),
                                  'flops_per_launch': int(gflops), 'distinct_token_rows_per_launch': int(rows),
                                  # what the launch must move whatever its form: the distinct rows in, 1,200 B per row out
                                  # (at E = 64 the stores, not the MFMAs, bound it: DESIGN 4.1b round 3)
                                  'algorithmic_GBs': round(rows * (hp['word_embed_size'] * 4 + 1200) / g_s / 1e9, 1),
                                  'positions_per_launch': int(towers * B * hp['input_length'])}
            # second leg: the gather-add-max kernel streams every position's three 400-B tap rows + an 8-B
            # token id (1.2 KB + 8 B per position).  The projected rows it reads were written by
            # the GEMM a moment ago and are L2 / Infinity-Cache resident (cfg3: 35 MB), so its ceiling is
            # that cache's bandwidth, not HBM's: reported as achieved load bandwidth + the HBM-side bytes
            # the PMC passes counted, with NO fraction of an HBM peak.  Only a vocabulary whose projected
            # rows outgrow the 256 MB Infinity Cache (cfg5: 1 M words) makes this a true HBM gather.
            avg_s = timed['proj_gather_max_kernel'][0] / 1000.0
            # What the launch WALKS (csrc/project.hip, gather-add-max): a 32-position slice whose 34 tokens are all the same
            # row -- the zero-padded tail of a document, data.py:198-199 -- is decided by its first position alone, so only
            # the walked positions load their three 400-byte tap rows; every position's token id (8 B) and row slot (4 B)
            # are read regardless.  Counted on the host from the very batches that were run.
            walked, total = [], []
            for d, _ in batches_np:
                w_t = [walked_positions(d[k]) for k in ((3, 4, 0) if is_tn else (3, 4))]
                walked.append(sum(w[0] for w in w_t) * (towers / len(w_t)))
                total.append(sum(w[1] for w in w_t) * (towers / len(w_t)))
            walked, total = float(np.mean(walked)), float(np.mean(total))
            tokens = towers * B * (hp['narre_num_reviews'] * hp['narre_num_words'] if hp['model_type'] == 'NARRE'
                                   else hp['input_length'])
            nbytes = int(walked * 1200 + tokens * 12)
            ach_b = nbytes / avg_s / 1e9
            g_traffic, g_src = measured_traffic('proj_gather_max_kernel', args, avg_s)
            ptab_bytes = rows * 1216                          # (rows are stored at a 1,216-byte stride)
            leg = {'kernel': 'proj_gather_max_kernel', 'achieved': round(ach_b, 1), 'unit': 'GB/s',
                   'avg_launch_ms': round(1000 * avg_s, 4), 'launches': timed['proj_gather_max_kernel'][1],
                   'bytes_per_launch': nbytes, 'positions_per_launch': int(total),
                   'walked_positions_per_launch': int(walked),
                   'traffic': g_traffic, 'traffic_source': g_src, 'projected_rows_bytes': int(ptab_bytes)}
            if ptab_bytes > 256e6:
                # the projected rows outgrow the 256 MB Infinity Cache: a true HBM gather.  frac = walked bytes against
                # the HBM peak; hbm_side_frac = what the PMC passes counted crossing HBM against the same peak
                leg.update({'bound': 'hbm', 'peak': PEAK_HBM_GBS, 'frac': round(ach_b / PEAK_HBM_GBS, 4),
                            'hbm_side_GBs': None if g_traffic is None else round(g_traffic / avg_s / 1e9, 1),
                            'hbm_side_frac': None if g_traffic is None else round(g_traffic / avg_s / 1e9 / PEAK_HBM_GBS, 4)})
            else:
                leg.update({'bound': 'l2/mall', 'peak': PEAK_L2_GBS, 'frac': round(ach_b / PEAK_L2_GBS, 4),
                            'hbm_side_GBs': None if g_traffic is None else round(g_traffic / avg_s / 1e9, 1),
                            })
            steady_of('proj_gather_max_kernel', leg)
            result['roofline_gather'] = leg
            steady_of('proj_gemm_kernel', result['roofline'])
            result['conv_equivalent'] = {
                'TFLOPs': round(towers * B * tower_flops_per_doc(hp) / (avg_s + g_s) / 1e12, 1),
                'peak_fp32_mfma': PEAK_FP32_MFMA_TFLOPS}
            if hp['model_type'] == 'transnet++' and engine is not None and 'adam_multi_kernel' in timed:
                # TransNet++: the dominant kernel is the Adam sweep over the two ID-vector tables (24 B per
                # parameter), HBM-bound; the GEMM leg moves aside
                result['roofline_gemm'] = result['roofline']
                nparam = model.user_embedding.weight.numel() + model.item_embedding.weight.numel()
                avg_s = timed['adam_multi_kernel'][0] / 1000.0
                ach_b = nparam * 24 / avg_s / 1e9
                traffic, src = measured_traffic('mf_adam_kernel', args, avg_s)
                result['roofline'] = {'kernel': 'mf_adam_kernel', 'bound': 'hbm', 'achieved': round(ach_b, 1),
                                      'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(ach_b / PEAK_HBM_GBS, 4),
                                      'traffic': traffic, 'traffic_source': src,
                                      'launches': timed['adam_multi_kernel'][1],
                                      'avg_launch_ms': round(1000 * avg_s, 4), 'bytes_per_launch': int(nparam * 24),
                                      'parameters': int(nparam)}
                blocked_sweep_leg(result['roofline'], engine, traffic, avg_s)
                steady_of('adam_multi_kernel', result['roofline'], 'achieved')
        elif 'textcnn_fwd_kernel' in timed and hp.get('vocab'):
            flops = towers * B * tower_flops_per_doc(hp)
            avg_s = timed['textcnn_fwd_kernel'][0] / 1000.0
            ach = flops / avg_s / 1e12
            result['roofline'] = {'kernel': 'textcnn_fwd_kernel', 'bound': 'mfma', 'achieved': round(ach, 2),
                                  'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': None,
                                  'launches': timed['textcnn_fwd_kernel'][1], 'avg_launch_ms': round(1000 * avg_s, 4),
                                  'flops_per_launch': flops}
        elif hp['model_type'] in ('MF_dot', 'bias_only') and 'adam_multi_kernel' in timed:
            # ID-only models: the dense Adam sweep is the step (24 B per parameter when no
            # dense gradient is materialised -- the native step -- 28 B when it is -- the module path)
            nparam = sum(p.numel() for p in model.parameters())
            per = 24 if engine is not None else 28
            avg_s = timed['adam_multi_kernel'][0] / 1000.0
            ach_b = nparam * per / avg_s / 1e9
            traffic, src = measured_traffic('mf_adam_kernel', args, avg_s) if engine is not None else (None, None)
            result['roofline'] = {'kernel': 'mf_adam_kernel' if engine is not None else 'adam_multi_kernel',
                                  'bound': 'hbm', 'achieved': round(ach_b, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                  'frac': round(ach_b / PEAK_HBM_GBS, 4), 'traffic': traffic, 'traffic_source': src,
                                  'launches': timed['adam_multi_kernel'][1], 'avg_launch_ms': round(1000 * avg_s, 4),
                                  'bytes_per_launch': int(nparam * per), 'parameters': int(nparam)}
            blocked_sweep_leg(result['roofline'], engine, traffic, avg_s)
        if not is_leg:                                       # (runs after the config legs: main())
            cpu_hp = {k: v for k, v in hp.items() if k != 'word_vectors'}
            result['_cpu_baseline_thunk'] = lambda: cpu_baseline(cpu_hp, table, batches_np[:4], args.cpu_seconds)
        result['train_mse_running'] = round(run_sse / (steps_run * B), 4)
        gc.unfreeze()
        return result
    gc.unfreeze()
    return None


if __name__ == '__main__':
    main()
