"""Training loop (counterpart of the reference's main.py train / train_complete /
main_pytorch).  The per-batch sequence is the reference's (main.py:26-60):

    zero_grad -> forward -> per-example SE -> metrics += sum(SE) -> mean -> backward -> step

and ``metrics['MSE'] = round(sum SE / N, 4)`` (main.py:66).  Two deliberate host-side
differences, both numerically neutral: the running sum of SE stays on the device and
is read once per epoch (the reference syncs with ``float(torch.sum(..))`` every
batch, main.py:57), and ``optimizer`` is this package's fused Adam (same surface).
``hyper_params['engine']`` (default 'auto'): models with a fused native step (DeepCoNN 'deepconn' and
'deepconn++', NARRE, TransNet(++), MF_dot, bias_only) run the whole sequence as one native call per batch; the others run the
op-by-op step captured once into a hipGraph and replayed ('module' forces plain eager).
``hyper_params['spans']`` (default True): with a native engine and this package's device-side loader the epoch's full
batches are enqueued K = 64 steps per host call from C (engine._Spans, include/r4r.h "Spans": the batch construction and
the native step; the same kernels, arguments and bits as the per-batch loop below, which False restores).

TransNet's three-optimiser step (main.py:35-53) raises on torch >= 1.5 in the
reference; ``train`` restates its torch-0.4 behaviour: the three
backward passes run against one retained graph of pre-step activations while each
optimiser writes through ``.data`` (no autograd version bump).
"""
import datetime as dt
import os
import time

import torch

from .eval import evaluate, eval_ranking
from .loss import MSELoss
from .engine import pad_width as engine_pad_width
from .utils import file_write, init_transnet_optim, is_cuda_available, log_end_epoch, xavier_init

INF = 10000.0


def _is_transnet(hyper_params):
    return hyper_params['model_type'] in ['transnet', 'transnet++']


class _GraphHolder:
    """Lazily captured hipGraph of the module-engine step (hyper_params['engine'] == 'graph')."""

    def __init__(self):
        self.step = None


def _with_next(batches):
    """(batch, next batch or None) pairs: the native engine prepares the next batch's token state
    while it trains on the current one (DeepCoNNEngine.train_step, next_data=)."""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


def train(model, criterion, optimizer, reader, hyper_params, engine=None, dp=None, graph=None):
    model.train()
    tn = _is_transnet(hyper_params)
    metrics = {'MSE': 0.0}
    if tn:
        metrics['MSE_target'], metrics['MSE_transform'] = 0.0, 0.0
    total_x, total_batches = 0.0, 0.0
    device_sum = None
    if engine is not None:
        engine.sse.zero_()

    dp_on = dp is not None and dp.on
    if tn and dp_on and engine is None:
        # the op-by-op three-optimiser step has no gradient exchange between its three backward passes:
        # replicas would diverge silently (and the sparse-capture placeholders would reach Adam)
        raise RuntimeError("TransNet under data parallelism needs the native step (hyper_params['engine'] = "
                           "'auto' or 'native', batch_size * world <= 32768); the op-by-op three-optimiser "
                           "step is single-process only")
    # global batch sizes for the loss scale 1/B_global: ONE collective per epoch from the readers' batch
    # sizes (a per-step all-reduce + .item() would serialise the launch queue behind every batch)
    counts = dp.epoch_counts(reader) if dp_on else None
    # a device-side loader + a native engine: the epoch's full batches go to the device K steps per host call
    # (engine._Spans: the batch construction and the steps enqueued from C; the same kernels, arguments and bits)
    spanned = None
    if engine is not None and hyper_params.get('spans', True) and hasattr(engine, 'train_epoch') and (not dp_on or counts is not None):
        spanned = engine.train_epoch(reader, counts=counts)
    if spanned is not None:
        total_x, total_batches = spanned
        batches = ()
    else:
        batches = _with_next(reader.iter()) if engine is not None else ((b, None) for b in reader.iter())
    for step_no, ((data, y), upcoming) in enumerate(batches):
        n_local = int(y.shape[0])
        if not dp_on:
            n_global = n_local
        elif counts is not None:
            n_global = counts[step_no]
        else:                                                 # a reader that cannot say its batch sizes up front
            n_global = dp.global_count(n_local, y.device)
        if engine is not None:
            nxt = upcoming[0] if upcoming is not None else None
            if getattr(engine, 'TEMPORAL_SWEEP', False):       # TransNet++: untouched table chunks take their updates every
                engine.train_step(data, y, n_global=n_global, next_data=nxt, defer_sweep=True)   # few steps, together
            else:
                engine.train_step(data, y, n_global=n_global, next_data=nxt)
            total_x += float(n_local)
            total_batches += 1
            continue
        if graph is not None and not tn and not (dp is not None and dp.on):
            if graph.step is None:
                from .graph import GraphedStep
                graph.step = GraphedStep(model, criterion, optimizer, data, y)
                graph.step.sse.zero_()
            se = graph.step(data, y)
            total_x += float(n_local)
            total_batches += 1
            continue

        model.zero_grad()
        if tn:
            for o in optimizer:
                o.zero_grad()
        else:
            optimizer.zero_grad()
        all_output = model(data)

        if tn:
            optimizer_source, optimizer_source_fm, optimizer_target, optimizer_all = optimizer
            loss_target = criterion(all_output[1], y)
            loss_target.backward(retain_graph=True)
            optimizer_target.step()
            loss_transform = all_output[2]
            loss_transform.backward(retain_graph=True)
            optimizer_source.step()
            loss_source = criterion(all_output[0], y, return_mean=False)
            se_sum = torch.sum(loss_source.detach())
            torch.mean(loss_source).backward()
            optimizer_source_fm.step()
            metrics['MSE_target'] += float(loss_target.detach())
            metrics['MSE_transform'] += float(loss_transform.detach())
        else:
            loss = criterion(all_output, y, return_mean=False)
            se_sum = torch.sum(loss.detach())
            (torch.sum(loss) / float(n_global)).backward()       # == torch.mean(loss) on one rank
            if dp is not None:
                dp.allreduce_grads()
            optimizer.step()
        device_sum = se_sum if device_sum is None else device_sum + se_sum
        total_x += float(n_local)
        total_batches += 1

    if engine is not None:
        # once per epoch, one int each: the blocked sweeps' device-side flag (more updates pending than a visit can
        # apply: a broken schedule -- engine.check_announcements) and the peer exchange's timed_out word
        for probe in ('check_announcements', 'check_exchange'):
            if hasattr(engine, probe):
                getattr(engine, probe)()
    sse = float(engine.sse[0].item()) if engine is not None else (float(device_sum) if device_sum is not None else 0.0)
    if engine is not None and tn:                            # TransNetEngine: sums of the per-batch means
        aux = engine.sse[1:3].clone()
        if dp is not None and dp.on:                         # (each rank holds its shard's share of every batch mean)
            dp.sum_scalar(aux)
        metrics['MSE_target'], metrics['MSE_transform'] = float(aux[0].item()), float(aux[1].item())
    if graph is not None and graph.step is not None and engine is None:
        sse += float(graph.step.sse.item())
        graph.step.sse.zero_()
    if dp is not None and dp.on:
        t = torch.tensor([sse, total_x], dtype=torch.float64, device=next(model.parameters()).device)
        dp.sum_scalar(t)
        sse, total_x = float(t[0]), float(t[1])
    metrics['MSE'] = round(sse / float(total_x), 4)
    if tn:
        metrics['MSE_target'] = round(metrics['MSE_target'] / float(total_batches), 4)
        metrics['MSE_transform'] = round(metrics['MSE_transform'] / float(total_batches), 4)
    return metrics


def make_optimizer(hyper_params, model):
    from .optim import Adam
    if _is_transnet(hyper_params):
        return init_transnet_optim(hyper_params, model)
    return Adam(model.parameters(), lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'])


def native_step_limits(hyper_params, world=1):
    """None when the model's fused native step can run this configuration, else the reason it cannot
    (the limits are those the C entry points enforce: include/r4r.h, csrc/step_device.h)."""
    mt = hyper_params['model_type']
    B = int(hyper_params.get('batch_size', 128))
    L = int(hyper_params.get('latent_size', 10))
    E = int(hyper_params.get('word_embed_size', 64))
    R = int(hyper_params.get('narre_num_reviews', 10))
    if mt in ('MF_dot', 'bias_only'):
        if mt == 'MF_dot' and L > 256:
            return 'latent_size %d > 256' % L
        if B * world > (1 << 20):                            # (csrc/mf_engine.hip: MF_MAX_B_STEP)
            return 'global batch %d > %d' % (B * world, 1 << 20)
        return None
    if mt in ('MF', 'NeuMF'):
        if L > 64:                                          # (csrc/idnet_engine.hip: the head's LDS arrays, 85 KB at 64)
            return 'latent_size %d > 64' % L
        if B * world > 32768:
            return 'global batch %d > 32768' % (B * world)
        return None
    if mt not in ('deepconn', 'deepconn++', 'NARRE', 'transnet', 'transnet++'):
        return 'no fused native step for model_type %r' % (mt,)
    # csrc/engine.hip: DeepCoNN's head up to 128 (four FM inputs per lane, 115 KB of LDS); the others 64 (NARRE: the head's
    # 64 x 64 instantiation + the split step -- its scorer matrices no longer fit LDS beyond; DeepCoNN++, TransNet: 64
    # instantiations of their heads)
    maxL = {'deepconn': 128}.get(mt, 64)
    if L > maxL:
        return 'latent_size %d > %d' % (L, maxL)
    # (any word_embed_size: the engines zero-pad rows to whole float4 / whole K chunks -- engine.pad_width, exact -- and
    # the weight-gradient window takes another pass per 512 float4 columns beyond 680, csrc/wgrad_device.h)
    if mt == 'NARRE':
        if R > 64:
            return 'narre_num_reviews %d > 64' % R
        # (no cap on the ID entries per step any more: beyond the fused launch's 4,096 / the stand-alone launch's
        # 16,384 the rows are applied by the bucketed entry waves of csrc/rows_large.hip, engine.NarreEngine)
        return None
    if mt not in ('deepconn', 'NARRE') and B * world > 32768:       # (the LDS-staged row sweeps of csrc/mf_engine.hip: MF_MAX_B)
        return 'global batch %d > 32768' % (B * world)
    return None


def module_path_limits(hyper_params):
    """Same question for the op-by-op HIP path (csrc/smallops.hip: FM input width and rank <= 64, a
    dense layer's input width <= 255, TextCNN rows 16-byte aligned)."""
    mt = hyper_params['model_type']
    L = int(hyper_params.get('latent_size', 10))
    E = int(hyper_params.get('word_embed_size', 64))
    if mt in ('deepconn', 'deepconn++', 'NARRE', 'transnet', 'transnet++'):
        pass                        # (any word_embed_size: ops._padded_table zero-pads rows to whole float4, exact)
    fm_in = {'MF': 2 * L, 'deepconn': 2 * L, 'transnet': L, 'transnet++': L + 10}.get(mt)
    if fm_in is not None and fm_in > 512:
        return 'factorization machine over %d inputs > 512 (latent_size %d)' % (fm_in, L)
    if mt in ('MF', 'NeuMF', 'NARRE', 'deepconn++') and 2 * L > 255:
        return 'dense layer over %d inputs > 255' % (2 * L)
    if mt == 'NARRE' and (int(hyper_params.get('narre_num_reviews', 10)) > 32 or L > 32):
        return 'NARRE attention over more than 32 reviews / 32 latent dimensions'
    return None


def make_engine(hyper_params, model, dp=None, rank=0):
    """The fused native step, where the model has one and the configuration fits it.  'auto' falls
    back to the op-by-op HIP path (with the reason logged) when it does not; 'native' raises; a
    configuration NEITHER path can run raises one clear error here, before any training."""
    want = hyper_params.get('engine', 'auto')
    world = dp.world if (dp is not None and dp.on) else 1
    why_not = native_step_limits(hyper_params, world)
    if want not in ('auto', 'native') or why_not is not None:
        if want == 'native':
            raise RuntimeError("hyper_params['engine'] = 'native': " + why_not)
        blocked = module_path_limits(hyper_params)
        if blocked is not None:
            raise RuntimeError('reviews4rec_amd cannot run this configuration on the HIP path: ' + blocked +
                               ((' (and the fused native step: ' + why_not + ')') if why_not else ''))
        if want == 'auto' and why_not is not None:          # loud: this path is several times slower
            import warnings
            msg = "engine 'auto': the fused native step cannot run this configuration (" + why_not + \
                "); using the op-by-op HIP path"
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
            if hyper_params.get('log_file'):
                file_write(hyper_params['log_file'], msg, dont_print=True)
        return None
    from . import engine as E
    kw = dict(lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'],
              seed=hyper_params.get('seed', 0x5EED5EED), rank=rank, dp=dp)
    mt = hyper_params['model_type']
    if mt in ('MF_dot', 'bias_only'):
        return E.MFEngine(model, **kw)
    if mt in ('MF', 'NeuMF'):
        return E.IdNetEngine(model, **kw)
    if mt == 'NARRE':
        return E.NarreEngine(model, **kw)
    if mt == 'deepconn++':
        return E.DeepCoNNPPEngine(model, **kw)
    if _is_transnet(hyper_params):
        return E.TransNetEngine(model, **kw)
    return E.DeepCoNNEngine(model, **kw)


def train_complete(hyper_params, Model, train_reader, val_reader, user_count, item_count, model, review=True,
                   dp=None):
    file_write(hyper_params['log_file'], '\n\nSimulation run on: ' + str(dt.datetime.now()) + '\n\n')
    file_write(hyper_params['log_file'], 'Data reading complete!')
    file_write(hyper_params['log_file'], 'Number of train batches: {:4d}'.format(len(train_reader)))
    file_write(hyper_params['log_file'], 'Number of validation batches: {:4d}'.format(len(val_reader)))

    criterion = MSELoss(hyper_params)
    rank = dp.rank if dp is not None else 0
    if dp is not None:
        dp.rebind(model)        # THIS model's parameters are what gets exchanged; replicas start from rank 0's
    engine = make_engine(hyper_params, model, dp=dp, rank=rank)
    # the readers hold millions of Python objects: without this a generation-2 collection stops the
    # host for tens of milliseconds in the middle of an epoch, longer than the launch queue is deep
    import gc
    gc.collect()
    gc.freeze()
    optimizer = None if engine is not None else make_optimizer(hyper_params, model)
    # no fused step for this model: capture the op-by-op step in a hipGraph (TransNet's three-optimiser
    # step and data-parallel runs stay eager inside train())
    graph = _GraphHolder() if (engine is None and hyper_params.get('engine', 'auto') in ('auto', 'graph')) else None

    file_write(hyper_params['log_file'], str(model))
    file_write(hyper_params['log_file'], '\nModel Built!\nStarting Training...\n')
    # Epoch-level resume (absent upstream): hyper_params['checkpoint_path'] names a file
    # rewritten after every epoch with the weights, the optimiser state (Adam moments, step counts,
    # dropout stream position) and the loop state; if it exists when training starts, training
    # continues after the epoch it records and lands on the same weights as an uninterrupted run.
    ckpt_path = hyper_params.get('checkpoint_path')
    first_epoch, best_MSE = 1, float(INF)
    if ckpt_path and os.path.exists(ckpt_path):
        ck = torch.load(ckpt_path, map_location='cpu')
        if ck.get('model_class', type(model).__name__) != type(model).__name__:
            raise RuntimeError('checkpoint {} holds a {} run, this run trains a {}'.format(
                ckpt_path, ck['model_class'], type(model).__name__))
        model.load_state_dict(ck['model'])
        if engine is not None:
            engine.load_state_dict(ck['optimizer'])
        elif isinstance(optimizer, (list, tuple)):
            for o, sd in zip(optimizer, ck['optimizer']):
                o.load_state_dict(sd)
        else:
            optimizer.load_state_dict(ck['optimizer'])
        from . import ops
        ops.DropoutState.offset = int(ck.get('dropout_offset', 0))    # module path's Philox stream position
        per_rank = ck.get('rank_offsets')                             # ragged shards: every rank has its own
        if per_rank is not None and len(per_rank) > rank:
            ops.DropoutState.offset = int(per_rank[rank][1])
            if engine is not None:
                engine.offset = int(per_rank[rank][0])
        if ops.DropoutState.device_counter is not None:               # ... which a captured step keeps on the device
            ops.DropoutState.device_counter.fill_(ops.DropoutState.offset)
        first_epoch, best_MSE = int(ck['epoch']) + 1, float(ck['best_MSE'])
        file_write(hyper_params['log_file'], 'Resuming after epoch {:d} from {}'.format(int(ck['epoch']), ckpt_path))
    try:
        for epoch in range(first_epoch, hyper_params['epochs'] + 1):
            epoch_start_time = time.time()
            metrics = train(model, criterion, optimizer, train_reader, hyper_params, engine=engine, dp=dp,
                            graph=graph)
            metrics['dataset'] = hyper_params['dataset']
            metrics, _, _ = evaluate(model, criterion, val_reader, hyper_params, user_count, item_count,
                                     review=review, engine=engine)
            metrics['dataset'] = hyper_params['dataset']
            log_end_epoch(hyper_params, metrics, epoch, time.time() - epoch_start_time, metrics_on='(VAL)')
            if metrics['MSE'] < best_MSE:
                if rank == 0:
                    print('Saving model...')
                    tmp = hyper_params['model_path'] + '.tmp'
                    torch.save(model.state_dict(), tmp)
                    os.replace(tmp, hyper_params['model_path'])     # never a partial file under the final name
                best_MSE = metrics['MSE']
            if ckpt_path:
                from . import ops
                drop_at = ops.DropoutState.offset if ops.DropoutState.device_counter is None else \
                    int(ops.DropoutState.device_counter.item())
                mine = [int(getattr(engine, 'offset', 0)) if engine is not None else 0, int(drop_at)]
                rank_offsets = dp.gather_ints(mine) if (dp is not None and dp.on) else [mine]
            if ckpt_path and rank == 0:
                if engine is not None:
                    opt_sd = engine.state_dict()
                elif isinstance(optimizer, (list, tuple)):
                    opt_sd = [o.state_dict() for o in optimizer]
                else:
                    opt_sd = optimizer.state_dict()
                tmp = ckpt_path + '.tmp'
                torch.save({'epoch': epoch, 'best_MSE': best_MSE, 'model': model.state_dict(),
                            'model_class': type(model).__name__,
                            'optimizer': opt_sd, 'dropout_offset': drop_at, 'rank_offsets': rank_offsets}, tmp)
                os.replace(tmp, ckpt_path)                  # a crash mid-write leaves the previous one intact
    except KeyboardInterrupt:
        print('Exiting from training early')

    # reload the best-on-validation checkpoint into a fresh model (main.py:131-134); under data
    # parallelism only rank 0 wrote it: nobody reads before the writer is done
    if dp is not None and dp.on:
        dp.barrier()
    model = Model(hyper_params)
    if is_cuda_available:
        model = model.cuda()
    model.load_state_dict(torch.load(hyper_params['model_path']))
    model.eval()
    return model


def _load_readers(hyper_params, review_based_model):
    """The readers main.py:357-372 builds.  The reference prefers its preprocessed-epoch loader
    (data_fast.py) for review models because its pickle loader rebuilds batches in Python; here the
    pickle loader builds batches on the device from HBM-resident token pools (data.py), so it is the
    default.  hyper_params['loader'] = 'fast' asks for the preprocessed files (quick_data_*)."""
    from .data import load_data
    if review_based_model and hyper_params.get('loader', 'pools') == 'fast':
        from .data_fast import load_data_fast
        train_reader, test_reader, val_reader, hyper_params = load_data_fast(hyper_params)
        _, ranking_reader, _, _ = load_data(hyper_params)               # main.py:394: needs the pickle reader
        return train_reader, test_reader, val_reader, ranking_reader, hyper_params
    train_reader, test_reader, val_reader, hyper_params = load_data(hyper_params)
    return train_reader, test_reader, val_reader, test_reader, hyper_params


def _final_engine(hyper_params, model):
    """A native engine around the reloaded best model, for the test-set passes (eval only)."""
    if not is_cuda_available:
        return None
    return make_engine(dict(hyper_params, log_file=None), model)


def main_NeuMF(hyper_params, readers=None, user_count=None, item_count=None, dp=None, ranking_reader=None,
               gpu_id=None):
    """Counterpart of main.main_NeuMF (main.py:289-340): pre-train GMF, pre-train MLP (each with its own
    checkpoint path), initialise NeuMF from both, train it, evaluate MSE + HR@1.  ``readers`` =
    (train, test, val) skips the loading (synthetic data); without it the dataset directory is read
    like the reference does."""
    from .pytorch_models.NeuMF import GMF, MLP, NeuMF
    from .utils import load_user_item_counts
    if readers is None:
        user_count, item_count = load_user_item_counts(hyper_params)
        train_reader, test_reader, val_reader, ranking_reader, hyper_params = _load_readers(hyper_params, False)
    else:
        train_reader, test_reader, val_reader = readers
    user_count = {} if user_count is None else user_count
    item_count = {} if item_count is None else item_count
    start_time = time.time()
    initial_path, initial_ckpt = hyper_params['model_path'], hyper_params.get('checkpoint_path')
    stage_models = {}
    for tag, cls in (('_gmf', GMF), ('_mlp', MLP)):
        hyper_params['model_path'] = initial_path + tag
        if initial_ckpt:                                  # every stage resumes from its OWN epoch checkpoint
            hyper_params['checkpoint_path'] = initial_ckpt + tag
        model = cls(hyper_params)
        if is_cuda_available:
            model = model.cuda()
        xavier_init(model)                                # main.py:300,309
        stage_models[tag] = train_complete(hyper_params, cls, train_reader, val_reader, user_count, item_count,
                                           model, review=False, dp=dp)
    hyper_params['model_path'] = initial_path
    if initial_ckpt:
        hyper_params['checkpoint_path'] = initial_ckpt
    model = NeuMF(hyper_params)
    if is_cuda_available:
        model = model.cuda()
    model.init(stage_models['_gmf'], stage_models['_mlp'])    # main.py:326 (no xavier_init here)
    model = train_complete(hyper_params, NeuMF, train_reader, val_reader, user_count, item_count, model,
                           review=False, dp=dp)
    criterion = MSELoss(hyper_params)
    metrics, user_count_mse_map, item_count_mse_map = evaluate(
        model, criterion, test_reader, hyper_params, user_count, item_count, review=False)
    if ranking_reader is not None:
        metrics.update(eval_ranking(model, ranking_reader, hyper_params, review=False))
    log_end_epoch(hyper_params, metrics, 'final', time.time() - start_time, metrics_on='(TEST)')
    return metrics, user_count_mse_map, item_count_mse_map


def main_pytorch(hyper_params, readers=None, user_count=None, item_count=None, review_based_model=None, dp=None,
                 ranking_reader=None, gpu_id=None):
    """Counterpart of main.main_pytorch (main.py:342-399).  ``readers`` = (train, test, val) skips the
    loading (synthetic data); without it the dataset directory is read like the reference does:
    train-set counts, the three loaders, and the negatives reader for HR@1."""
    import reviews4rec_amd
    from .utils import load_user_item_counts
    Model = reviews4rec_amd.get_model_class(hyper_params['model_type'])
    if review_based_model is None:
        review_based_model = hyper_params['model_type'] not in ['bias_only', 'MF', 'MF_dot', 'NeuMF']
    if readers is None:
        user_count, item_count = load_user_item_counts(hyper_params)
        train_reader, test_reader, val_reader, ranking_reader, hyper_params = \
            _load_readers(hyper_params, review_based_model)
    else:
        train_reader, test_reader, val_reader = readers
    user_count = {} if user_count is None else user_count
    item_count = {} if item_count is None else item_count
    model = Model(hyper_params)
    if is_cuda_available:
        model = model.cuda()
    xavier_init(model)                                    # main.py:377
    start_time = time.time()
    model = train_complete(hyper_params, Model, train_reader, val_reader, user_count, item_count, model,
                           review=review_based_model, dp=dp)
    criterion = MSELoss(hyper_params)
    engine = _final_engine(hyper_params, model)           # the fused eval forward scores the test passes
    metrics, user_count_mse_map, item_count_mse_map = evaluate(
        model, criterion, test_reader, hyper_params, user_count, item_count, review=review_based_model,
        engine=engine)
    if ranking_reader is not None:
        metrics.update(eval_ranking(model, ranking_reader, hyper_params, review=review_based_model, engine=engine))
    log_end_epoch(hyper_params, metrics, 'final', time.time() - start_time, metrics_on='(TEST)')
    return metrics, user_count_mse_map, item_count_mse_map


def main(hyper_params, gpu_id=None):
    """main.main (main.py:400-430) for the model families on the accelerated path; returns the metrics."""
    if gpu_id is not None:
        torch.cuda.set_device(int(gpu_id))
    mt = hyper_params['model_type']
    if mt in ['SVD', 'kNN', 'NMF', 'SVD++', 'baseline', 'HFT', 'MPCN']:
        raise ValueError('model_type %r (Surprise / HFT / MPCN) is outside the accelerated path; run it with the '
                         'reference' % (mt,))
    method = main_NeuMF if mt == 'NeuMF' else main_pytorch
    metrics, user_count_mse_map, item_count_mse_map = method(hyper_params, gpu_id=gpu_id)
    return metrics


if __name__ == '__main__':                                # main.py:432-434: python -m reviews4rec_amd.main
    from .hyper_params import hyper_params
    main(hyper_params)
