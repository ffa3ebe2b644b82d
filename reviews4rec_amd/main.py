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

    batches = _with_next(reader.iter()) if engine is not None else ((b, None) for b in reader.iter())
    for (data, y), upcoming in batches:
        n_local = int(y.shape[0])
        n_global = dp.global_count(n_local, y.device) if (dp is not None and dp.on) else n_local
        if engine is not None:
            engine.train_step(data, y, n_global=n_global, next_data=upcoming[0] if upcoming is not None else None)
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


def make_engine(hyper_params, model, dp=None, rank=0):
    """The fused native step, where the model has one and the config asks for it."""
    if hyper_params.get('engine', 'auto') not in ('auto', 'native'):
        return None
    if hyper_params['model_type'] in ('MF_dot', 'bias_only'):
        world = dp.world if (dp is not None and dp.on) else 1
        if int(hyper_params.get('batch_size', 128)) * world > 16384:
            return None                                   # very large (global) batches: module path (dist.py C2)
        from .engine import MFEngine
        return MFEngine(model, lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'],
                        seed=hyper_params.get('seed', 0x5EED5EED), rank=rank, dp=dp)
    if hyper_params['model_type'] == 'NARRE':
        world = dp.world if (dp is not None and dp.on) else 1
        entries = int(hyper_params.get('batch_size', 128)) * (1 + int(hyper_params.get('narre_num_reviews', 10)))
        if entries > 4096 or (world > 1 and entries * world > 16384):
            return None                                   # more ID entries than the rows role holds: module path
        from .engine import NarreEngine
        return NarreEngine(model, lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'],
                           seed=hyper_params.get('seed', 0x5EED5EED), rank=rank, dp=dp)
    if hyper_params['model_type'] == 'deepconn++':
        world = dp.world if (dp is not None and dp.on) else 1
        if int(hyper_params.get('batch_size', 128)) * world > 16384:
            return None
        from .engine import DeepCoNNPPEngine
        return DeepCoNNPPEngine(model, lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'],
                                seed=hyper_params.get('seed', 0x5EED5EED), rank=rank, dp=dp)
    if _is_transnet(hyper_params):
        world = dp.world if (dp is not None and dp.on) else 1
        if int(hyper_params.get('batch_size', 128)) * world > 16384:
            return None
        from .engine import TransNetEngine
        return TransNetEngine(model, lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'],
                              seed=hyper_params.get('seed', 0x5EED5EED), rank=rank, dp=dp)
    if hyper_params['model_type'] != 'deepconn':
        return None
    from .engine import DeepCoNNEngine
    return DeepCoNNEngine(model, lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'], dp=dp,
                          seed=hyper_params.get('seed', 0x5EED5EED), rank=rank)


def train_complete(hyper_params, Model, train_reader, val_reader, user_count, item_count, model, review=True,
                   dp=None):
    file_write(hyper_params['log_file'], '\n\nSimulation run on: ' + str(dt.datetime.now()) + '\n\n')
    file_write(hyper_params['log_file'], 'Data reading complete!')
    file_write(hyper_params['log_file'], 'Number of train batches: {:4d}'.format(len(train_reader)))
    file_write(hyper_params['log_file'], 'Number of validation batches: {:4d}'.format(len(val_reader)))

    criterion = MSELoss(hyper_params)
    rank = dp.rank if dp is not None else 0
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
                    torch.save(model.state_dict(), hyper_params['model_path'])
                best_MSE = metrics['MSE']
            if ckpt_path and rank == 0:
                if engine is not None:
                    opt_sd = engine.state_dict()
                elif isinstance(optimizer, (list, tuple)):
                    opt_sd = [o.state_dict() for o in optimizer]
                else:
                    opt_sd = optimizer.state_dict()
                tmp = ckpt_path + '.tmp'
                from . import ops
                drop_at = ops.DropoutState.offset if ops.DropoutState.device_counter is None else \
                    int(ops.DropoutState.device_counter.item())
                torch.save({'epoch': epoch, 'best_MSE': best_MSE, 'model': model.state_dict(),
                            'optimizer': opt_sd, 'dropout_offset': drop_at}, tmp)
                os.replace(tmp, ckpt_path)                  # a crash mid-write leaves the previous one intact
    except KeyboardInterrupt:
        print('Exiting from training early')

    # reload the best-on-validation checkpoint into a fresh model (main.py:131-134)
    model = Model(hyper_params)
    if is_cuda_available:
        model = model.cuda()
    model.load_state_dict(torch.load(hyper_params['model_path']))
    model.eval()
    return model


def main_NeuMF(hyper_params, readers, user_count=None, item_count=None, dp=None, ranking_reader=None):
    """Counterpart of main.main_NeuMF (main.py:289-340) over already-built readers: pre-train GMF,
    pre-train MLP (each with its own checkpoint path), initialise NeuMF from both, train it,
    evaluate MSE (+ HR@1 when a negatives reader is given)."""
    from .pytorch_models.NeuMF import GMF, MLP, NeuMF
    train_reader, test_reader, val_reader = readers
    user_count = {} if user_count is None else user_count
    item_count = {} if item_count is None else item_count
    start_time = time.time()
    initial_path = hyper_params['model_path']
    stage_models = {}
    for tag, cls in (('_gmf', GMF), ('_mlp', MLP)):
        hyper_params['model_path'] = initial_path + tag
        model = cls(hyper_params)
        if is_cuda_available:
            model = model.cuda()
        xavier_init(model)                                # main.py:300,309
        stage_models[tag] = train_complete(hyper_params, cls, train_reader, val_reader, user_count, item_count,
                                           model, review=False, dp=dp)
    hyper_params['model_path'] = initial_path
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


def main_pytorch(hyper_params, readers, user_count=None, item_count=None, review_based_model=True, dp=None,
                 ranking_reader=None):
    """Counterpart of main.main_pytorch (main.py:342-399) over already-built readers
    ``(train, test, val)`` -- the pickle-based slow loader of the reference is out of scope."""
    import reviews4rec_amd
    Model = reviews4rec_amd.get_model_class(hyper_params['model_type'])
    train_reader, test_reader, val_reader = readers
    user_count = {} if user_count is None else user_count
    item_count = {} if item_count is None else item_count
    model = Model(hyper_params)
    if is_cuda_available:
        model = model.cuda()
    xavier_init(model)                                    # main.py:377
    if dp is not None:
        dp.model, dp.params = model, [p for p in model.parameters() if p.requires_grad]
        dp.broadcast_parameters()
    start_time = time.time()
    model = train_complete(hyper_params, Model, train_reader, val_reader, user_count, item_count, model,
                           review=review_based_model, dp=dp)
    criterion = MSELoss(hyper_params)
    metrics, user_count_mse_map, item_count_mse_map = evaluate(
        model, criterion, test_reader, hyper_params, user_count, item_count, review=review_based_model)
    if ranking_reader is not None:
        metrics.update(eval_ranking(model, ranking_reader, hyper_params, review=review_based_model))
    log_end_epoch(hyper_params, metrics, 'final', time.time() - start_time, metrics_on='(TEST)')
    return metrics, user_count_mse_map, item_count_mse_map
