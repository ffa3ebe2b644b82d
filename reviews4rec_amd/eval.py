"""Evaluation (counterpart of the reference's eval.py).

``evaluate`` returns exactly what eval.py:11-62 returns -- ``{'MSE': round(sum SE / N,
4)}`` (+ MSE_right / MSE_transform for TransNet) and the two train-frequency -> [SE]
maps -- but keeps the per-example SE on the device during the pass, copies once per split and
groups on the host with numpy instead of the reference's two scalar reads per example
(eval.py:42-53).  ``eval_ranking`` is HR@1
over negatives-shaped batches (eval.py:64-92).
"""
import torch


def _count_maps(ids, ses, counts):
    """{train frequency of the id: [SE, ...] in encounter order} (eval.py:45-53), vectorised:
    one dict lookup per DISTINCT id instead of per example; ids the map has not seen are
    entered with 0, as the reference does."""
    import numpy as np
    uniq, inv = np.unique(ids, return_inverse=True)
    freq = np.empty(len(uniq), dtype=np.int64)
    for k, u in enumerate(uniq.tolist()):
        freq[k] = counts.setdefault(u, 0)
    f = freq[inv]
    order = np.argsort(f, kind='stable')                     # stable: encounter order inside a group
    fs, ss = f[order], ses[order]
    cuts = np.flatnonzero(np.diff(fs)) + 1
    keys = fs[np.concatenate(([0], cuts))] if len(fs) else []
    return {int(k): grp.tolist() for k, grp in zip(keys, np.split(ss, cuts))}


EVAL_LAUNCH = 8          # x batch_size ratings per launch of a validation pass scored by a native engine
EVAL_LAUNCH_CAP = 4096   # ... but never more than max(batch_size, this) ratings
RANK_LAUNCH = 4          # x batch_size ranking rows (of 6 candidates each)


def _launch_size(reader, hyper_params, engine, factor):
    """Slice length of a validation pass.  The reference scores slices of ``batch_size`` (eval.py:24,
    data.py:379); a native engine's eval forward scores every rating by itself -- distinct-token
    projection, gather, one workgroup or wave per rating -- so larger slices of the same stream give the
    same scores with fewer launches and less Python per rating (``eval_batch_size`` overrides; the
    op-by-op module path keeps the reference's slices: its Linear layers are rocBLAS GEMMs over the
    batch).  TransNet's MSE_right / MSE_transform are means of per-slice means (eval.py:33-35): evaluate()
    takes them over the reference's slices INSIDE each launch, from the engine's per-rating values."""
    if engine is None or not getattr(reader, 'takes_batch', False):
        return None
    bsz = int(hyper_params['batch_size'])
    # the larger slices only pay while a launch is small: past EVAL_LAUNCH_CAP ratings the slice stays at
    # batch_size (the device batcher allocates n * (3 * doc + 20) * 8 bytes per launch and the engine keeps a
    # workspace per launch shape: a batch_size that trains must not run the validation pass out of memory)
    n = int(hyper_params.get('eval_batch_size') or min(factor * bsz, max(bsz, EVAL_LAUNCH_CAP)))
    if hyper_params['model_type'] in ['transnet', 'transnet++']:
        n = max(bsz, n - n % bsz)                            # launches must start on the reference's slice boundaries
    return n


def evaluate(model, criterion, reader, hyper_params, user_count, item_count, review, engine=None):
    metrics = {}
    big = _launch_size(reader, hyper_params, engine, EVAL_LAUNCH)
    batches = reader.iter(eval=True, batch=big) if big else reader.iter(eval=True)
    total_n, total_batches = 0.0, 0.0
    bsz = int(hyper_params['batch_size'])
    is_tn = hyper_params['model_type'] in ['transnet', 'transnet++']
    se_parts, user_parts, item_parts = [], [], []
    total_se = mse_right = conv_loss = None                  # device scalars: no sync inside the pass
    model.eval()
    with torch.no_grad():
        for data, y in batches:
            user, item = data[5], data[6]
            if engine is not None:
                output, mse = engine.predict(data, y)
                mse = mse.clone()                            # the engine reuses its output buffer
                if is_tn:                                    # TransNetEngine: per rating (target pred, its SE, transform)
                    aux = engine.aux(data)[:, 1:3]
                    n_here, full = aux.shape[0], aux.shape[0] // bsz
                    if full:                                 # the reference's slices of batch_size inside this launch
                        rc = aux[:full * bsz].reshape(full, bsz, 2).mean(dim=1).sum(dim=0)
                        mse_right = rc[0] if mse_right is None else mse_right + rc[0]
                        conv_loss = rc[1] if conv_loss is None else conv_loss + rc[1]
                    if n_here > full * bsz:                  # ... and the ragged last one
                        rc = aux[full * bsz:].mean(dim=0)
                        mse_right = rc[0] if mse_right is None else mse_right + rc[0]
                        conv_loss = rc[1] if conv_loss is None else conv_loss + rc[1]
                    total_batches += float(full + (n_here > full * bsz)) - 1.0      # (+ 1.0 below)
            else:
                output = model(data)
                if is_tn:
                    mse = criterion(output[0], y, return_mean=False).data
                    r, c = criterion(output[1], y).data, output[2].data
                    mse_right = r if mse_right is None else mse_right + r
                    conv_loss = c if conv_loss is None else conv_loss + c
                else:
                    mse = criterion(output, y, return_mean=False).data
            s = torch.sum(mse)
            total_se = s if total_se is None else total_se + s
            total_n += float(int(y.shape[0]))
            se_parts.append(mse.reshape(-1))
            user_parts.append(user.reshape(-1))
            item_parts.append(item.reshape(-1))
            total_batches += 1.0
        # ONE device -> host copy for the whole split (the reference reads two scalars per example,
        # eval.py:42-53, which after acceleration costs more than the epoch it follows)
        ses = torch.cat(se_parts).double().cpu().numpy() if se_parts else None
        metrics['MSE'] = round(float(total_se) / total_n, 4)
        if is_tn:
            metrics['MSE_right'] = round(float(mse_right) / total_batches, 4)
            metrics['MSE_transform'] = round(float(conv_loss) / total_batches, 4)
    if ses is None:
        return metrics, {}, {}
    users = torch.cat(user_parts).cpu().numpy()
    items = torch.cat(item_parts).cpu().numpy()
    return metrics, _count_maps(users, ses, user_count), _count_maps(items, ses, item_count)


def eval_ranking(model, reader, hyper_params, review=False, engine=None):
    """HR@1 (eval.py:64-92): % of ranking rows whose positive -- column 0 of the [B, 6] scores -- is what
    ``torch.topk(scores[row], k=1)`` returns.  The scores of the whole split stay on the device and are
    copied ONCE; the top-1 is then taken on the host by the same ATen routine the reference calls per
    row (batched here), so exact score ties resolve the way they do there (a device-side topk may order
    ties differently).  ``engine``: a native engine of `model` -- its fused eval forward scores the batches."""
    parts = []
    is_tn = hyper_params['model_type'] in ['transnet', 'transnet++']
    big = _launch_size(reader, hyper_params, engine, RANK_LAUNCH)
    with torch.no_grad():
        for data, y in (reader.iter_negs(review, batch=big) if big else reader.iter_negs(review)):
            if engine is not None:
                output = engine.predict(data, None)[0].clone()   # the engine reuses its output buffer
            else:
                output = model(data)
                if is_tn:
                    output = output[0]
            parts.append(output.reshape(-1, output.shape[-1]))
    scores = torch.cat(parts).float().cpu() if parts else torch.zeros(0, 6)
    # batched host topk == the reference's per-row call, ties included (probed: 10k rows of 3-valued scores)
    top = torch.topk(scores, k=1, dim=-1, sorted=True).indices[:, 0]
    hits, total = float((top == 0).sum()), float(scores.shape[0])
    return {'HR@1': round(100.0 * hits / total, 2)}        # no ranking rows: ZeroDivisionError, like eval.py:90
