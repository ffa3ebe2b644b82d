"""Evaluation (counterpart of the reference's eval.py).

``evaluate`` returns exactly what eval.py:11-62 returns -- ``{'MSE': round(sum SE / N,
4)}`` (+ MSE_right / MSE_transform for TransNet) and the two train-frequency -> [SE]
maps -- but builds the maps from one device->host copy per batch instead of the
reference's two scalar reads per example (eval.py:42-53).  ``eval_ranking`` is HR@1
over negatives-shaped batches (eval.py:64-92).
"""
import torch


def evaluate(model, criterion, reader, hyper_params, user_count, item_count, review, engine=None):
    metrics = {}
    total_se, total_n, total_batches = 0.0, 0.0, 0.0
    mse_right, conv_loss = 0.0, 0.0
    user_count_mse_map, item_count_mse_map = {}, {}
    is_tn = hyper_params['model_type'] in ['transnet', 'transnet++']
    model.eval()
    with torch.no_grad():
        for data, y in reader.iter(eval=True):
            user, item = data[5], data[6]
            if engine is not None:
                output, mse = engine.predict(data, y)
            else:
                output = model(data)
                if is_tn:
                    mse = criterion(output[0], y, return_mean=False).data
                    mse_right += float(criterion(output[1], y).data)
                    conv_loss += float(output[2].data)
                else:
                    mse = criterion(output, y, return_mean=False).data
            total_se += float(torch.sum(mse))
            total_n += float(int(y.shape[0]))
            users, items, ses = user.reshape(-1).tolist(), item.reshape(-1).tolist(), mse.reshape(-1).tolist()
            for user_id, item_id, se in zip(users, items, ses):
                if user_id not in user_count:
                    user_count[user_id] = 0
                if item_id not in item_count:
                    item_count[item_id] = 0
                user_count_mse_map.setdefault(user_count[user_id], []).append(se)
                item_count_mse_map.setdefault(item_count[item_id], []).append(se)
            total_batches += 1.0
        metrics['MSE'] = round(total_se / total_n, 4)
        if is_tn:
            metrics['MSE_right'] = round(mse_right / total_batches, 4)
            metrics['MSE_transform'] = round(conv_loss / total_batches, 4)
    return metrics, user_count_mse_map, item_count_mse_map


def eval_ranking(model, reader, hyper_params, review=False):
    """HR@1: % of rows whose positive (column 0 of the [B, 6] scores) ranks first."""
    hits, total = 0.0, 0.0
    is_tn = hyper_params['model_type'] in ['transnet', 'transnet++']
    with torch.no_grad():
        for data, y in reader.iter_negs(review):
            output = model(data)
            if is_tn:
                output = output[0]
            top = torch.topk(output, k=1, dim=-1, sorted=True).indices[:, 0]
            hits += float((top == 0).sum())
            total += float(top.numel())
    return {'HR@1': round(100.0 * hits / total, 2)}
