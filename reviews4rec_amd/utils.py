"""Host-side helpers (counterpart of the reference's utils.py).

Parity-relevant pieces: ``xavier_init`` (utils.py:65-68: xavier-uniform on every
parameter with dim > 1, INCLUDING the frozen word table and FM ``V``) and
``init_transnet_optim`` (utils.py:70-92: the parameter grouping of TransNet's
four optimisers).
"""
import json
import os
import pickle

import numpy as np
import torch

is_cuda_available = torch.cuda.is_available()


def device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def LongTensor(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.int64).to(device())


def FloatTensor(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(device())


def save_obj(obj, name):
    with open(name + '.pkl', 'wb') as f:
        pickle.dump(obj, f, 2)


def load_obj(name):
    with open(name + '.pkl', 'rb') as f:
        return pickle.load(f)


def save_obj_json(obj, name):
    with open(name + '.json', 'w') as f:
        json.dump(obj, f)


def load_obj_json(name):
    with open(name + '.json', 'r') as f:
        return json.load(f)


def load_word_vectors(hyper_params):
    """The word table the models freeze.  Same source as the reference
    (``data_dir + '/word2vec'`` pickle of a V x E list, DeepCoNN.py:14); a
    ``word2vec.npy`` next to it, or an in-memory ``hyper_params['word_vectors']``
    (namespaced extra for synthetic runs), is accepted too."""
    if hyper_params.get('word_vectors') is not None:
        # always a private copy: xavier_init later overwrites the table in place
        return torch.tensor(np.asarray(hyper_params['word_vectors']), dtype=torch.float32)
    base = hyper_params['data_dir'] + '/word2vec'
    if os.path.exists(base + '.npy'):
        return torch.tensor(np.load(base + '.npy'), dtype=torch.float32)
    return torch.tensor(np.asarray(load_obj(base), dtype=np.float32))


def load_user_item_counts(hyper_params):
    user_count = load_obj(hyper_params['data_dir'] + 'user_count')
    item_count = load_obj(hyper_params['data_dir'] + 'item_count')
    return user_count, item_count


def file_write(log_file, s, dont_print=False):
    if not dont_print:
        print(s)
    with open(log_file, 'a') as f:
        f.write(s + '\n')


def clear_log_file(log_file):
    open(log_file, 'w').close()


def pretty_print(h):
    print('{')
    for key in h:
        print(' ' * 4 + str(key) + ': ' + h[key])
    print('}\n')


def log_end_epoch(hyper_params, metrics, epoch, time_elpased, metrics_on='(VAL)'):
    string2 = ''.join(' | ' + m + ' = ' + str(metrics[m]) for m in metrics) + ' ' + metrics_on
    ss = '-' * 89
    ss += '\n| end of epoch {} | time: {:5.2f}s'.format(epoch, time_elpased) + string2 + '\n'
    ss += '-' * 89
    file_write(hyper_params['log_file'], ss)


def xavier_init(model):
    for p in model.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)


def init_transnet_optim(hyper_params, model, Optim=None):
    """[optimizer_source, optimizer_source_fm, optimizer_target, optimizer_all] with the
    reference's parameter grouping (utils.py:70-92)."""
    if Optim is None:
        from .optim import Adam as Optim
    kw = dict(lr=hyper_params['lr'], weight_decay=hyper_params['weight_decay'])
    optimizer_source = Optim(model.source.parameters(), **kw)
    fm_params = list(model.source_fm.parameters())
    if hyper_params['model_type'] == 'transnet++':
        fm_params += [model.user_embedding.weight, model.item_embedding.weight]
    optimizer_source_fm = Optim(fm_params, **kw)
    optimizer_target = Optim(model.target.parameters(), **kw)
    optimizer_all = Optim(model.parameters(), **kw)
    return [optimizer_source, optimizer_source_fm, optimizer_target, optimizer_all]
