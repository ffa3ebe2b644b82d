"""Configuration (counterpart of the reference's hyper_params.py): the same keys
with the same defaults (hyper_params.py:50-80) and the same derived paths
(get_common_path, hyper_params.py:3-48; log_file / model_path / data_dir,
:82-95).  Differences, both forced by defects in the reference:
``only_reviews`` exists (the reference reads it for NARRE but never defines it),
and importing this module has no filesystem side effects -- directories are
created by ``finalize()``.

Namespaced extras understood by this package (all default to reference behaviour):
  engine           'auto' (default: the fused native step -- every model family has one -- and, where a configuration
                   exceeds one of its limits, the op-by-op step captured into one hipGraph, with a warning that says
                   why) | 'native' (native or an error) | 'graph' | 'module' (op-by-op autograd, eager)
  word_vectors     in-memory V x E table instead of data_dir/word2vec.pkl (synthetic runs)
  seed             dropout Philox seed
  eval_batch_size  ratings per launch of a validation pass scored by a native engine (default 8 x batch_size; 4 x
                   batch_size rows for HR@1): same scores, fewer launches (eval.py)
  checkpoint_path  epoch-level resume file of main.train_complete (absent upstream)
  sweep_period     1 .. 8 (default 8): visit period of the temporally blocked Adam sweeps over the ID tables of MF_dot /
                   MF / NeuMF / TransNet++ in main.train's loop (same weights and moments as the dense sweep, bit for
                   bit; 1 = the dense sweep every step; the environment's R4R_SWEEP_PERIOD overrides it)
"""
import os


def get_common_path(hyper_params):
    method, fm = hyper_params['model_type'], True
    if method == 'deepconn++':
        method, fm = 'deepconn', False
    parts = [str(method), str(hyper_params['dataset']), str(hyper_params['k_core']) + '_core_']
    mt = hyper_params['model_type']

    def add(label, key):
        parts.append(label + '_' + str(hyper_params[key]))

    if mt in ['MF', 'MF_dot', 'NeuMF']:
        add('_latent_size', 'latent_size')
    elif mt == 'HFT':
        add('_latent_size', 'latent_size')
        add('percent_reviews', 'percent_reviews_to_keep')
    elif mt in ['deepconn', 'deepconn++', 'transnet', 'transnet++']:
        add('_word_embed_size', 'word_embed_size')
        add('latent_size', 'latent_size')
        add('percent_reviews', 'percent_reviews_to_keep')
        parts.append('fm_' + str(fm))
    elif mt == 'NARRE':
        add('_num_reviews', 'narre_num_reviews')
        add('num_words', 'narre_num_words')
        add('word_embed_size', 'word_embed_size')
        add('latent_size', 'latent_size')
        add('only_reviews', 'only_reviews')
        add('percent_reviews', 'percent_reviews_to_keep')
    elif mt == 'MPCN':
        add('_latent_size', 'latent_size')
        add('percent_reviews', 'percent_reviews_to_keep')
        return '_'.join(parts).replace('_core___', '_core__')
    add('wd', 'weight_decay')
    add('lr', 'lr')
    add('dropout', 'dropout')
    add('input_len', 'input_length')
    return '_'.join(parts).replace('_core___', '_core__')


def default_hyper_params():
    return {
        'dataset': 'InstantVideo',
        'k_core': 5,
        'percent_reviews_to_keep': 100,
        'weight_decay': float(1e-6),
        'lr': 0.002,
        'epochs': 2,
        'batch_size': 128,
        'shuffle_data_every_epoch': False,     # dead key in the reference too (never read)
        'latent_size': 10,
        'word_embed_size': 64,
        'input_length': 1000,
        'dropout': 0.6,
        'model_type': 'bias_only',
        'lamda': 0.1,
        'latent_reg': 0.0,
        'narre_num_reviews': 10,
        'narre_num_words': 100,
        'only_reviews': False,
    }


def finalize(hyper_params, make_dirs=True, root='.'):
    """Derive common_path / log_file / model_path / data_dir like hyper_params.py:82-95."""
    hp = hyper_params
    hp.setdefault('only_reviews', False)
    common_path = get_common_path(hp)
    hp['common_path'] = common_path
    hp['log_file'] = os.path.join(root, 'saved_logs', common_path)
    hp['model_path'] = os.path.join(root, 'saved_models', common_path)
    if make_dirs:
        os.makedirs(os.path.join(root, 'saved_logs'), exist_ok=True)
        os.makedirs(os.path.join(root, 'saved_models'), exist_ok=True)
    if 'data_dir' not in hp:
        d = 'data/' + hp['dataset'] + '/' + str(hp['k_core']) + '_core/'
        if hp['percent_reviews_to_keep'] != 100:
            d += str(hp['percent_reviews_to_keep']) + '_percent/'
        hp['data_dir'] = d
    return hp


hyper_params = default_hyper_params()
