"""reviews4rec_amd -- MI355X-native rating-prediction training path for the
review-aware recommenders of noveens/reviews4rec.

Layout
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/r4r.h)
  _lib.py          ctypes binding, signatures parsed from the header
  ops.py           torch.autograd glue over the C ABI
  optim.py         fused dense Adam with torch.optim.Adam's surface
  pytorch_models/  drop-in Model(hyper_params) classes (same names as the reference)
  main.py eval.py loss.py data_fast.py hyper_params.py utils.py   host-surface counterparts
  dist.py          one-process-per-GPU data parallelism over RCCL
"""
__version__ = '0.1.0'


def get_model_class(model_type):
    """Counterpart of the class selection at main.py:349-352."""
    if model_type in ('deepconn', 'deepconn++'):
        from .pytorch_models.DeepCoNN import DeepCoNN as Model
    elif model_type in ('transnet', 'transnet++'):
        from .pytorch_models.TransNet import TransNet as Model
    elif model_type == 'NARRE':
        from .pytorch_models.NARRE import NARRE as Model
    elif model_type in ('bias_only', 'MF', 'MF_dot'):
        from .pytorch_models.MF import MF as Model
    elif model_type == 'NeuMF':                      # GMF / MLP / NeuMF by hyper_params['neumf_stage']
        from .pytorch_models.NeuMF import build as Model
    else:
        raise ValueError('model_type %r is not on the accelerated path' % (model_type,))
    return Model
