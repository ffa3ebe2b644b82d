"""TextCNN tower and FM head on the HIP path.

Counterpart of the reference's pytorch_models/common_pytorch_models.py (same
class names and state_dict keys: ``convs.0.weight [100,1,3,E]``,
``convs.0.bias``, ``fc.weight``, ``fc.bias``; ``V``, ``lin.weight``,
``lin.bias``).  One interface difference, forced by the fusion: the reference's
``TextCNN.forward`` takes already-gathered word vectors ``[N, T, E]``
(common_pytorch_models.py:22); here it takes the token ids and the frozen
table, because the gather is fused into the convolution kernel and the
``[N, T, E]`` tensor never exists.
"""
import torch
import torch.nn as nn

from .. import ops


class TextCNN(nn.Module):
    def __init__(self, hyper_params, window_sizes=[3]):
        super(TextCNN, self).__init__()
        if list(window_sizes) != [3]:
            raise NotImplementedError('the HIP tower implements the reference\'s only window size, 3')
        self.hyper_params = hyper_params
        self.num_filters = 100                       # common_pytorch_models.py:11
        # nn.Conv2d / nn.Linear are used as parameter containers only (same ctor
        # init and state_dict layout as the reference); they are never called.
        self.convs = nn.ModuleList([
            nn.Conv2d(1, self.num_filters, [3, hyper_params['word_embed_size']], padding=(2, 0))
        ])
        self.fc = nn.Linear(self.num_filters, hyper_params['latent_size'])
        self.p = float(hyper_params['dropout'])
        self.site = None                             # set by the owning model, names the dropout site

    def pooled(self, idx, table):
        """idx [N, T] int64, table [V, E] -> (max-pooled relu(conv) [N, 100], argmax)."""
        conv = self.convs[0]
        return ops.TextCNNPool.apply(idx, table, conv.weight, conv.bias)

    def forward(self, idx, table):
        pooled, _ = self.pooled(idx, table)
        z = ops.linear(pooled, self.fc.weight, self.fc.bias)
        return ops.dropout(z, self.p, self.training, self.site)


class TorchFM(nn.Module):
    """Factorisation machine without a global bias (common_pytorch_models.py:41-57)."""

    def __init__(self, n=None, k=None):
        super().__init__()
        self.V = nn.Parameter(torch.randn(n, k), requires_grad=True)
        self.lin = nn.Linear(n, 1)

    def forward(self, x):
        # [N, 1] like the reference, whose callers take [:, 0]
        return ops.fm(x, self.V, self.lin.weight, self.lin.bias).unsqueeze(-1)
