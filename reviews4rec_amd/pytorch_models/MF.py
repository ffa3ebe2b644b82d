"""bias_only / MF_dot / MF on the HIP path (counterpart of pytorch_models/MF.py)."""
import torch
import torch.nn as nn

from .. import ops
from .common_pytorch_models import TorchFM


class MF(nn.Module):
    def __init__(self, hyper_params):
        super(MF, self).__init__()
        self.hyper_params = hyper_params
        U, I = hyper_params['total_users'], hyper_params['total_items']
        self.user_bias = nn.Parameter(torch.full((U + 1,), 0.1))     # MF.py:14-16
        self.item_bias = nn.Parameter(torch.full((I + 1,), 0.1))
        self.global_bias = nn.Parameter(torch.full((1,), 4.0))
        self.p = float(hyper_params['dropout'])
        if hyper_params['model_type'] in ['MF', 'MF_dot']:
            latent_size = hyper_params['latent_size']
            self.user_embedding = nn.Embedding(U + 1, latent_size)
            self.item_embedding = nn.Embedding(I + 1, latent_size)
            self.dropout = nn.Dropout(hyper_params['dropout'])       # container parity only
        if hyper_params['model_type'] == 'MF':
            self.projection = nn.Sequential(
                nn.Dropout(hyper_params['dropout']),
                nn.Linear(2 * latent_size, latent_size),
                nn.ReLU(),
                nn.Linear(latent_size, latent_size)
            )
            self.final = TorchFM(2 * latent_size, latent_size)

    def forward(self, data):
        user_id, item_id = data[5], data[6]
        shape = user_id.shape
        uid, iid = user_id.reshape(-1), item_id.reshape(-1)
        mt = self.hyper_params['model_type']
        if mt == 'bias_only':
            return ops.bias_head(None, self.user_bias, self.item_bias, self.global_bias, uid, iid).view(shape)

        user = ops.dropout(ops.embed(self.user_embedding.weight, uid), self.p, self.training, 'dropout.user')
        item = ops.dropout(ops.embed(self.item_embedding.weight, iid), self.p, self.training, 'dropout.item')
        if mt == 'MF_dot':
            rating = ops.rowdot(user, item)
        else:
            cat = ops.dropout(torch.cat([user, item], dim=-1), self.p, self.training, 'projection.0')
            h = ops.linear(cat, self.projection[1].weight, self.projection[1].bias, relu=True)
            mlp_vector = ops.linear(h, self.projection[3].weight, self.projection[3].bias)
            cat = torch.cat([mlp_vector, ops.mul(user, item)], dim=-1)
            rating = self.final(cat)[:, 0]
        return ops.bias_head(rating, self.user_bias, self.item_bias, self.global_bias, uid, iid).view(shape)
