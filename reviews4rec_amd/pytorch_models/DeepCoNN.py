"""DeepCoNN / DeepCoNN++ on the HIP path (counterpart of pytorch_models/DeepCoNN.py)."""
import torch
import torch.nn as nn

from .. import ops
from ..utils import load_word_vectors
from .common_pytorch_models import TextCNN, TorchFM


class DeepCoNN(nn.Module):
    def __init__(self, hyper_params):
        super(DeepCoNN, self).__init__()
        self.hyper_params = hyper_params
        self.word2vec = nn.Embedding.from_pretrained(load_word_vectors(hyper_params))   # frozen
        self.user_conv = TextCNN(hyper_params)
        self.item_conv = TextCNN(hyper_params)
        self.user_conv.site, self.item_conv.site = 'user_conv.dropout', 'item_conv.dropout'
        L = hyper_params['latent_size']
        self.final = nn.Sequential(
            nn.Linear(2 * L, L), nn.ReLU(), nn.Dropout(hyper_params['dropout']), nn.Linear(L, 1)
        )
        U, I = hyper_params['total_users'], hyper_params['total_items']
        self.user_bias = nn.Parameter(torch.full((U + 2,), 0.1))
        self.item_bias = nn.Parameter(torch.full((I + 2,), 0.1))
        self.global_bias = nn.Parameter(torch.full((1,), 4.0))
        self.fm = TorchFM(2 * L, 8)
        self.p = float(hyper_params['dropout'])

    def forward(self, data):
        user_reviews, item_reviews, user_id, item_id = data[3], data[4], data[5], data[6]
        final_shape = tuple(user_id.shape)                 # [B] or [B, 6] (negatives)
        first_dim = user_id.numel()
        table = self.word2vec.weight
        user = self.user_conv(user_reviews.reshape(first_dim, -1), table)
        item = self.item_conv(item_reviews.reshape(first_dim, -1), table)
        cat = torch.cat([user, item], dim=-1)
        if self.hyper_params['model_type'] == 'deepconn':
            fm_out = self.fm(cat)[:, 0]
            return ops.bias_head(fm_out, None, None, self.global_bias, None, None).view(final_shape)
        h = ops.linear(cat, self.final[0].weight, self.final[0].bias, relu=True)
        h = ops.dropout(h, self.p, self.training, 'final.2')
        rating = ops.linear(h, self.final[3].weight, self.final[3].bias)[:, 0]
        return ops.bias_head(rating, self.user_bias, self.item_bias, self.global_bias,
                             user_id.reshape(-1), item_id.reshape(-1)).view(final_shape)
