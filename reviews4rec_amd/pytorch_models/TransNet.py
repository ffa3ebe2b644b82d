"""TransNet / TransNet++ on the HIP path (counterpart of pytorch_models/TransNet.py)."""
import torch
import torch.nn as nn

from .. import ops
from ..utils import load_word_vectors, xavier_init
from .common_pytorch_models import TextCNN, TorchFM


class Source(nn.Module):
    def __init__(self, hyper_params):
        super(Source, self).__init__()
        self.hyper_params = hyper_params
        L = hyper_params['latent_size']
        self.user_conv = TextCNN(hyper_params)
        self.item_conv = TextCNN(hyper_params)
        self.user_conv.site, self.item_conv.site = 'source.user_conv.dropout', 'source.item_conv.dropout'
        self.project = nn.Sequential(nn.Linear(2 * L, L), nn.ReLU(), nn.Linear(L, L))
        self.p = float(hyper_params['dropout'])

    def forward(self, user_idx, item_idx, table):
        user = self.user_conv(user_idx, table)
        item = self.item_conv(item_idx, table)
        cat = torch.cat([user, item], dim=-1)
        h = ops.linear(cat, self.project[0].weight, self.project[0].bias, relu=True)
        temp = ops.linear(h, self.project[2].weight, self.project[2].bias)
        self.ir = ops.dropout(temp, self.p, self.training, 'source.dropout')
        return None


class Target(nn.Module):
    def __init__(self, hyper_params):
        super(Target, self).__init__()
        self.hyper_params = hyper_params
        self.word2vec = nn.Embedding.from_pretrained(load_word_vectors(hyper_params))   # frozen
        self.conv = TextCNN(hyper_params)
        self.conv.site = 'target.conv.dropout'
        self.fm = TorchFM(hyper_params['latent_size'], 8)
        self.p = float(hyper_params['dropout'])

    def forward(self, this_idx):
        this = self.conv(this_idx, self.word2vec.weight)
        self.ir = ops.dropout(this, self.p, self.training, 'target.dropout')
        return self.fm(self.ir)


class TransNet(nn.Module):
    def __init__(self, hyper_params):
        super(TransNet, self).__init__()
        self.hyper_params = hyper_params
        self.target = Target(hyper_params)
        xavier_init(self.target)                          # TransNet.py:69
        self.source = Source(hyper_params)
        xavier_init(self.source)                          # TransNet.py:72
        L = hyper_params['latent_size']
        if hyper_params['model_type'] == 'transnet++':
            self.user_embedding = nn.Embedding(hyper_params['total_users'] + 2, 5)
            self.item_embedding = nn.Embedding(hyper_params['total_items'] + 2, 5)
            self.source_fm = TorchFM(10 + L, 8)
        else:
            self.source_fm = TorchFM(L, 8)
        self.p = float(hyper_params['dropout'])

    def forward(self, data):
        this_reviews, _, _, user_reviews, item_reviews, user_id, item_id = data
        final_shape = tuple(user_id.shape)
        n = user_id.numel()
        self.source(user_reviews.reshape(n, -1), item_reviews.reshape(n, -1), self.target.word2vec.weight)
        if self.hyper_params['model_type'] == 'transnet++':
            uid, iid = user_id.reshape(-1), item_id.reshape(-1)
            u = ops.dropout(ops.embed(self.user_embedding.weight, uid), self.p, self.training, 'dropout.user')
            i = ops.dropout(ops.embed(self.item_embedding.weight, iid), self.p, self.training, 'dropout.item')
            final = torch.cat([u, i, self.source.ir], dim=-1)
        else:
            final = self.source.ir
        source_out = self.source_fm(final)
        target_out = self.target(this_reviews.reshape(n, -1))
        return [
            source_out[:, 0].view(final_shape),
            target_out[:, 0].view(final_shape),
            ops.transform_loss(self.source.ir, self.target.ir),
        ]
