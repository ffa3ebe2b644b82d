"""NARRE on the HIP path (counterpart of pytorch_models/NARRE.py)."""
import torch
import torch.nn as nn

from .. import ops
from ..utils import load_word_vectors
from .common_pytorch_models import TextCNN


class NARRE(nn.Module):
    def __init__(self, hyper_params):
        super(NARRE, self).__init__()
        self.hyper_params = hyper_params
        L = hyper_params['latent_size']
        U, I = hyper_params['total_users'], hyper_params['total_items']
        self.word2vec = nn.Embedding.from_pretrained(load_word_vectors(hyper_params))   # frozen
        self.user_embedding = nn.Embedding(U + 2, L)     # +2: the padding id total+1 (data.py:275-276)
        self.item_embedding = nn.Embedding(I + 2, L)
        self.user_conv = TextCNN(hyper_params)
        self.item_conv = TextCNN(hyper_params)
        self.user_conv.site, self.item_conv.site = 'user_conv.dropout', 'item_conv.dropout'

        def scorer():
            return nn.Sequential(nn.Linear(2 * L, L), nn.ReLU(), nn.Dropout(hyper_params['dropout']), nn.Linear(L, 1))
        self.attention_scorer_user = scorer()
        self.attention_scorer_item = scorer()
        self.final = nn.Sequential(nn.Dropout(hyper_params['dropout']), nn.Linear(L, L), nn.ReLU(), nn.Linear(L, 1))
        self.user_bias = nn.Parameter(torch.full((U + 2,), 0.1))
        self.item_bias = nn.Parameter(torch.full((I + 2,), 0.1))
        self.global_bias = nn.Parameter(torch.full((1,), 4.0))
        self.p = float(hyper_params['dropout'])

    def attention(self, x, other_x, scorer, site):
        return ops.narre_attention(x, other_x, scorer[0].weight, scorer[0].bias, scorer[3].weight, scorer[3].bias,
                                   self.p, self.training, site)

    def forward(self, data):
        _, users_who_reviewed, reviewed_items, user_reviews, item_reviews, user_id, item_id = data
        final_shape = tuple(user_id.shape)
        n = user_id.numel()
        R_u, W_u = user_reviews.shape[-2], user_reviews.shape[-1]
        R_i, W_i = item_reviews.shape[-2], item_reviews.shape[-1]
        users_who_reviewed = users_who_reviewed.reshape(n, -1)
        reviewed_items = reviewed_items.reshape(n, -1)
        uid, iid = user_id.reshape(-1), item_id.reshape(-1)
        table = self.word2vec.weight

        # one TextCNN pass per review: N = bsz * num_reviews documents of num_words tokens
        user = self.user_conv(user_reviews.reshape(n * R_u, W_u), table).view(n, R_u, -1)
        item = self.item_conv(item_reviews.reshape(n * R_i, W_i), table).view(n, R_i, -1)

        user = self.attention(user, ops.embed(self.item_embedding.weight, reviewed_items),
                              self.attention_scorer_user, 'attention_scorer_user.2')
        item = self.attention(item, ops.embed(self.user_embedding.weight, users_who_reviewed),
                              self.attention_scorer_item, 'attention_scorer_item.2')

        user_vec = ops.dropout(ops.embed(self.user_embedding.weight, uid), self.p, self.training, 'dropout.user')
        item_vec = ops.dropout(ops.embed(self.item_embedding.weight, iid), self.p, self.training, 'dropout.item')
        cat = ops.mul(ops.add(user, user_vec), ops.add(item, item_vec))

        h = ops.dropout(cat, self.p, self.training, 'final.0')
        h = ops.linear(h, self.final[1].weight, self.final[1].bias, relu=True)
        rating = ops.linear(h, self.final[3].weight, self.final[3].bias)[:, 0]
        return ops.bias_head(rating, self.user_bias, self.item_bias, self.global_bias, uid, iid).view(final_shape)
