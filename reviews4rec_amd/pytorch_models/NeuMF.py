"""GMF / MLP / NeuMF on the HIP path (counterpart of pytorch_models/NeuMF.py).

Same state_dict keys, constructor defaults and forward(data) contract as the reference
classes; every op is one of the package's HIP kernels (ID gathers, Philox dropout, elementwise
product, small dense layers, the fused bias head).  ``main.main_NeuMF`` is the reference's
three-stage schedule: pre-train GMF, pre-train MLP, initialise NeuMF from both
(``NeuMF.init``, NeuMF.py:100-118), train it.
"""
import torch
import torch.nn as nn

from .. import ops


def _biases(module, hyper_params):
    U, I = hyper_params['total_users'], hyper_params['total_items']
    module.user_bias = nn.Parameter(torch.full((U + 1,), 0.1))       # NeuMF.py:15-17
    module.item_bias = nn.Parameter(torch.full((I + 1,), 0.1))
    module.global_bias = nn.Parameter(torch.full((1,), 4.0))
    return U, I


def _project(hyper_params):
    L = hyper_params['latent_size']
    return nn.Sequential(nn.Dropout(hyper_params['dropout']), nn.Linear(2 * L, L), nn.ReLU(), nn.Linear(L, L))


def _run_project(seq, x, p, training):
    h = ops.dropout(x, p, training, 'project.0')
    h = ops.linear(h, seq[1].weight, seq[1].bias, relu=True)
    return ops.linear(h, seq[3].weight, seq[3].bias)


class GMF(nn.Module):
    def __init__(self, hyper_params):
        super(GMF, self).__init__()
        self.hyper_params = hyper_params
        U, I = _biases(self, hyper_params)
        L = hyper_params['latent_size']
        self.user_embedding = nn.Embedding(U + 1, L)
        self.item_embedding = nn.Embedding(I + 1, L)
        self.final = nn.Linear(L, 1)
        self.dropout = nn.Dropout(hyper_params['dropout'])           # container parity only
        self.p = float(hyper_params['dropout'])

    def forward(self, data):
        user_id, item_id = data[5], data[6]
        shape = user_id.shape
        uid, iid = user_id.reshape(-1), item_id.reshape(-1)
        user = ops.dropout(ops.embed(self.user_embedding.weight, uid), self.p, self.training, 'dropout.user')
        item = ops.dropout(ops.embed(self.item_embedding.weight, iid), self.p, self.training, 'dropout.item')
        rating = ops.linear(ops.mul(user, item), self.final.weight, self.final.bias)[:, 0]
        return ops.bias_head(rating, self.user_bias, self.item_bias, self.global_bias, uid, iid).view(shape)


class MLP(nn.Module):
    def __init__(self, hyper_params):
        super(MLP, self).__init__()
        self.hyper_params = hyper_params
        U, I = _biases(self, hyper_params)
        L = hyper_params['latent_size']
        self.user_embedding = nn.Embedding(U + 1, L)
        self.item_embedding = nn.Embedding(I + 1, L)
        self.project = _project(hyper_params)
        self.final = nn.Linear(L, 1)
        self.dropout = nn.Dropout(hyper_params['dropout'])
        self.p = float(hyper_params['dropout'])

    def forward(self, data):
        user_id, item_id = data[5], data[6]
        shape = user_id.shape
        uid, iid = user_id.reshape(-1), item_id.reshape(-1)
        user = ops.dropout(ops.embed(self.user_embedding.weight, uid), self.p, self.training, 'dropout.user')
        item = ops.dropout(ops.embed(self.item_embedding.weight, iid), self.p, self.training, 'dropout.item')
        joint = _run_project(self.project, torch.cat([user, item], dim=-1), self.p, self.training)
        rating = ops.linear(joint, self.final.weight, self.final.bias)[:, 0]
        return ops.bias_head(rating, self.user_bias, self.item_bias, self.global_bias, uid, iid).view(shape)


class NeuMF(nn.Module):
    def __init__(self, hyper_params):
        super(NeuMF, self).__init__()
        self.hyper_params = hyper_params
        U, I = _biases(self, hyper_params)
        L = hyper_params['latent_size']
        self.gmf_user_embedding = nn.Embedding(U + 1, L)
        self.gmf_item_embedding = nn.Embedding(I + 1, L)
        self.mlp_user_embedding = nn.Embedding(U + 1, L)
        self.mlp_item_embedding = nn.Embedding(I + 1, L)
        self.project = _project(hyper_params)
        self.final = nn.Linear(2 * L, 1)
        self.dropout = nn.Dropout(hyper_params['dropout'])
        self.p = float(hyper_params['dropout'])

    def init(self, gmf_model, mlp_model):
        """NeuMF.py:100-118: embeddings and the MLP tower copied, the two `final` layers
        concatenated, their biases and the user / item bias vectors averaged."""
        with torch.no_grad():
            self.gmf_user_embedding.weight.copy_(gmf_model.user_embedding.weight)
            self.gmf_item_embedding.weight.copy_(gmf_model.item_embedding.weight)
            self.mlp_user_embedding.weight.copy_(mlp_model.user_embedding.weight)
            self.mlp_item_embedding.weight.copy_(mlp_model.item_embedding.weight)
            for i in (1, 3):
                self.project[i].weight.copy_(mlp_model.project[i].weight)
                self.project[i].bias.copy_(mlp_model.project[i].bias)
            self.final.weight.copy_(torch.cat([gmf_model.final.weight, mlp_model.final.weight], dim=-1))
            self.final.bias.copy_(0.5 * (gmf_model.final.bias + mlp_model.final.bias))
            self.user_bias.copy_(0.5 * (gmf_model.user_bias + mlp_model.user_bias))
            self.item_bias.copy_(0.5 * (gmf_model.item_bias + mlp_model.item_bias))

    def forward(self, data):
        user_id, item_id = data[5], data[6]
        shape = user_id.shape
        uid, iid = user_id.reshape(-1), item_id.reshape(-1)
        p, tr = self.p, self.training
        gmf = ops.mul(ops.dropout(ops.embed(self.gmf_user_embedding.weight, uid), p, tr, 'dropout.gmf_user'),
                      ops.dropout(ops.embed(self.gmf_item_embedding.weight, iid), p, tr, 'dropout.gmf_item'))
        mlp = torch.cat([ops.dropout(ops.embed(self.mlp_user_embedding.weight, uid), p, tr, 'dropout.mlp_user'),
                         ops.dropout(ops.embed(self.mlp_item_embedding.weight, iid), p, tr, 'dropout.mlp_item')],
                        dim=-1)
        mlp = _run_project(self.project, mlp, p, tr)
        rating = ops.linear(torch.cat([gmf, mlp], dim=-1), self.final.weight, self.final.bias)[:, 0]
        return ops.bias_head(rating, self.user_bias, self.item_bias, self.global_bias, uid, iid).view(shape)


def build(hyper_params):
    """`get_model_class('NeuMF')`: the class of hyper_params['neumf_stage'] ('GMF' / 'MLP' /
    'NeuMF', default 'NeuMF') -- the reference uses one model_type for all three stages."""
    return {'GMF': GMF, 'MLP': MLP, 'NeuMF': NeuMF}[hyper_params.get('neumf_stage', 'NeuMF')](hyper_params)
