"""Mirror of src/ctr/dcn/model.py (Deep & Cross) on the HIP kernels.  `feature_columns` is the
sparse list only (dcn/model.py:31); layer_num = len(hidden_units) (:32)."""
import torch

from ctr.layers.modules import CrossNetwork, DNN
from recamd import nn, ops
from recamd.nn import Model, to_device_ids


class DCN(Model):
    def __init__(self, feature_columns, hidden_units, activation='relu',
                 dnn_dropout=0., embed_reg=1e-6, cross_w_reg=1e-6, cross_b_reg=1e-6):
        super().__init__()
        self.sparse_feature_columns = feature_columns
        self.layer_num = len(hidden_units)
        self.embed_layers = {
            'embed_' + str(i): self.track('embed_' + str(i), nn.Embedding(
                input_dim=feat['feat_num'], input_length=1, output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for i, feat in enumerate(self.sparse_feature_columns)
        }
        self.cross_network = CrossNetwork(self.layer_num, cross_w_reg, cross_b_reg)
        self.dnn_network = DNN(hidden_units, activation, dnn_dropout)
        self.dense_final = nn.Dense(1, activation=None)
        self._group = ops.TableGroup([self.embed_layers['embed_%d' % i].table
                                      for i in range(len(self.sparse_feature_columns))])

    def call(self, inputs, **kwargs):
        sparse_inputs = to_device_ids(inputs, self.device)
        x = ops.gather_concat(self._group, sparse_inputs)                  # dcn/model.py:47
        # tf.concat([cross_x, dnn_x]) (:55) without a copy: both producers write straight into one
        # (B, dim + H) buffer (16-B aligned column offsets)
        B, dim = x.shape
        H = self.dnn_network.dnn_network[-1].units
        pad = (-dim) % 4
        total = torch.empty((B, dim + pad + H + ((-H) % 4)), dtype=torch.float32, device=self.device)
        if not self.cross_network.built:
            self.cross_network.build(dim)
        cw = self.cross_network._w
        ops.cross_network(x, cw['cross_weights'], cw['cross_bias'], out=total[:, :dim])     # :51
        if pad == 0:
            self.dnn_network(x, out=total[:, dim:dim + H])                                    # :53
            total_x = total[:, :dim + H]
        else:
            total_x = torch.cat([total[:, :dim], self.dnn_network(x)], dim=-1)
        return ops.add_sigmoid(self.dense_final(total_x))                  # :56
