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
        cross_x = self.cross_network(x)                                    # :51
        dnn_x = self.dnn_network(x)                                        # :53
        total_x = torch.cat([cross_x, dnn_x], dim=-1)                      # :55
        return ops.add_sigmoid(self.dense_final(total_x))                  # :56
