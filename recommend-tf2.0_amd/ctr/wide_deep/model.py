"""Mirror of src/ctr/wide_deep/model.py (Wide & Deep) on the HIP kernels.  The file defines its own `Linear`
(:15-23) and a BatchNorm-free `DNN` (:26-46); `feature_columns = [dense_feature_columns, sparse_feature_columns]`."""
import torch

from recamd import nn, ops
from recamd.nn import Model, to_device_f32, to_device_ids


class Linear(nn.Layer):
    """wide_deep/model.py:15-23: Dense(1)."""

    def __init__(self):
        super().__init__()
        self.dense = self.track('dense', nn.Dense(1, activation=None))

    def call(self, inputs, **kwargs):
        return self.dense(inputs)


class DNN(nn.Layer):
    """wide_deep/model.py:26-46: Dense stack + Dropout (no BatchNormalization, unlike ctr.layers.modules.DNN)."""

    def __init__(self, hidden_units, activation='relu', dropout=0.):
        super().__init__()
        self.dnn_network = [self.track(f'dense_{i}', nn.Dense(units=unit, activation=activation))
                            for i, unit in enumerate(hidden_units)]
        self.dropout = nn.Dropout(dropout)
        self._padded = None

    def call(self, inputs, tail_pad=0, **kwargs):
        """tail_pad > 0: `inputs` carries that many ZERO columns behind the features (a row stride rounded up so that
        K % 32 == 0); the first layer's kernel gets as many zero rows, the product is unchanged."""
        first = None
        dnn = self.dnn_network[0]
        if tail_pad:
            if not dnn.built:
                dnn.build(inputs.shape[-1] - tail_pad)
            key = (dnn._version, tail_pad)
            if self._padded is None or self._padded[0] != key:
                W = dnn._w['kernel']
                z = torch.zeros((tail_pad, W.shape[1]), dtype=W.dtype, device=W.device)
                self._padded = (key, torch.cat([W, z], dim=0).contiguous())
            first = (self._padded[1], dnn._w.get('bias'))
        x = nn.dense_chain(self.dnn_network, inputs, first=first)
        return self.dropout(x)


class WideDeep(Model):
    def __init__(self, feature_columns, hidden_units, activation='relu', dnn_dropout=0., embed_reg=1e-4):
        super().__init__()
        self.dense_feature_columns, self.sparse_feature_columns = feature_columns
        self.embed_reg = embed_reg
        self.embed_layers = {
            'embed_' + str(i): self.track('embed_' + str(i), nn.Embedding(
                input_dim=feat['feat_num'], input_length=1, output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for i, feat in enumerate(self.sparse_feature_columns)
        }
        self.dnn_network = DNN(hidden_units, activation, dnn_dropout)
        self.linear = Linear()
        self.final_dense = nn.Dense(1, activation=None)
        self._group = ops.TableGroup([self.embed_layers['embed_%d' % i].table
                                      for i in range(len(self.sparse_feature_columns))])

    def call(self, inputs, **kwargs):
        dense_inputs, sparse_inputs = inputs
        dense_inputs = to_device_f32(dense_inputs, self.device)
        sparse_inputs = to_device_ids(sparse_inputs, self.device)
        # x = concat([sparse_embed, dense_inputs]) (:70): the gather writes straight into the concat buffer
        B, nd, We = dense_inputs.shape[0], dense_inputs.shape[1], self._group.width
        # row stride padded to a multiple of 32 floats: 16-B aligned rows keep the gather on its vector path (a tight
        # 3341-float stride halves the gather rate) and K % 32 == 0 puts the first Dense on its hand-counted kernel;
        # the pad columns are zero and meet zero rows of the folded kernel
        wide = (We + nd + 31) // 32 * 32
        tail = wide - (We + nd)
        x = torch.empty((B, wide), dtype=torch.float32, device=self.device)
        if tail:
            x[:, We + nd:] = 0.0
        ops.gather_concat(self._group, sparse_inputs, out=x)               # :68-69
        ops.copy_cols(dense_inputs, x[:, We:We + nd])                    # tf.concat part written at its column offset
        wide_out = self.linear(dense_inputs)                               # :73
        deep_out = self.final_dense(self.dnn_network(x, tail_pad=tail))    # :75-76
        return ops.axpby_act(wide_out, deep_out, 0.5, 0.5, 'sigmoid')      # :78
