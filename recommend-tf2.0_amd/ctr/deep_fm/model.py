"""Mirror of src/ctr/deep_fm/model.py (DeepFM) on the HIP kernels.

Data layout: ONE (B, pad + nd + sum D) buffer holds `embeds = concat([dense, sparse_embed])`
(deep_fm/model.py:56) without a concat pass: the gather kernel writes the sparse part at a
16-B aligned column, `embeds` and `sparse_embed` are views of it."""
import torch

from ctr.layers.modules import DNN, FM
from recamd import nn, ops
from recamd.nn import Model, to_device_f32, to_device_ids


class DeepFM(Model):
    def __init__(self, feature_columns, hidden_units=(128, 64, 32), dnn_dropout=0.,
                 activation='relu', fm_w_reg=1e-6, embed_reg=1e-6):
        super().__init__()
        self.dense_feature_columns, self.sparse_feature_columns = feature_columns
        self.embed_reg = embed_reg
        self.embed_layers = {
            'embed_' + str(i): self.track('embed_' + str(i), nn.Embedding(
                input_dim=feat['feat_num'], input_length=1, output_dim=feat['embed_dim'],
                embeddings_initializer='random_normal'))                       # deep_fm/model.py:35
            for i, feat in enumerate(self.sparse_feature_columns)
        }
        self.nd = len(self.dense_feature_columns)
        self.sparse_width = sum(feat['embed_dim'] for feat in self.sparse_feature_columns)
        self.feature_length = self.nd + self.sparse_width
        self.embed_dim = self.sparse_feature_columns[0]['embed_dim']
        self.fm = FM(self.feature_length, fm_w_reg)
        self.dnn = DNN(hidden_units, activation, dnn_dropout)
        self.dense = nn.Dense(1, activation=None)
        self.pad = (-self.nd) % 4
        cols, c = [], self.pad + self.nd
        for feat in self.sparse_feature_columns:
            cols.append(c)
            c += feat['embed_dim']
        # row stride rounded up so the DNN's first GEMM reads K % 32 == 0 (the hand-counted Dense kernel): at most 31
        # zero columns behind the features, matched by zero rows in the folded kernel
        self.width = (c + 31) // 32 * 32
        self.tail = self.width - c
        self._group = ops.TableGroup([self.embed_layers['embed_%d' % i].table
                                      for i in range(len(self.sparse_feature_columns))], out_cols=cols)
        dims = set(self._group.dims)
        d0 = self._group.dims[0]
        self._fused_fm = (len(dims) == 1 and d0 % 4 == 0 and (d0 // 4) & (d0 // 4 - 1) == 0 and d0 <= 256)
        self._w_pad = None

    def call(self, inputs, **kwargs):
        am = None
        dense_inputs, sparse_inputs = inputs
        dense_inputs = to_device_f32(dense_inputs, self.device)
        sparse_inputs = to_device_ids(sparse_inputs, self.device)
        B = sparse_inputs.shape[0]
        buf = torch.empty((B, self.width), dtype=torch.float32, device=self.device)
        if self.pad:
            buf[:, :self.pad] = 0.0                                             # pad columns of the dense block
        if self.tail:
            buf[:, self.width - self.tail:] = 0.0                               # pad columns behind the features
        buf[:, self.pad:self.pad + self.nd] = dense_inputs                      # dense part of the concat
        embeds = buf[:, self.pad:self.pad + self.feature_length]               # :56
        sparse_embed = buf[:, self.pad + self.nd:self.pad + self.feature_length]
        if self._fused_fm:
            # one pass: rows -> concat buffer, and the FM layer's sums on the fly (no re-read)
            key = self.fm._version
            if self._w_pad is None or self._w_pad[0] != key:
                wp = torch.zeros(self.pad + self.feature_length, dtype=torch.float32, device=self.device)
                wp[self.pad:] = self.fm._w['w'].reshape(-1)
                self._w_pad = (key, wp)
            # the same pass also delivers every row's largest magnitude: the DNN's first Dense scales by it (f16x2 kernel)
            am = torch.empty(B, dtype=torch.float32, device=self.device) if B >= 1024 else None
            fm_outputs = ops.gather_fm(self._group, sparse_inputs, buf[:, :self.pad + self.nd], self._w_pad[1],
                                       self.pad + self.nd, buf, row_absmax=am)   # :53 + :59
        else:
            ops.gather_concat(self._group, sparse_inputs, out=buf)              # :53 (sparse part)
            fm_outputs = self.fm([embeds, sparse_embed])                       # :59
        # the DNN reads the whole 16-B aligned buffer, zeroed pad columns included (zero rows in its folded kernel)
        deep_outputs = self.dense(self.dnn(buf, lead_pad=self.pad, tail_pad=self.tail, row_absmax=am))      # :61-62
        return ops.add_sigmoid(fm_outputs, deep_outputs)                       # :64
