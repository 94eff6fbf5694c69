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
        self.embed_reg = embed_reg
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

    def _cross_constants(self, dim):
        """Weight-only constants of the closed-form cross tower, cached per weight version:
        Wd = [w_0 .. w_{L-1}, w_c] (L+1, dim_padded), G_l = sum_{j<l} b_j . w_l, c = (sum_j b_j) . w_c + bias."""
        if not self.cross_network.built:
            self.cross_network.build(dim)
        if not self.dense_final.built:
            self.dense_final.build(dim + self.dnn_network.dnn_network[-1].units)
        key = (self.cross_network._version, self.dense_final._version)
        if getattr(self, '_cc', None) is None or self._cc[0] != key:
            cw, cb = self.cross_network._w['cross_weights'], self.cross_network._w['cross_bias']    # (L, dim)
            wf, bf = self.dense_final._w['kernel'], self.dense_final._w['bias']                       # (dim+H, 1), (1,)
            w_c, w_d = wf[:dim, 0], wf[dim:, :].contiguous()
            L = cw.shape[0]
            dimp = (dim + 3) // 4 * 4
            Wd = torch.zeros((L + 1, dimp), dtype=torch.float32, device=self.device)
            Wd[:L, :dim] = cw
            Wd[L, :dim] = w_c
            csum = torch.cumsum(cb, dim=0)                                  # prefix sums of the biases
            G = torch.zeros(L, dtype=torch.float32, device=self.device)
            if L > 1:
                G[1:] = (csum[:-1] * cw[1:]).sum(dim=1)                     # sum_{j<l} b_j . w_l
            c = float((csum[-1] * w_c).sum() + bf[0]) if L > 0 else float(bf[0])
            self._cc = (key, Wd.contiguous(), G, c, w_d)
        return self._cc[1:]

    def call(self, inputs, **kwargs):
        sparse_inputs = to_device_ids(inputs, self.device)
        dim = self._group.width
        L = self.layer_num
        if dim % 4 == 0 and 1 <= L <= 7 and (L + 1) * dim * 4 <= 64 * 1024 and len(set(self._group.dims)) == 1:
            # The cross tower in closed form needs only d_l = x0 . w_l (x_l = alpha_l x0 + sum_{j<l} b_j), and its
            # output meets nothing but the final Dense(1) (:55-56), so neither x_l nor the concat is materialised:
            # the dots ride along with the gather, the logit is alpha_L (x0 . w_c) + const + dnn_x . w_d.
            Wd, G, c, w_d = self._cross_constants(dim)
            B = sparse_inputs.shape[0]
            am = torch.empty(B, dtype=torch.float32, device=self.device) if B >= 1024 else None   # row maxima for the DNN
            x, dots = ops.gather_dots(self._group, sparse_inputs, Wd, row_absmax=am)   # :47 + :51 (+ the cross half of :56)
            dnn_part = ops.dense(self.dnn_network(x, row_absmax=am), w_d)    # :53 + the dnn half of :56
            return ops.dcn_logit(dots, G, c, dnn_part)
        return self._call_unfused(sparse_inputs)

    def _call_unfused(self, sparse_inputs):
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
            total_x = ops.concat_cols([total[:, :dim], self.dnn_network(x)])
        return ops.add_sigmoid(self.dense_final(total_x))                  # :56
