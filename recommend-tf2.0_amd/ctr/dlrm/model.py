"""Mirror of src/ctr/dlrm/model.py on the HIP kernels.

The reference's DLRM.call cannot run: it reads `self.dense_inputs` (dlrm/model.py:44) and
`self.dnn_network` (:50), neither of which exists, and it has no interaction op — it only
concatenates (:48).  Two modes:
  interaction='cat' : the INTENDED form of the file: bot_dnn(dense) ; concat[sparse_embed,
                      dense_fea] ; top_dnn ; Dense(1) ; sigmoid.
  interaction='dot' : the pairwise-dot interaction of the paper the file cites (:7) — the
                      BASELINE headline path: one fused gather + pairwise-dot launch; needs
                      bot_dnn_hidden_units[-1] == embed_dim.  Not reference-pinned (the reference
                      has no such op); order of the dots: (i,j), i>j, row-major, X = [emb_0..emb_F-1,
                      dense_fea], output = concat[dots, dense_fea].
"""
import torch

from ctr.layers.modules import DNN
from recamd import nn, ops
from recamd.nn import Model, to_device_f32, to_device_ids


class DLRM(Model):
    def __init__(self, feature_columns, bot_dnn_hidden_units=[64, 32, 16], top_dnn_hidden_units=[128, 64],
                 activation='relu', dnn_dropout=0., embed_reg=1e-4, interaction='cat'):
        super().__init__()
        if interaction not in ('cat', 'dot'):
            raise ValueError("interaction must be 'cat' or 'dot'")
        self.interaction = interaction
        self._dot_buf = None
        self.embed_reg = embed_reg
        self.dense_feature_columns, self.sparse_feature_columns = feature_columns
        self.embed_layers = {
            'embed_' + str(i): self.track('embed_' + str(i), nn.Embedding(
                input_dim=feat['feat_num'], input_length=1, output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for i, feat in enumerate(self.sparse_feature_columns)
        }
        self.bot_dnn = DNN(bot_dnn_hidden_units, activation, dnn_dropout)
        self.top_dnn = DNN(top_dnn_hidden_units, activation, dnn_dropout)
        self.final_dense = nn.Dense(1, activation=None)
        self._group = ops.TableGroup([self.embed_layers['embed_%d' % i].table
                                      for i in range(len(self.sparse_feature_columns))])
        dims = set(self._group.dims)
        if interaction == 'dot' and (len(dims) != 1 or bot_dnn_hidden_units[-1] != self._group.dims[0]):
            raise ValueError("interaction='dot' needs one shared embed_dim equal to bot_dnn_hidden_units[-1]")

    def call(self, inputs, **kwargs):
        dense_inputs, sparse_inputs = inputs
        dense_inputs = to_device_f32(dense_inputs, self.device)
        sparse_inputs = to_device_ids(sparse_inputs, self.device)
        B = sparse_inputs.shape[0]
        tail_pad = 0
        if self.interaction == 'dot':
            dense_fea = self.bot_dnn(dense_inputs)                             # intended :44
            # the (B, P + Hb) result lives in a buffer whose row stride is rounded up to 4 floats; the buffer is kept
            # (zero-initialised once, the kernels never write anything but zeros into its pad columns), so the top MLP
            # can read the padded width with a zero weight row appended: its first GEMM then has aligned rows and, for
            # the DLRM shape (351 + 128 = 479 -> 480), K a multiple of 32
            F = len(self._group)
            width = (F + 1) * F // 2 + dense_fea.shape[1]
            wide = (width + 3) // 4 * 4
            if self._dot_buf is None or self._dot_buf.shape != (B, wide):
                self._dot_buf = torch.zeros((B, wide), dtype=torch.float32, device=self.device)
            x = ops.gather_pairwise_dot(self._group, sparse_inputs, dense_fea, append_dense=True,
                                        out=self._dot_buf[:, :width])
            if wide != width and x.data_ptr() == self._dot_buf.data_ptr():
                x, tail_pad = self._dot_buf, wide - width
        else:
            # tf.concat([sparse_embed, dense_fea]) (:48) without a copy: the gather writes the sparse part of one
            # (B, sum D + bot) buffer and the bottom MLP's last layer writes its tail (16-B aligned tail: directly;
            # otherwise through one strided copy kernel)
            W = self._group.width
            Hb = self.bot_dnn.dnn_network[-1].units
            buf = torch.empty((B, W + Hb + ((-(W + Hb)) % 4)), dtype=torch.float32, device=self.device)
            ops.gather_concat(self._group, sparse_inputs, out=buf)                 # :45
            if W % 4 == 0:
                self.bot_dnn(dense_inputs, out=buf[:, W:W + Hb])                   # intended :44
            else:
                ops.copy_cols(self.bot_dnn(dense_inputs), buf[:, W:])
            x = buf[:, :W + Hb]
        top = self.final_dense(self.top_dnn(x, tail_pad=tail_pad))             # intended :50-51
        return ops.add_sigmoid(top)                                            # :53
