"""Mirror of src/ctr/autoint/model.py on the HIP kernels.

mode='as_written': reproduces the reference exactly, including its bug: the 2-D (B, F*D + nd)
  tensor is fed to a layer written for 3-D input, so `reshape([-1, q.shape[1], H, S])`
  (src/ctr/layers/modules.py:211-212) regroups S consecutive SAMPLES into one pseudo-sample of S
  pseudo-fields; head_num = 1 (autoint/model.py:40); output is (B/S, 1) and B must be a multiple of
  S = att_hidden_units.
mode='intended' (default): the (B, fields, D) interacting layers of the AutoInt paper, stacked
  `att_layer_num` times with `head_num` heads (BASELINE config 3); dense features join as fields
  through a learned (nd, D) embedding scaled by the feature value (`embed_dense=True`).
  Not reference-pinned: the reference file cannot express it.
"""
import torch

from ctr.layers.modules import MultiHeadAttention
from recamd import nn, ops
from recamd.nn import Model, to_device_f32, to_device_ids


class AutoInt(Model):
    def __init__(self, feature_columns, att_hidden_units, att_activation='relu',
                 dnn_dropout=0., embed_reg=1e-4, mode='intended', head_num=1, att_layer_num=1,
                 use_res=False, embed_dense=True, fused=True):
        super().__init__()
        self.fused = fused          # False: separate launches even where the one-launch kernel applies
        if mode not in ('intended', 'as_written'):
            raise ValueError("mode must be 'intended' or 'as_written'")
        self.mode = mode
        self.embed_reg = embed_reg
        self.dense_feature_columns, self.sparse_feature_columns = feature_columns
        self.embed_layers = {
            'embed_' + str(i): self.track('embed_' + str(i), nn.Embedding(
                input_dim=feat['feat_num'], input_length=1, output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for i, feat in enumerate(self.sparse_feature_columns)
        }
        self.att_hidden_units = att_hidden_units
        self.att_activation = att_activation
        self.nd = len(self.dense_feature_columns)
        self._group = ops.TableGroup([self.embed_layers['embed_%d' % i].table
                                      for i in range(len(self.sparse_feature_columns))])
        if mode == 'as_written':
            head_num, att_layer_num, use_res = 1, 1, False          # autoint/model.py:40
        self.head_num = head_num
        self.attention_layers = [self.track('attention_%d' % i, MultiHeadAttention(
            head_size=att_hidden_units, head_num=head_num, activation=att_activation, use_res=use_res))
            for i in range(att_layer_num)]
        self.__dict__['attention_layer'] = self.attention_layers[0]     # alias of attention_0 (autoint/model.py:40): not tracked twice
        self.embed_dense = embed_dense and self.nd > 0 and mode == 'intended'
        if self.embed_dense:
            D = self._group.dims[0]
            self.add_weight('dense_embed', (self.nd, D), 'random_uniform')
        self.final_dense = nn.Dense(1, activation=None)
        self._eye = None

    def _head(self, flat):
        """sigmoid(Dense(1)(flat)) in ONE launch (the Dense kernel's fused activation)"""
        fd = self.final_dense
        if not fd.built:
            fd.build(flat.shape[-1])
        return ops.dense(flat, fd._w['kernel'], fd._w.get('bias'), 'sigmoid')

    def _fused_forward(self, dense_inputs, sparse_inputs, F, D):
        """the whole call — lookup, dense-field embedding, interacting layers, Dense(1), sigmoid — in ONE launch
        (rec_autoint_forward_f32) when the configuration is covered; None otherwise.  `self.fused = False` keeps the
        separate launches."""
        if not self.fused or sparse_inputs.dtype != torch.int32:
            return None
        layers = self.attention_layers
        hs = layers[0]._head_num * layers[0]._head_size
        same = all(L._head_num == layers[0]._head_num and L._head_size == layers[0]._head_size and
                   L._activation == layers[0]._activation for L in layers)
        if not same or not isinstance(layers[0]._activation, (str, type(None))):
            return None
        for i, L in enumerate(layers):
            if not L.built:
                L.build(D if i == 0 else hs)
        fd = self.final_dense
        if not fd.built:
            fd.build((F + self.nd) * hs)
        return ops.autoint_forward(self._group, sparse_inputs, dense_inputs, self._w['dense_embed'],
                                   [(L._w['Wq'], L._w['Wk'], L._w['Wv'], L._w.get('W0')) for L in layers],
                                   layers[0]._head_num, layers[0]._head_size, layers[0]._activation, fd._w['kernel'],
                                   fd._w.get('bias'))

    def _interact(self, h):
        """the stacked interacting layers: ONE launch when the fused stack kernel covers them (activations stay in
        registers between layers), else layer by layer"""
        layers = self.attention_layers
        hs = layers[0]._head_num * layers[0]._head_size
        for i, L in enumerate(layers):
            if not L.built:
                L.build(h.shape[-1] if i == 0 else hs)
        same = all(L._head_num == layers[0]._head_num and L._head_size == layers[0]._head_size and
                   L._activation == layers[0]._activation for L in layers)
        if same and isinstance(layers[0]._activation, (str, type(None))):
            out = ops.mha_ctr_stack(h.contiguous(), [(L._w['Wq'], L._w['Wk'], L._w['Wv'], L._w.get('W0')) for L in layers],
                                    layers[0]._head_num, layers[0]._head_size, layers[0]._activation)
            if out is not None:
                return out
        for L in layers:
            h = L(h)
        return h

    def call(self, inputs, **kwargs):
        dense_inputs, sparse_inputs = inputs
        dense_inputs = to_device_f32(dense_inputs, self.device)
        sparse_inputs = to_device_ids(sparse_inputs, self.device)
        B, F = sparse_inputs.shape
        if self.mode == 'intended' and self.embed_dense and len(set(self._group.dims)) == 1:
            # fields of one sample = [26 looked-up rows | 13 value-scaled dense embeddings]: the gather and the scaling
            # kernel write the two parts of ONE (B, (F + nd) * D) buffer (tf.concat as column offsets, no copy pass)
            D = self._group.dims[0]
            fused = self._fused_forward(dense_inputs, sparse_inputs, F, D)
            if fused is not None:
                return fused
            buf = torch.empty((B, (F + self.nd) * D), dtype=torch.float32, device=self.device)
            ops.gather_concat(self._group, sparse_inputs, out=buf)                # :46
            ops.scale_embed(dense_inputs, self._w['dense_embed'], buf[:, F * D:])
            h = self._interact(buf.view(B, F + self.nd, D))
            return self._head(h.reshape(B, -1))                                   # :54-55
        sparse_embed = ops.gather_concat(self._group, sparse_inputs)               # :46
        if self.mode == 'as_written':
            x = torch.cat([sparse_embed, dense_inputs], dim=-1)                    # :48 (2-D!)
            S = self.att_hidden_units
            if B % S:
                raise ValueError(f"as_written: the reference's reshape needs batch % att_hidden_units == 0 ({B} % {S})")
            L = self.attention_layer
            if not L.built:
                L.build(x.shape[-1])
            # Dense on the 2-D tensor (modules.py:255-269), then the 3-D attention core on the
            # sample-mixing view (B/S, S, S); identity projections keep q,k,v bit-identical.
            q = ops.dense(x, L._w['Wq'], None, self.att_activation).view(B // S, S, S)
            k = ops.dense(x, L._w['Wk'], None, self.att_activation).view(B // S, S, S)
            v = ops.dense(x, L._w['Wv'], None, self.att_activation).view(B // S, S, S)
            if self._eye is None:
                self._eye = torch.eye(S, dtype=torch.float32, device=self.device)
            att = ops.mha_ctr(q, k, v, self._eye, self._eye, self._eye, None, 1, S, None)
            flat = att.reshape(B // S, S * S)                                      # :52
        else:
            D = self._group.dims[0]
            h = sparse_embed.view(B, F, D)
            if self.embed_dense:
                h = torch.cat([h, dense_inputs[:, :, None] * self._w['dense_embed'][None]], dim=1).contiguous()
            h = self._interact(h)
            flat = h.reshape(B, -1)
        return self._head(flat)                                                    # :54-55
