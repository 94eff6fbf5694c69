"""Mirror of src/match/layers/modules.py (hot-path part): DNN, split_heads,
scaled_dot_product_attention, MultiHeadAttention, FFN, TransformerEncoder on the HIP kernels.
PoolingLayer / CapsuleLayer / LabelAwareAttention / SELayer (MIND, SENet) and SampledSoftmaxLayer
(stochastic, see match/youtube_dnn/model.py) are out of the hot path's scope."""
import torch

from recamd import nn, ops


class DNN(nn.Layer):
    """src/match/layers/modules.py:8-26: Dense stack, no BatchNormalization."""

    def __init__(self, hidden_units, activation='relu', dnn_dropout=0., **kwargs):
        super().__init__(kwargs.get('name'))
        self.dnn_network = [self.track('dense_%d' % i, nn.Dense(units=unit, activation=activation))
                            for i, unit in enumerate(hidden_units)]
        self.dropout = nn.Dropout(dnn_dropout)

    def call(self, inputs, **kwargs):
        if inputs.dim() == 2:
            return self.dropout(nn.dense_chain(self.dnn_network, inputs))    # row maxima handed from layer to layer
        x = inputs
        for dnn in self.dnn_network:
            x = dnn(x)
        return self.dropout(x)


def split_heads(x, seq_len, num_heads, depth):
    """(B, S, H*depth) -> (B, H, S, depth) (src/match/layers/modules.py:63-74)."""
    return x.reshape(-1, seq_len, num_heads, depth).permute(0, 2, 1, 3)


def scaled_dot_product_attention(q, k, v, mask):
    """src/match/layers/modules.py:76-96 on (B, H, S, dk) tensors, mask (B, H, S, 1)."""
    B, H, Sq, dk = q.shape
    merge = lambda t: t.permute(0, 2, 1, 3).reshape(B, t.shape[2], H * dk).contiguous()  # noqa: E731
    m = mask[:, 0, :, 0].to(torch.float32).contiguous()
    out = ops.mha_rowmask(merge(q), merge(k), merge(v), m, H)
    return out.reshape(B, Sq, H, dk).permute(0, 2, 1, 3)


class MultiHeadAttention(nn.Layer):
    """src/match/layers/modules.py:98-131: wq/wk/wv Dense WITH bias, no activation, no output
    projection; the mask (B,S,1) masks whole QUERY rows; not causal."""

    def __init__(self, d_model, num_heads):
        super().__init__()
        self.d_model = d_model
        self.num_heads = num_heads
        self.wq = nn.Dense(d_model, activation=None)
        self.wk = nn.Dense(d_model, activation=None)
        self.wv = nn.Dense(d_model, activation=None)

    def project(self, xq, xk, xv):
        """wq(xq), wk(xk), wv(xv) (:115-117).  When inputs coincide (self-attention) the Dense layers that share an
        input run as ONE GEMM over a concatenated kernel [Wq | Wk | Wv] (a cached parameter transform) and the
        results are column slices of its output: the input is read once instead of two or three times."""
        for layer, x in ((self.wq, xq), (self.wk, xk), (self.wv, xv)):
            if not layer.built:
                layer.build(x.shape[-1])
        if xk is xv:
            group = (self.wq, self.wk, self.wv) if xq is xk else (self.wk, self.wv)
            key = tuple(l._version for l in group) + (len(group),)
            if getattr(self, '_fused', None) is None or self._fused[0] != key:
                W = torch.cat([l._w['kernel'] for l in group], dim=1).contiguous()
                b = torch.cat([l._w['bias'] for l in group], dim=0).contiguous()
                self._fused = (key, W, b)
            y = ops.dense(xk, self._fused[1], self._fused[2])
            d = self.d_model
            if len(group) == 3:
                return y[..., :d], y[..., d:2 * d], y[..., 2 * d:]
            return self.wq(xq), y[..., :d], y[..., d:]
        return self.wq(xq), self.wk(xk), self.wv(xv)

    def call(self, q, k=None, v=None, mask=None, **kwargs):
        k = q if k is None else k
        v = q if v is None else v
        q, k, v = self.project(q, k, v)
        m = mask.reshape(mask.shape[0], -1).to(torch.float32).contiguous()
        return ops.mha_rowmask(q, k, v, m, self.num_heads)


class FFN(nn.Layer):
    """src/match/layers/modules.py:134-149: Conv1D(k=1, relu) -> Conv1D(k=1) == two Dense."""

    def __init__(self, hidden_unit, d_model):
        super().__init__()
        self.conv1 = nn.Dense(hidden_unit, activation='relu', use_bias=True)
        self.conv2 = nn.Dense(d_model, activation=None, use_bias=True)

    def call(self, inputs, **kwargs):
        return self.conv2(self.conv1(inputs))


class TransformerEncoder(nn.Layer):
    """src/match/layers/modules.py:152-185.  call([x (B,S,d), mask (B,S,1)]).
    Extra keyword `query_rows`: when given (B, Sq, d) only those query rows are encoded against the
    full keys/values — SASRec consumes only the last position of its last block
    (src/match/sasrec/model.py:88), an exact saving; `out_mask` (B,Sq) is fused into the last LN."""

    def __init__(self, d_model, num_heads=1, ffn_hidden_unit=128, dropout=0., layer_norm_eps=1e-6):
        super().__init__()
        self.mha = MultiHeadAttention(d_model, num_heads)
        self.ffn = FFN(ffn_hidden_unit, d_model)
        self.layernorm1 = nn.LayerNormalization(epsilon=layer_norm_eps)
        self.layernorm2 = nn.LayerNormalization(epsilon=layer_norm_eps)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def call(self, inputs, query_rows=None, query_mask=None, out_mask=None, gather=None, **kwargs):
        """gather=(table, ids): the keys/values are table[ids] (ids (B,S), out-of-range -> zero rows) and `x` may be
        None — only valid together with query_rows on the few-rows/one-head path, where the sequence tensor is then
        never materialised."""
        x, mask = inputs
        if query_rows is None:
            xq, mq = x, mask
        else:
            xq, mq = query_rows, query_mask
        mha = self.mha
        m = mq.reshape(mq.shape[0], -1).to(torch.float32).contiguous()
        if query_rows is not None and mha.num_heads == 1 and xq.shape[1] <= 8:
            # Few query rows, one head: the K and V projections of all S positions are not needed.
            #   q . k_j = q . (x_j Wk + bk) = x_j . (Wk q) + const   (a per-query constant does not change the softmax)
            #   sum_j p_j (x_j Wv + bv)     = (sum_j p_j x_j) Wv + bv  (the p_j sum to 1)
            # so the attention runs over the RAW sequence rows with the query pulled back through Wk, and only the
            # pooled row is projected by Wv: one pass over x instead of a (B*S, d) x (d, 2d) GEMM plus two passes.
            for layer in (mha.wq, mha.wk, mha.wv):
                if not layer.built:
                    layer.build(xq.shape[-1])
            key = mha.wk._version
            if getattr(self, '_wk_t', None) is None or self._wk_t[0] != key:
                self._wk_t = (key, mha.wk._w['kernel'].t().contiguous())
            q_back = ops.dense(mha.wq(xq), self._wk_t[1])                     # (B, Sq, d_in) = Wk q
            if gather is not None:
                pooled = ops.gather_mha_fewq(q_back, gather[0], gather[1], m, 1)
            else:
                pooled = ops.mha_rowmask(q_back, x, x, m, 1)                  # softmax(q_back . x_j / sqrt(d)) x_j
            att_out = mha.wv(pooled)
        else:
            q, k, v = mha.project(xq, x, x)
            att_out = ops.mha_rowmask(q, k, v, m, mha.num_heads)
        out1 = self.layernorm1(xq, residual=att_out)                         # LN(x + att)
        ffn_out = self.ffn(out1)
        return self.layernorm2(out1, residual=ffn_out, row_mask=out_mask)    # LN(out1 + ffn) [* mask]
