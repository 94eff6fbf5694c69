"""Mirror of src/ctr/layers/modules.py: Residual_Units, FM, CrossNetwork, DNN, AttentionLayer, MultiHeadAttention,
Dice — same constructor signatures and `call(inputs)` structure, arithmetic on the HIP kernels."""
import torch

from recamd import nn, ops
from recamd.nn import Dice  # noqa: F401  (src/ctr/layers/modules.py:327-337)


class Residual_Units(nn.Layer):
    """src/ctr/layers/modules.py:15-34: Dense(hidden_unit, relu) -> Dense(dim_stack) -> relu(x + inputs)."""

    def __init__(self, hidden_unit, dim_stack):
        super().__init__()
        self.layer1 = self.track('layer1', nn.Dense(units=hidden_unit, activation='relu'))
        self.layer2 = self.track('layer2', nn.Dense(units=dim_stack, activation=None))

    def call(self, inputs, **kwargs):
        x = self.layer2(self.layer1(inputs))
        return ops.axpby_act(x, inputs, 1.0, 1.0, 'relu')        # :33


class FM(nn.Layer):
    """Wide part of DeepFM (src/ctr/layers/modules.py:36-72).  call([first (B,L1), second (B,M)]):
    first-order = ONE scalar over the whole batch (:65); second-order on the 2-D `second` (:67-69)."""

    def __init__(self, feature_length, w_reg=1e-6):
        super().__init__()
        self.feature_length = feature_length
        self.w_reg = w_reg
        self.w = self.add_weight('w', (feature_length, 1), 'random_normal')

    def call(self, inputs, **kwargs):
        first_inputs, second_inputs = inputs
        return ops.fm_layer(first_inputs, second_inputs, self._w['w'])


class CrossNetwork(nn.Layer):
    """src/ctr/layers/modules.py:74-112: x_{l+1} = x0 (x_l . w_l) + b_l + x_l."""

    def __init__(self, layer_num, reg_w=1e-6, reg_b=1e-6):
        super().__init__()
        self.layer_num = layer_num
        self.reg_w, self.reg_b = reg_w, reg_b      # l2 coefficients of w_i / b_i (:92, :100), training only
        self.built = False

    def build(self, dim):
        # w_i, b_i are (dim, 1) random_normal each (bias is NOT zero-init, :88-101); stored (L, dim)
        self.add_weight('cross_weights', (self.layer_num, dim), 'random_normal')
        self.add_weight('cross_bias', (self.layer_num, dim), 'random_normal')
        self.built = True

    def call(self, inputs, **kwargs):
        if not self.built:
            self.build(inputs.shape[-1])
        return ops.cross_network(inputs, self._w['cross_weights'], self._w['cross_bias'])


class DNN(nn.Layer):
    """Deep part (src/ctr/layers/modules.py:114-135): BatchNormalization()(x) -> Dense stack ->
    Dropout.  The BN is folded into the first Dense (one pass over the activations)."""

    def __init__(self, hidden_units, activation='relu', dnn_dropout=0.):
        super().__init__()
        self.dnn_network = [nn.Dense(units=unit, activation=activation) for unit in hidden_units]
        for i, d in enumerate(self.dnn_network):
            self.track(f'dense_{i}', d)
        self.bn = nn.BatchNormalization()
        self.dropout = nn.Dropout(dnn_dropout)
        self._folded = None

    def call(self, inputs, out=None, lead_pad=0, tail_pad=0, row_absmax=None, **kwargs):
        """lead_pad / tail_pad > 0: `inputs` carries that many extra ZERO columns in front of / behind the features (a
        16-B aligned view of a concat buffer; the zero pad column of a row stride rounded up to 4 floats); the folded
        first-layer kernel gets as many zero rows, so the product is unchanged and the GEMM stays on its aligned path
        (K a multiple of 32 takes the pipelined kernel)."""
        layer = self.dnn_network[0]
        if not layer.built:
            layer.build(inputs.shape[-1] - lead_pad - tail_pad)
        key = (self._version, self.bn._version, layer._version, lead_pad, tail_pad)
        if self._folded is None or self._folded[0] != key:
            Wf, bf = self.bn.fold(layer._w['kernel'], layer._w.get('bias'))
            if lead_pad or tail_pad:
                z = lambda r: torch.zeros((r, Wf.shape[1]), dtype=Wf.dtype, device=Wf.device)  # noqa: E731
                Wf = torch.cat([z(lead_pad), Wf, z(tail_pad)], dim=0).contiguous()
            self._folded = (key, (Wf, bf))
        x = nn.dense_chain(self.dnn_network, inputs, out=out, first=self._folded[1], row_absmax=row_absmax)
        return self.dropout(x)


class AttentionLayer(nn.Layer):
    """DIN history attention (src/ctr/layers/modules.py:137-175).  call([q (B,d), k (B,T,d),
    v (B,T,d), mask (B,T) | None]).  Only hidden_unit == 1 is coherent with the reference's
    reshape(-1, T) (:159).  `activation='prelu'` (the reference default) is not a valid Keras
    activation string; pass 'sigmoid' / 'relu' / None, or 'prelu' to get a learnable scalar slope."""

    def __init__(self, hidden_unit, activation='prelu'):
        super().__init__()
        if hidden_unit != 1:
            raise ValueError("AttentionLayer: the reference's reshape(-1, seq_len) only works for hidden_unit=1")
        self.activation = activation
        self.built = False

    def build(self, d):
        self.add_weight('kernel', (4 * d, 1), 'glorot_uniform')
        self.add_weight('bias', (1,), 'zeros')
        if self.activation == 'prelu':
            self.add_weight('alpha', (1,), 'zeros')
        self.built = True

    def call(self, inputs, **kwargs):
        q, k, v, mask = inputs
        if not self.built:
            self.build(k.shape[-1])
        if mask is not None and not isinstance(mask, torch.Tensor):
            mask = None  # non-tensor mask => all scores replaced by the padding (:164-165)
        if mask is not None and mask.dtype != torch.float32:
            mask = mask.to(torch.float32)
        return ops.din_attention_pool(q, k, v, mask, self._w['kernel'], self._w['bias'], self.activation,
                                      self._w.get('alpha'))


class MultiHeadAttention(nn.Layer):
    """AutoInt interacting layer (src/ctr/layers/modules.py:177-325).  The reference creates its
    Dense layers inside call() (:255-269, :317); here they are explicit weights built on first use."""

    def __init__(self, head_size, head_num=1, l2_reg=1e-4, activation='relu', use_res=False, name=''):
        super().__init__(name or None)
        self._l2_reg = l2_reg          # coefficient of the reference's kernel_regularizer=l2(1e-4) (:179), training only
        self._head_num = head_num
        self._head_size = head_size
        self._activation = activation
        self._use_res = use_res
        self.built = False

    def build(self, din):
        hs = self._head_num * self._head_size
        for n in ('Wq', 'Wk', 'Wv'):
            self.add_weight(n, (din, hs), 'glorot_uniform')
        if self._use_res:
            self.add_weight('W0', (din, hs), 'glorot_uniform')
        self.built = True

    def call(self, inputs, **kwargs):
        if isinstance(inputs, list):
            assert len(inputs) == 3 or len(inputs) == 1, \
                'If the input of multi_head_attention is a list, the length must be 1 or 3.'
            ori_q, ori_k, ori_v = (inputs if len(inputs) == 3 else (inputs[0],) * 3)
        else:
            ori_q = ori_k = ori_v = inputs
        if ori_q.dim() != 3:
            raise ValueError("MultiHeadAttention: expects (B, feature_num, d_model); the reference's 2-D call "
                             "(src/ctr/autoint/model.py:48-51) mixes samples — see AutoInt(mode='as_written')")
        if not self.built:
            self.build(ori_q.shape[-1])
        w = self._w
        return ops.mha_ctr(ori_q, ori_k, ori_v, w['Wq'], w['Wk'], w['Wv'], w.get('W0'), self._head_num,
                           self._head_size, self._activation)
