"""Mirror of src/match/fm/model.py (FM over the concatenated user / item embeddings) on the HIP kernels.

call([user_sparse_inputs {feat: (B,1)}, item_sparse_inputs {feat: (B,1)}]) -> sigmoid(first + second) (B, 1), with
`stack` = all embeddings of a sample side by side (:73-75).  `user_embeds` / `item_embeds` (:66,70; the reduce_sum
over the length-1 axis is a squeeze) are what src/match/fm/train.py:60-75 feeds to faiss.IndexFlatIP.

second = 0.5 * sum_k[(stack V^T)_k^2 - (stack^2 (V^T)^2)_k] (:77-79): the subtracted term only needs the row sums
of V^2, sum_k (stack^2 (V^T)^2)_k = stack^2 . (sum_k V_k^2), so it is one (L,1) mat-vec on a transformed weight."""
import torch

from recamd import nn, ops
from recamd.nn import Model, to_device_f32


class FM(Model):
    def __init__(self, user_sparse_feature_columns, item_sparse_feature_columns, k, w_reg=1e-4, v_reg=1e-4,
                 l2_reg_embedding=1e-6):
        super().__init__()
        self.user_sparse_feature_columns = user_sparse_feature_columns
        self.item_sparse_feature_columns = item_sparse_feature_columns
        self.feature_length = sum(f['embed_dim'] for f in user_sparse_feature_columns) \
            + sum(f['embed_dim'] for f in item_sparse_feature_columns)
        self.k = k
        self.w_reg, self.v_reg, self.embed_reg = w_reg, v_reg, l2_reg_embedding    # :43,47,56,65 (recamd.train.default_l2)
        self.w0 = self.add_weight('w0', (1,), 'zeros')
        self.w = self.add_weight('w', (self.feature_length, 1), 'random_normal')
        self.V = self.add_weight('V', (self.k, self.feature_length), 'random_normal')
        mk = lambda feat: nn.Embedding(input_dim=feat['feat_num'], input_length=feat['feat_len'],  # noqa: E731
                                       output_dim=feat['embed_dim'], embeddings_initializer='random_uniform')
        self.user_embed_layers = {'embed_' + str(f['feat']): self.track('user_embed_' + str(f['feat']), mk(f))
                                  for f in user_sparse_feature_columns}
        self.item_embed_layers = {'embed_' + str(f['feat']): self.track('item_embed_' + str(f['feat']), mk(f))
                                  for f in item_sparse_feature_columns}
        self.user_embeds = None
        self.item_embeds = None
        self._derived = None

    def _embed(self, inputs, layers, out):
        keys = list(inputs.keys())
        ids = torch.cat([to_device_f32(inputs[k], self.device).reshape(-1, 1) for k in keys], dim=1)
        g = ops.TableGroup([layers['embed_{}'.format(k)].table for k in keys])
        return ops.gather_concat(g, ids.contiguous(), out=out)

    def _weights(self):
        if self._derived is None or self._derived[0] != self._version:
            V = self._w['V']
            self._derived = (self._version, V.t().contiguous(), (V * V).sum(dim=0).reshape(-1, 1).contiguous(),
                             torch.ones((self.k, 1), dtype=torch.float32, device=self.device))
        return self._derived[1:]

    def call(self, inputs, **kwargs):
        user_sparse_inputs, item_sparse_inputs = inputs
        B = next(iter(user_sparse_inputs.values())).shape[0]
        Lu = sum(f['embed_dim'] for f in self.user_sparse_feature_columns)
        stack = torch.empty((B, self.feature_length), dtype=torch.float32, device=self.device)
        self._embed(user_sparse_inputs, self.user_embed_layers, stack[:, :Lu])            # :64-65
        if Lu % 4 == 0:
            self._embed(item_sparse_inputs, self.item_embed_layers, stack[:, Lu:])        # :68-69, in place (:73)
        else:
            stack[:, Lu:] = self._embed(item_sparse_inputs, self.item_embed_layers, None)
        self.user_embeds, self.item_embeds = stack[:, :Lu], stack[:, Lu:]                 # :66, :70
        Vt, vsq, ones = self._weights()
        first = ops.dense(stack, self._w['w'], self._w['w0'])                            # :76
        a = ops.dense(stack, Vt)                                                         # (B, k)
        s1 = ops.dense(ops.mul_act(a, a), ones)                                          # sum_k (stack V^T)_k^2
        t = ops.dense(ops.mul_act(stack, stack), vsq)                                    # sum_k (stack^2 (V^T)^2)_k
        second = ops.axpby_act(s1, t, 0.5, -0.5)                                         # :77-79
        return ops.add_sigmoid(first, second)                                            # :80-82
