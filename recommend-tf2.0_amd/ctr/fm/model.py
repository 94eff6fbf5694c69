"""Mirror of src/ctr/fm/model.py: classic FM.  The reference materialises a (B, nd + sum V_f)
one-hot stack and runs three dense matmuls over ~99.99 % zeros (fm/model.py:37-48); here the
equivalent gather form runs in one HIP kernel (rec_fm_onehot_f32), never building the one-hot."""
from recamd import nn, ops
from recamd.nn import Model, to_device_f32, to_device_ids


class FM(Model):
    def __init__(self, feature_columns, k, w_reg=1e-4, v_reg=1e-4):
        super().__init__()
        self.dense_feature_columns, self.sparse_feature_columns = feature_columns
        self.feature_length = sum(feat['feat_num'] for feat in self.sparse_feature_columns) \
            + len(self.dense_feature_columns)
        self.k = k
        self.w_reg, self.v_reg = w_reg, v_reg                                        # fm/model.py:19-20 (l2 on w and V)
        self.w0 = self.add_weight('w0', (1,), 'zeros')                              # fm/model.py:22-24
        self.w = self.add_weight('w', (self.feature_length, 1), 'random_normal')   # :25-28
        self.V = self.add_weight('V', (self.k, self.feature_length), 'random_normal')  # :29-32 (k, L)

    def call(self, inputs, **kwargs):
        dense_inputs, sparse_inputs = inputs
        dense_inputs = to_device_f32(dense_inputs, self.device)
        sparse_inputs = to_device_ids(sparse_inputs, self.device)
        vocab = [feat['feat_num'] for feat in self.sparse_feature_columns]
        return ops.fm_onehot(dense_inputs, sparse_inputs, vocab, self._w['w0'], self._w['w'], self._w['V'])
