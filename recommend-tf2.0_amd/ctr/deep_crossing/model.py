"""Mirror of src/ctr/deep_crossing/model.py (Deep & Crossing) on the HIP kernels: embedding stack ->
residual units -> Dense(1) -> sigmoid.  `feature_columns` is the sparse list only (:26)."""
from ctr.layers.modules import Residual_Units
from recamd import nn, ops
from recamd.nn import Model, to_device_ids


class Deep_Crossing(Model):
    def __init__(self, feature_columns, hidden_units, res_dropout=0., embed_reg=1e-6):
        super().__init__()
        self.sparse_feature_columns = feature_columns
        self.embed_reg = embed_reg
        self.embed_layers = {
            'embed_' + str(i): self.track('embed_' + str(i), nn.Embedding(
                input_dim=feat['feat_num'], input_length=1, output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for i, feat in enumerate(self.sparse_feature_columns)
        }
        embed_layers_len = sum(feat['embed_dim'] for feat in self.sparse_feature_columns)
        self.res_network = [self.track(f'res_{i}', Residual_Units(unit, embed_layers_len))
                            for i, unit in enumerate(hidden_units)]
        self.res_dropout = nn.Dropout(res_dropout)
        self.dense = nn.Dense(1, activation=None)
        self._group = ops.TableGroup([self.embed_layers['embed_%d' % i].table
                                      for i in range(len(self.sparse_feature_columns))])

    def call(self, inputs, **kwargs):
        sparse_inputs = to_device_ids(inputs, self.device)
        r = ops.gather_concat(self._group, sparse_inputs)                  # deep_crossing/model.py:44-45
        for res in self.res_network:                                       # :47-48
            r = res(r)
        r = self.res_dropout(r)
        return ops.add_sigmoid(self.dense(r))                              # :50
