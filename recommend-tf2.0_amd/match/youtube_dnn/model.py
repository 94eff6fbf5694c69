"""Mirror of src/match/youtube_dnn/model.py (towers) on the HIP kernels.

call([user_sparse_inputs {feat: (B,1)}, item_sparse_inputs {feat: (B,1)}, labels]) computes the two
towers (youtube_dnn/model.py:47-56) and returns (item_dnn_out, user_dnn_out), each (B, 1, units) —
the tensors the reference exports as `item_embeding` / `user_embeding` (:77-78).
The reference then feeds them to `SampledSoftmaxLayer` (:59), which calls
tf.nn.sampled_softmax_loss with an UNSEEDED log-uniform sampler, the tower width as num_classes and
in-batch item vectors as class weights (src/match/layers/modules.py:35,43-61): non-deterministic
and semantically incoherent, so it is excluded from parity and not reproduced."""
import torch

from match.layers.modules import DNN
from recamd import nn, ops
from recamd.nn import Model, to_device_f32


class YoutubeDNN(Model):
    def __init__(self, user_sparse_feature_columns, item_sparse_feature_columns, user_dense_feature_columns=(),
                 item_dense_feature_columns=(), num_sampled=1,
                 user_dnn_hidden_units=(64, 32), item_dnn_hidden_units=(64, 32), dnn_activation='relu',
                 l2_reg_embedding=1e-6, dnn_dropout=0, **kwargs):
        super().__init__()
        self.num_sampled = num_sampled
        self.user_sparse_feature_columns = user_sparse_feature_columns
        self.user_dense_feature_columns = user_dense_feature_columns
        self.item_sparse_feature_columns = item_sparse_feature_columns
        self.item_dense_feature_columns = item_dense_feature_columns
        self.user_embed_layers = {
            'embed_' + str(feat['feat']): self.track('user_embed_' + str(feat['feat']), nn.Embedding(
                input_dim=feat['feat_num'], input_length=feat['feat_len'], output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for feat in self.user_sparse_feature_columns
        }
        self.item_embed_layers = {
            'embed_' + str(feat['feat']): self.track('item_embed_' + str(feat['feat']), nn.Embedding(
                input_dim=feat['feat_num'], input_length=feat['feat_len'], output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for feat in self.item_sparse_feature_columns
        }
        self.user_dnn = DNN(user_dnn_hidden_units, dnn_activation, dnn_dropout)
        self.item_dnn = DNN(item_dnn_hidden_units, dnn_activation, dnn_dropout)
        self.user_dnn_out = None
        self.item_dnn_out = None

    def _tower(self, inputs, layers, dnn):
        keys = list(inputs.keys())                                   # dict order, like `.items()` (:47)
        ids = torch.cat([to_device_f32(inputs[k], self.device).reshape(-1, 1) for k in keys], dim=1)
        g = ops.TableGroup([layers['embed_{}'.format(k)].table for k in keys])
        emb = ops.gather_concat(g, ids.contiguous())                  # float ids truncate (Keras cast)
        return dnn(emb)[:, None, :]                                   # (B, 1, units)

    def call(self, inputs, training=None, mask=None):
        user_sparse_inputs, item_sparse_inputs, labels = inputs
        self.user_dnn_out = self._tower(user_sparse_inputs, self.user_embed_layers, self.user_dnn)   # :47-51
        self.item_dnn_out = self._tower(item_sparse_inputs, self.item_embed_layers, self.item_dnn)   # :53-56
        return self.item_dnn_out, self.user_dnn_out
