"""Mirror of src/match/dssm/model.py on the HIP kernels.

call([user_sparse_inputs {feat: (B,1)}, item_sparse_inputs {feat: (B,1)}]) -> sigmoid(cosine) reshaped (-1, 1).
As written (:49-62) `cosine_similarity` flattens BOTH tower outputs of the whole batch into one vector each, so the
model's output is ONE value of shape (1, 1) whatever the batch size; that is reproduced.  The tensors a train script
actually uses afterwards are the towers, exported as `user_embed` / `item_embed` (:99-100) and fed to
faiss.IndexFlatIP (src/match/dssm/dssm_train.py:63-78) -> `self.user_dnn_out` / `self.item_dnn_out` here and
recamd.retrieval.IndexFlatIP."""
import torch

from match.layers.modules import DNN
from recamd import nn, ops
from recamd.nn import Model, to_device_f32


class Dssm(Model):
    def __init__(self, user_sparse_feature_columns, item_sparse_feature_columns, user_dense_feature_columns=(),
                 item_dense_feature_columns=(), num_sampled=1,
                 user_dnn_hidden_units=(64, 32), item_dnn_hidden_units=(64, 32), dnn_activation='relu',
                 l2_reg_embedding=1e-6, dnn_dropout=0, **kwargs):
        super().__init__()
        self.num_sampled = num_sampled
        self.embed_reg = l2_reg_embedding                              # :33,42 (recamd.train.default_l2)
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
        self.user_dnn = self.track('user_dnn', DNN(user_dnn_hidden_units, dnn_activation, dnn_dropout))
        self.item_dnn = self.track('item_dnn', DNN(item_dnn_hidden_units, dnn_activation, dnn_dropout))
        self.user_dnn_out = None
        self.item_dnn_out = None

    def _tower(self, inputs, layers, dnn):
        keys = list(inputs.keys())                                   # dict order, like `.items()` (:68)
        ids = torch.cat([to_device_f32(inputs[k], self.device).reshape(-1, 1) for k in keys], dim=1)
        g = ops.TableGroup([layers['embed_{}'.format(k)].table for k in keys])
        emb = ops.gather_concat(g, ids.contiguous())                  # float ids truncate (Keras cast, :87-90)
        return dnn(emb)[:, None, :]                                   # (B, 1, units)

    def user_tower(self, user_sparse_inputs):
        return self._tower(user_sparse_inputs, self.user_embed_layers, self.user_dnn)

    def item_tower(self, item_sparse_inputs):
        return self._tower(item_sparse_inputs, self.item_embed_layers, self.item_dnn)

    def cosine_similarity(self, tensor1, tensor2):
        return ops.cosine_flat(tensor1, tensor2)                       # :49-62, one scalar

    def call(self, inputs, training=None, mask=None):
        user_sparse_inputs, item_sparse_inputs = inputs
        self.user_dnn_out = self.user_tower(user_sparse_inputs)        # :68-72
        self.item_dnn_out = self.item_tower(item_sparse_inputs)        # :74-77
        return ops.cosine_flat(self.item_dnn_out, self.user_dnn_out, sigmoid=True).reshape(-1, 1)   # :79-80
