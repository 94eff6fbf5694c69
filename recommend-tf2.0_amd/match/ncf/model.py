"""Mirror of src/match/ncf/model.py (NeuMF: GMF + MLP over user / positive / negative item embeddings).

call([user (B,1), pos (B,1), neg (B,neg_num)]) -> logits (B, 1 + neg_num) = concat([pos_logits, neg_logits]) (:79).
The reference adds its BCE-style loss with `add_loss` (:75-78): available as `model.losses` after a call
(rec_pairwise_rank_loss_f32); the forward returns the logits that loss is computed from.  Note the GMF and MLP
parts share the SAME embeddings (`mlp_*_embed = self.*_embedding(...)`, :57-59), and the negative items have their
own table `neg_item_embedding` (:38-42)."""
import torch

from match.layers.modules import DNN
from recamd import nn, ops
from recamd.nn import Model, to_device_ids


class NCF(Model):
    def __init__(self, user_feature_columns, item_feature_columns, hidden_units=[64, 16, 8], dropout=0.2,
                 activation='relu', neg_num=10, embed_reg=1e-6, **kwargs):
        super().__init__()
        self.neg_num = neg_num
        self.embed_reg = embed_reg
        emb = lambda fc: nn.Embedding(input_dim=fc['feat_num'], output_dim=fc['embed_dim'],  # noqa: E731
                                      embeddings_initializer='random_normal')
        self.user_embedding = self.track('user_embedding', emb(user_feature_columns))
        self.item_embedding = self.track('item_embedding', emb(item_feature_columns))
        self.neg_item_embedding = self.track('neg_item_embedding', emb(item_feature_columns))
        self.dnn = self.track('dnn', DNN(hidden_units, activation=activation, dnn_dropout=dropout))
        self.dense = nn.Dense(1, activation=None)

    def _lookup(self, layer, ids):
        """(B, T) ids -> (B*T, dim) rows of one table."""
        ids = to_device_ids(ids, self.device)
        B, T = ids.shape
        return ops.gather_concat(ops.TableGroup([layer.table]), ids.reshape(B * T, 1).contiguous()), B, T

    def _branch(self, user, item, B, T):
        """user (B, dim), item (B*T, dim) -> logits (B, T): Dense([sigmoid(u*i), DNN([u, i])]) (:53-73)."""
        dim = user.shape[1]
        u = user[:, None, :].expand(B, T, dim).reshape(B * T, dim).contiguous()        # tf.tile (:63)
        gmf = ops.mul_act(u, item, 'sigmoid')                                            # :53-54
        mlp = self.dnn(ops.concat_cols([u, item]))                                     # :61-66
        return self.dense(ops.concat_cols([gmf, mlp])).reshape(B, T)                   # :69-73

    def call(self, inputs, training=None, mask=None):
        user_inputs, pos_inputs, neg_inputs = inputs
        user, B, _ = self._lookup(self.user_embedding, user_inputs)
        pos, _, Tp = self._lookup(self.item_embedding, pos_inputs)
        neg, _, Tn = self._lookup(self.neg_item_embedding, neg_inputs)
        pos_logits = self._branch(user, pos, B, Tp)
        neg_logits = self._branch(user, neg, B, Tn)
        self._logits = ops.concat_cols([pos_logits, neg_logits])                      # :79
        return self._logits

    @property
    def losses(self):
        """[mean(-log sigmoid(pos) - log(1 - sigmoid(neg))) / 2] of the last call (the add_loss of :75-77), computed on
        demand by rec_pairwise_rank_loss_f32 (pos_num = 1, as in the reference's data)."""
        lg = getattr(self, '_logits', None)
        return [] if lg is None else [ops.pairwise_rank_loss(lg)[0]]
