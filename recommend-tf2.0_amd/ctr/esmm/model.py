"""Mirror of src/ctr/esmm/model.py (ESMM: CTR and CVR towers sharing the embeddings, pCTCVR = pCTR * pCVR).

`cate_feature_columns`: dict name -> (vocab, dim); `cate_feature_dict = [user_dict, item_dict]`, each
name -> (column index in the corresponding categorical input, ...) (:19-21).  The categorical inputs are float32
(`layers.Input(shape=(5,))`, :93-96), so ids take the Keras Embedding cast.  The reference creates the head layers
(`BatchNormalization`, two `Dense`) inside build_ctr_model / build_cvr_model (:68-71, :90-93), i.e. once per
tower when the graph is built; they are created here at construction for the same effect."""
import torch

from ctr.layers.modules import DNN
from recamd import nn, ops
from recamd.nn import Model, to_device_f32


class _Head(nn.Layer):
    """concat -> Dropout -> BatchNormalization -> Dense(hidden[-1], relu) -> Dense(1, sigmoid) (:66-72)."""

    def __init__(self, units):
        super().__init__()
        self.bn = self.track('bn', nn.BatchNormalization())
        self.dense = self.track('dense', nn.Dense(units, activation='relu'))
        self.out = self.track('out', nn.Dense(1, activation='sigmoid'))

    def call(self, x, **kwargs):
        if not self.dense.built:
            self.dense.build(x.shape[-1])
        Wf, bf = self.bn.fold(self.dense._w['kernel'], self.dense._w.get('bias'))
        return self.out(self.dense.apply(x, Wf, bf))


class ESMM(Model):
    def __init__(self, cate_feature_columns, cate_feature_dict, hidden_units=[128, 64], activation='relu',
                 dropout=0., embed_reg=1e-4):
        super().__init__()
        self.cate_feature_columns = cate_feature_columns
        self.embed_reg = embed_reg
        self.user_cate_feature_dict, self.item_cate_feature_dict = cate_feature_dict
        self.hidden_units = hidden_units
        self.embed_layers = {
            'embed_' + k: self.track('embed_' + k, nn.Embedding(input_dim=v[0], input_length=1, output_dim=v[1],
                                                                 embeddings_initializer='random_uniform'))
            for k, v in self.cate_feature_columns.items()
        }
        self.user_dnn = self.track('user_dnn', DNN(hidden_units, activation, dropout))
        self.item_dnn = self.track('item_dnn', DNN(hidden_units, activation, dropout))
        self.ctr_head = self.track('ctr_head', _Head(hidden_units[-1]))
        self.cvr_head = self.track('cvr_head', _Head(hidden_units[-1]))
        self._user_group = ops.TableGroup([self.embed_layers['embed_' + k].table for k in self.user_cate_feature_dict])
        self._item_group = ops.TableGroup([self.embed_layers['embed_' + k].table for k in self.item_cate_feature_dict])
        self._user_cols = [v[0] for v in self.user_cate_feature_dict.values()]
        self._item_cols = [v[0] for v in self.item_cate_feature_dict.values()]

    def _tower_input(self, numerical, cate, group, cols):
        numerical = to_device_f32(numerical, self.device)
        cate = to_device_f32(cate, self.device)
        ids = cate[:, cols].contiguous()                 # cate_input[:, v[0]] per feature (:44-48)
        nn_ = numerical.shape[1]
        x = torch.empty((numerical.shape[0], nn_ + group.width), dtype=torch.float32, device=self.device)
        x[:, :nn_] = numerical                           # concat([numerical, embeddings]) (:50-51)
        ops.gather_concat(group, ids, out=x[:, nn_:])
        return x

    def _tower(self, head, un, uc, inum, ic):
        user_feature = self.user_dnn(self._tower_input(un, uc, self._user_group, self._user_cols))   # :53
        item_feature = self.item_dnn(self._tower_input(inum, ic, self._item_group, self._item_cols))  # :54
        return head(ops.concat_cols([user_feature, item_feature]))                                  # :56-61

    def call(self, inputs, **kwargs):
        (ctr_un, ctr_uc, ctr_in, ctr_ic, cvr_un, cvr_uc, cvr_in, cvr_ic) = inputs
        ctr_pred = self._tower(self.ctr_head, ctr_un, ctr_uc, ctr_in, ctr_ic)
        cvr_pred = self._tower(self.cvr_head, cvr_un, cvr_uc, cvr_in, cvr_ic)
        ctcvr_pred = ops.scale_rows(ctr_pred, cvr_pred.reshape(-1))                                   # :37
        return [ctr_pred, ctcvr_pred]
