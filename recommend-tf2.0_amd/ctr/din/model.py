"""Mirror of src/ctr/din/model.py on the HIP kernels.

mode='as_written': the reference's forward — self-attention (`MultiHeadAttention`, head_num=1) over
  the behaviour embeddings (din/model.py:77), `reshape(-1, head_size)` (:79) and a concat with the
  (B, .) user/item features (:81).  For maxlen > 1 that concat has mismatched batch sizes and
  TensorFlow raises; so does this class (ValueError).  maxlen == 1 runs.
mode='intended' (default): canonical DIN — the `AttentionLayer` that sits unused in
  src/ctr/layers/modules.py:137-175 pools the history against the candidate item
  (q = item embeddings, k = v = behaviour embeddings, mask = behaviour id != 0); BASELINE config 4.
Quirks kept in both modes: `item_embed` concatenates the raw item ids AS FLOATS with their
embeddings and `item_dense_input` is unused (:68); behaviour column 'item_sparse_{ml}_{i}' reads
table 'item_sparse_{i}' (:71); all five inputs are float32 (:96-100) and ids are truncated by the
Embedding cast."""
import torch

from ctr.layers.modules import AttentionLayer, MultiHeadAttention
from recamd import nn, ops
from recamd.nn import Model, to_device_f32


class DIN(Model):
    def __init__(self, sparse_feature_dict, sparse_feature_index, att_hidden_units=64,
                 ffn_hidden_units=(80, 40), att_activation='prelu', ffn_activation='prelu', maxlen=10,
                 dnn_dropout=0., att_l2_reg=1e-4, embed_reg=1e-4, mode='intended', fuse_history=True):
        super().__init__()
        if mode not in ('intended', 'as_written'):
            raise ValueError("mode must be 'intended' or 'as_written'")
        self.mode = mode
        self.embed_reg = embed_reg
        self.fuse_history = fuse_history
        self.maxlen = maxlen
        self.sparse_feature_dict = sparse_feature_dict
        self.user_sparse_feature_index, self.item_sparse_feature_index, self.behavior_feature_index = \
            sparse_feature_index
        self.embed_layers = {
            'embed_' + k: self.track('embed_' + k, nn.Embedding(
                input_dim=v[0], input_length=1, output_dim=v[1], embeddings_initializer='random_uniform'))
            for k, v in self.sparse_feature_dict.items()
        }
        if mode == 'as_written':
            act = att_activation if att_activation != 'prelu' else 'relu'  # 'prelu' is not a Keras string
            self.attention_layer = MultiHeadAttention(head_size=att_hidden_units, activation=act)
        else:
            self.attention_layer = AttentionLayer(1, activation=att_activation)
        self.att_hidden_units = att_hidden_units
        self.bn = nn.BatchNormalization(trainable=True)
        self.ffn = [self.track('ffn_%d' % i, nn.Dense(unit, activation=nn.PReLU() if ffn_activation == 'prelu'
                                                      else nn.Dice()))
                    for i, unit in enumerate(ffn_hidden_units)]
        self.dropout = nn.Dropout(dnn_dropout)
        self.final_output = nn.Dense(1)
        tab = lambda k: self.embed_layers['embed_' + k].table  # noqa: E731
        self._user_group = ops.TableGroup([tab(k) for k in self.user_sparse_feature_index])
        self._user_cols = list(self.user_sparse_feature_index.values())
        self._item_group = ops.TableGroup([tab(k) for k in self.item_sparse_feature_index])
        self._item_cols = list(self.item_sparse_feature_index.values())
        # behaviour key 'item_sparse_{ml}_{i}' -> table 'item_sparse_{i}'   (din/model.py:71)
        beh_tables = []
        for k in self.behavior_feature_index:
            p = k.split('_')
            beh_tables.append(tab(f"{p[0]}_{p[1]}_{p[3]}"))
        self._beh_cols = list(self.behavior_feature_index.values())
        self._beh_tables = beh_tables
        n_item = len(self.item_sparse_feature_index)
        # fast path: the behaviour columns are maxlen repeats of the item tables in order -> view the
        # (B, maxlen*n_item) ids as (B*maxlen, n_item) and gather with n_item descriptors
        self._beh_regular = (len(beh_tables) == maxlen * n_item and
                             self._beh_cols == list(range(maxlen * n_item)) and
                             all(beh_tables[j].data_ptr() == self._item_group.tables[j % n_item].data_ptr()
                                 for j in range(len(beh_tables))))
        self._beh_group = None if self._beh_regular else ops.TableGroup(beh_tables)
        self._folded = None

    def _cols(self, x, cols):
        if cols == list(range(x.shape[1])):
            return x
        return x[:, cols].contiguous()

    def call(self, inputs, **kwargs):
        user_dense_input, user_sparse_input, item_dense_input, item_sparse_input, behavior_input = \
            [to_device_f32(t, self.device) for t in inputs]
        B = user_sparse_input.shape[0]
        # all_inputs = concat[user_dense, user_emb | item_sparse (ids as floats), item_emb | att] (:62-68, :81) lives in
        # ONE buffer: every part is written at its column offset (gathers with out=, raw inputs with copy_cols)
        nud, wu = user_dense_input.shape[1], self._user_group.width
        nis, wi = item_sparse_input.shape[1], self._item_group.width
        att_w = wi if self.mode == 'intended' else self.att_hidden_units
        o_uemb, o_isp, o_iemb, o_att = nud, nud + wu, nud + wu + nis, nud + wu + nis + wi
        total = o_att + att_w
        all_buf = torch.empty((B, (total + 3) // 4 * 4), dtype=torch.float32, device=self.device)[:, :total]
        ops.copy_cols(user_dense_input, all_buf)
        user_embeddings = ops.gather_concat(self._user_group, self._cols(user_sparse_input, self._user_cols),
                                            out=all_buf[:, o_uemb:o_uemb + wu])           # :62-64
        ops.copy_cols(item_sparse_input, all_buf[:, o_isp:])                            # :68 (ids as floats)
        # the item embeddings are also the attention query, which the pooling kernels want contiguous: gathered once
        # into their own (B, wi) buffer and copied into the concat (6 MB at config 4)
        item_embeddings = ops.gather_concat(self._item_group, self._cols(item_sparse_input, self._item_cols))  # :66-68
        ops.copy_cols(item_embeddings, all_buf[:, o_iemb:])
        self._all_buf, self._o_att = all_buf, o_att
        user_embed = item_embed = None
        d_item = item_embeddings.shape[1]
        beh_ids = self._cols(behavior_input, self._beh_cols)
        if self.mode == 'intended' and self._beh_regular and self.fuse_history:
            # fused history gather + pooling: the (B, maxlen, d) behaviour tensor is never materialised
            n_item = len(self._item_group)
            L = self.attention_layer
            if not L.built:
                L.build(d_item)
            att_outputs = ops.gather_din_attention_pool(
                item_embeddings, self._item_group, beh_ids.contiguous().view(B, self.maxlen, n_item), None,
                L._w['kernel'], L._w['bias'], L.activation, L._w.get('alpha'), mask_from_ids=True)
            return self._head(user_embed, item_embed, att_outputs)
        if self._beh_regular:
            n_item = len(self._item_group)
            behavior_embed = ops.gather_concat(self._item_group, beh_ids.view(B * self.maxlen, n_item))
        else:
            behavior_embed = ops.gather_concat(self._beh_group, beh_ids)
        behavior_embed = behavior_embed.view(B, self.maxlen, d_item)                  # :74
        if self.mode == 'as_written':
            att_outputs = self.attention_layer(behavior_embed)                         # :77 (B, maxlen, hs)
            att_outputs = att_outputs.reshape(-1, att_outputs.shape[2])                # :79
            if att_outputs.shape[0] != B:
                raise ValueError(f"DIN as written: concat of (B={B}, .) with (B*maxlen={att_outputs.shape[0]}, .) "
                                 "(src/ctr/din/model.py:79-81) — the reference raises here for maxlen > 1")
        else:
            # mask: history slot is real iff its FIRST item id is non-zero (pad id 0)
            n_item = len(self._item_group)
            mask = (beh_ids.view(B, self.maxlen, n_item)[:, :, 0] != 0).to(torch.float32)
            att_outputs = self.attention_layer([item_embeddings, behavior_embed, behavior_embed, mask])
        return self._head(user_embed, item_embed, att_outputs)

    def _head(self, user_embed, item_embed, att_outputs):
        ops.copy_cols(att_outputs if att_outputs.stride(1) == 1 else att_outputs.contiguous(),
                      self._all_buf[:, self._o_att:])                                  # :81
        x = self._all_buf
        for i, dense in enumerate(self.ffn):                                           # :83-87 (BN folded)
            if i == 0:
                if not dense.built:
                    dense.build(x.shape[-1])
                key = (self._version, self.bn._version, dense._version)
                if self._folded is None or self._folded[0] != key:
                    self._folded = (key, self.bn.fold(dense._w['kernel'], dense._w.get('bias')))
                x = dense.apply(x, *self._folded[1])
            else:
                x = dense(x)
        x = self.dropout(x)
        return ops.add_sigmoid(self.final_output(x))                                   # :91
