"""Mirror of src/match/sasrec/model.py on the HIP kernels.

call([seq (B,S) int32, pos (B,1) int32, neg (B,neg_len) int32]) -> logits (B, 1+neg_len); the
BCE-style loss of :93-95 is kept in `self.losses[-1]`.  Three DIFFERENT tables seq/pos/neg
(:75-79), no positional embedding (:74), mask = (seq != 0) (:72), pad rows multiplied by 0 (:82).
Exact work savings: the last encoder block only encodes the final query row (only x[:, -1] is
consumed, :88) and the pos/neg lookups are fused with their dot products (K10).

`sharded=(rank, world)` (or a ready recamd.dist.ShardedTables): BASELINE configs[4] — the seq/pos/neg tables are
row-sharded cyclically over the ranks and the three lookups of :75-79 travel in ONE exchange (ids -> unique ids ->
RCCL all-to-all -> owner gather -> all-to-all back); pad id 0 of the sequence is dropped before the exchange (its
row is multiplied by 0 at :82 anyway).  The attention / dot-score kernels then read the returned rows through the
per-lookup index exactly as they read a table through ids, so everything after the lookup is unchanged."""
import torch

from match.layers.modules import TransformerEncoder
from recamd import nn, ops
from recamd.nn import Model, to_device_ids


class SASRec(Model):
    def __init__(self, user_sparse_feature_columns, item_sparse_feature_columns,
                 user_dense_feature_columns=(), item_dense_feature_columns=(),
                 blocks=1, num_heads=1, att_hidden_unit=128, ffn_hidden_unit=128,
                 dnn_dropout=0., layer_norm_eps=1e-6, seq_len=10, neg_len=100, embed_reg=1e-6,
                 last_row_only=True, sharded=None, shard_factory=None, fused=True):
        super().__init__()
        from recamd.dist import ShardedTables, local_rows_of
        if isinstance(sharded, ShardedTables):
            self._shard_rw, self._sharded = (sharded.rank, sharded.world), sharded
        else:
            self._shard_rw, self._sharded = (tuple(sharded) if sharded is not None else None), None
        self.seq_len = seq_len
        self.embed_reg = embed_reg
        self.neg_len = neg_len
        self.user_sparse_feature_columns = user_sparse_feature_columns
        self.user_dense_feature_columns = user_dense_feature_columns
        self.item_sparse_feature_columns = item_sparse_feature_columns
        self.item_dense_feature_columns = item_dense_feature_columns
        self.d_model = att_hidden_unit
        rw = self._shard_rw

        def rows_of(feat):  # a rank of a sharded model holds rows r with r % world == rank
            return feat['feat_num'] if rw is None else local_rows_of(feat['feat_num'], rw[0], rw[1])
        self.user_embed_layers = {
            'embed_' + str(feat['feat']): self.track('user_embed_' + str(feat['feat']), nn.Embedding(
                input_dim=rows_of(feat), input_length=feat['feat_len'], output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for feat in self.user_sparse_feature_columns
        }
        if rw is not None and self._sharded is None:
            names = ('seq_item', 'pos_item', 'neg_item')
            feats = {f['feat']: f for f in self.user_sparse_feature_columns}
            # shard_factory: a ShardedTables subclass / factory with the same signature (tests simulate ranks with one)
            self._sharded = (shard_factory or ShardedTables)([self.user_embed_layers['embed_' + k].table for k in names],
                                                             [feats[k]['feat_num'] for k in names], rw[0], rw[1])

            def point_layers(st):  # the row space is the single source of truth: the layers' weights are views of it
                for f, k in enumerate(names):
                    layer = self.user_embed_layers['embed_' + k]
                    layer._w['embeddings'] = st.tables[f][:layer.input_dim]
            point_layers(self._sharded)
            self._sharded.on_rebuild.append(point_layers)   # receive slots sized by the first lookup: views move once
        self.item_embed_layers = {
            'embed_' + str(feat['feat']): self.track('item_embed_' + str(feat['feat']), nn.Embedding(
                input_dim=feat['feat_num'], input_length=feat['feat_len'], output_dim=feat['embed_dim'],
                embeddings_initializer='random_uniform'))
            for feat in self.item_sparse_feature_columns
        }
        self.dropout = nn.Dropout(dnn_dropout)
        self.encoder_layer = [self.track('encoder_%d' % i, TransformerEncoder(
            self.d_model, num_heads, ffn_hidden_unit, dnn_dropout, layer_norm_eps)) for i in range(blocks)]
        self.last_row_only = last_row_only
        self.fused = fused          # False: the layer-by-layer path even where the one-launch kernel applies
        self._logits = None
        self.embed = None

    def call(self, inputs, **kwargs):
        seq_inputs, pos_inputs, neg_inputs = [to_device_ids(t, self.device) for t in inputs]
        B, S = seq_inputs.shape
        fused = self._fused_block(S, seq_inputs, pos_inputs, neg_inputs)
        if fused and self._sharded is None:
            # everything from the ids to the logits in ONE launch: pad id 0 -> zero row inside the kernel (:81-82)
            tb = self.user_embed_layers
            logits, seq_info = ops.sasrec_last_row(
                fused, self.encoder_layer[0].layernorm1.epsilon, self.encoder_layer[0].layernorm2.epsilon,
                tb['embed_seq_item'].table, seq_inputs, 0, seq_inputs[:, -1], seq_inputs.stride(0),
                tb['embed_pos_item'].table, pos_inputs, tb['embed_neg_item'].table, neg_inputs)
            self.embed = seq_info[:, None, :]
            self._logits = logits
            return logits
        mask = (seq_inputs != 0).to(torch.float32)                                        # :72 (B,S)
        if self._sharded is None:
            # :75 + :81-82 in one pass: `seq_embed * mask` zeroes exactly the rows whose id is 0, so pad ids
            # are sent to the gather as out-of-range (-1) and read as zero rows (finite tables: identical)
            seq_m = torch.where(seq_inputs == 0, torch.full_like(seq_inputs, -1), seq_inputs)
            seq_table = self.user_embed_layers['embed_seq_item'].table
            pos_table = self.user_embed_layers['embed_pos_item'].table
            neg_table = self.user_embed_layers['embed_neg_item'].table
        else:
            # the three lookups of :75-79 in ONE exchange; afterwards "table" = the returned rows, "ids" = uidx
            st = self._sharded
            n_neg = neg_inputs.shape[1]
            vids = torch.cat([st.virtual_ids(0, seq_inputs, pad_id=0).reshape(-1), st.virtual_ids(1, pos_inputs).reshape(-1),
                              st.virtual_ids(2, neg_inputs).reshape(-1)])
            rows, uidx = st.lookup_rows(vids)
            seq_m = uidx[:B * S].view(B, S)
            pos_inputs = uidx[B * S: B * S + B].view(B, 1)
            neg_inputs = uidx[B * S + B:].view(B, n_neg)
            seq_table = pos_table = neg_table = rows
            if fused:   # the same single launch, reading the returned rows through the per-lookup index
                logits, seq_info = ops.sasrec_last_row(
                    fused, self.encoder_layer[0].layernorm1.epsilon, self.encoder_layer[0].layernorm2.epsilon,
                    rows, seq_m, -1, seq_inputs[:, -1], seq_inputs.stride(0), rows, pos_inputs, rows, neg_inputs)
                self.embed = seq_info[:, None, :]
                self._logits = logits
                return logits

        def embed(table, ids):                                                            # Embedding.call on `table`
            return ops.gather_concat(ops.TableGroup([table]), ids.reshape(-1, 1).contiguous()).view(*ids.shape, -1)
        nb = len(self.encoder_layer)
        seq_info = None
        if nb == 1 and self.last_row_only and self.encoder_layer[0].mha.num_heads == 1 and \
                self.d_model in (16, 32, 64) and seq_m.dtype == torch.int32:
            # one block, last row only, one head: the attention reads the item table directly by id (fused lookup),
            # only the last position's embedding is materialised (query row + residual)
            last = embed(seq_table, seq_m[:, -1:].contiguous())                           # (B,1,d)
            seq_info = self.encoder_layer[0]([None, mask], query_rows=last, query_mask=mask[:, -1:].contiguous(),
                                             out_mask=mask[:, -1].contiguous(),
                                             gather=(seq_table, seq_m.contiguous()))[:, 0, :]
            nb = 0
        else:
            att_outputs = embed(seq_table, seq_m)                                         # (B,S,d)
        for i, block in enumerate(self.encoder_layer[:nb] if nb else []):
            if i == nb - 1 and self.last_row_only:
                seq_info = block([att_outputs, mask], query_rows=att_outputs[:, -1:, :].contiguous(),
                                 query_mask=mask[:, -1:].contiguous(),
                                 out_mask=mask[:, -1].contiguous())[:, 0, :]              # :85-88
            else:
                att_outputs = block([att_outputs, mask], out_mask=mask.reshape(-1))        # :85-86
        if seq_info is None:
            seq_info = att_outputs[:, -1].contiguous()                                    # :88
        self.embed = seq_info[:, None, :]
        logits = torch.empty((B, 1 + neg_inputs.shape[1]), dtype=torch.float32, device=self.device)
        ops.gather_dot_scores(seq_info, pos_table, pos_inputs.contiguous(), out=logits[:, :1])   # :77,:90
        ops.gather_dot_scores(seq_info, neg_table, neg_inputs.contiguous(), out=logits[:, 1:])   # :79,:91
        self._logits = logits          # the add_loss value (:93-95) is computed on demand: `model.losses`
        return logits                                                                     # :96

    def _fused_block(self, S, seq_inputs, pos_inputs, neg_inputs):
        """The 13 weight tensors of the single encoder block when the one-launch kernel serves this configuration
        (one block, one head, last row only, d_model 64, ffn 64/128, int32 ids), else None.  `self.fused = False` keeps
        the layer-by-layer path."""
        if len(self.encoder_layer) != 1 or not self.last_row_only or not self.fused:
            return None
        enc = self.encoder_layer[0]
        if enc.mha.num_heads != 1 or any(t.dtype != torch.int32 for t in (seq_inputs, pos_inputs, neg_inputs)):
            return None
        ffn_hidden = enc.ffn.conv1.units
        if not ops.sasrec_last_row_supported(self.d_model, ffn_hidden, S, pos_inputs.shape[1] + neg_inputs.shape[1]):
            return None
        d = self.d_model
        for layer, n_in in ((enc.mha.wq, d), (enc.mha.wk, d), (enc.mha.wv, d), (enc.ffn.conv1, d), (enc.ffn.conv2, ffn_hidden),
                            (enc.layernorm1, d), (enc.layernorm2, d)):
            if not layer.built:
                layer.build(n_in)
        w = lambda l, k: l._w[k]  # noqa: E731
        return (w(enc.mha.wq, 'kernel'), w(enc.mha.wq, 'bias'), w(enc.mha.wk, 'kernel'), w(enc.mha.wv, 'kernel'),
                w(enc.mha.wv, 'bias'), w(enc.layernorm1, 'gamma'), w(enc.layernorm1, 'beta'), w(enc.ffn.conv1, 'kernel'),
                w(enc.ffn.conv1, 'bias'), w(enc.ffn.conv2, 'kernel'), w(enc.ffn.conv2, 'bias'), w(enc.layernorm2, 'gamma'),
                w(enc.layernorm2, 'beta'))

    @property
    def losses(self):
        """[mean(-log sigmoid(pos) - log(1 - sigmoid(neg))) / 2] of the last call (src/match/sasrec/model.py:93-95;
        (B,1) + (B,neg) broadcasting).  Lazy: a forward pass used for scoring does not pay for it."""
        if self._logits is None:
            return []
        return [ops.pairwise_rank_loss(self._logits)[0]]
