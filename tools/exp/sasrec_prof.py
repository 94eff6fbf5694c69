import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from match.sasrec.model import SASRec
dev = torch.device("cuda:0")
B, S, n, V, d = 8192, 200, 100, 10_000_000, 64
uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': d}, {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': d},
      {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': d}]
m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n, last_row_only=True)
lens = torch.randint(1, S + 1, (B,), device=dev)
seq = torch.randint(1, V, (B, S), device=dev, dtype=torch.int32)
seq[torch.arange(S, device=dev)[None, :] < (S - lens)[:, None]] = 0
pos = torch.randint(1, V, (B, 1), device=dev, dtype=torch.int32)
neg = torch.randint(1, V, (B, n), device=dev, dtype=torch.int32)
for _ in range(20):
    m([seq, pos, neg])
torch.cuda.synchronize()
