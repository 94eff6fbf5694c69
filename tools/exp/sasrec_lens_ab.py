"""One-launch SASRec kernel: time with uniform lengths 1..S vs every sample at the mean length (load-balance check)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from match.sasrec.model import SASRec
dev = torch.device("cuda:0")
B, S, n, V, d = 8192, 200, 100, 10_000_000, 64
uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': d}, {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': d},
      {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': d}]
m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n, last_row_only=True)


def batch(kind, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    if kind == "uniform":
        lens = torch.randint(1, S + 1, (B,), device=dev, generator=g)
    elif kind == "const":
        lens = torch.full((B,), 100, device=dev)
    else:  # sorted: uniform lengths, samples sorted by length (neighbouring samples alike)
        lens = torch.randint(1, S + 1, (B,), device=dev, generator=g).sort().values
    seq = torch.randint(1, V, (B, S), device=dev, dtype=torch.int32, generator=g)
    seq[torch.arange(S, device=dev)[None, :] < (S - lens)[:, None]] = 0
    pos = torch.randint(1, V, (B, 1), device=dev, dtype=torch.int32, generator=g)
    neg = torch.randint(1, V, (B, n), device=dev, dtype=torch.int32, generator=g)
    return [seq, pos, neg], float(lens.float().mean())


for kind in ("uniform", "const", "sorted", "uniform"):
    bs = [batch(kind, s) for s in range(8)]
    for i in range(200):
        m(bs[i % 8][0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(400):
        m(bs[i % 8][0])
    e1.record(); torch.cuda.synchronize()
    print(kind, "mean len %.1f" % bs[0][1], "us/forward %.1f" % (e0.elapsed_time(e1) * 1000 / 400))
