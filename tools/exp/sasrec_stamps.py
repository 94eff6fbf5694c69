"""round 3: where do a SASRec sample's ~20 us go?  Builds sasrec_fused.hip with -DREC_SASREC_STAMPS (s_memtime at the phase
boundaries, written over seq_info[b, 0..7]) on the box, runs the configs[4] forward, prints the mean phase durations and the
per-wave timeline (a wave's two samples: b and b + 4096).  s_memtime ticks = shader cycles (~2.1 GHz under this load)."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
subprocess.check_call(["touch", os.path.join(ROOT, "recommend-tf2.0_amd/csrc/sasrec_fused.hip")])
subprocess.check_call(["make", "-C", os.path.join(ROOT, "recommend-tf2.0_amd/csrc"), "EXTRA_HIPFLAGS=-DREC_SASREC_STAMPS"],
                      stdout=subprocess.DEVNULL)
sys.path.insert(0, os.path.join(ROOT, "recommend-tf2.0_amd"))
import torch
from match.sasrec.model import SASRec
dev = torch.device("cuda:0")
B, S, n, V, d = 8192, 200, 100, 10_000_000, 64
uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': d}, {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': d},
      {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': d}]
m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n)
gen = torch.Generator(device=dev).manual_seed(5)
batches = []
for j in range(4):
    lens = torch.randint(1, S + 1, (B,), device=dev, generator=gen)
    seq = torch.randint(1, V, (B, S), device=dev, dtype=torch.int32, generator=gen)
    seq[torch.arange(S, device=dev)[None, :] < (S - lens)[:, None]] = 0
    batches.append([seq, torch.randint(1, V, (B, 1), device=dev, dtype=torch.int32, generator=gen),
                    torch.randint(1, V, (B, n), device=dev, dtype=torch.int32, generator=gen)])
for i in range(40):
    m(batches[i % 4])
torch.cuda.synchronize()
st = m.embed[:, 0, :9].cpu().numpy()      # seq_info[b, 0..8]
names = ["ids wait", "compaction + first issues", "x_last + Wq + WkT", "attention loop", "Wv/LN/FFN/LN chain", "candidate loop"]
tick_ns = 1.0 / 2.1
tot = st[:, :6].sum(1)
print("samples", len(st), "mean real rows", st[:, 6].mean())
for i, nm in enumerate(names):
    print(f"  {nm:28s} mean {st[:, i].mean() * tick_ns / 1e3:6.2f} us   p90 {sorted(st[:, i])[int(.9 * len(st))] * tick_ns / 1e3:6.2f} us")
print(f"  {'sample total':28s} mean {tot.mean() * tick_ns / 1e3:6.2f} us   max {tot.max() * tick_ns / 1e3:6.2f} us")
import numpy as np
first = st[:, 7]
print("att loop us vs real rows (binned):", [(int(lo), round(float(st[(st[:, 6] >= lo) & (st[:, 6] < lo + 50), 3].mean()) * tick_ns / 1e3, 2)) for lo in (0, 50, 100, 150)])

t0 = st[:, 7]                                   # start stamp (low 24 bits) of every sample
nw = 4096
first, second = st[:nw], st[nw:2 * nw]
# s_memtime is per XCD: only differences inside one wave mean something
busy = np.mod(second[:, 7] - first[:, 7], 2 ** 24) + second[:, :6].sum(1)       # first sample start -> second sample end
q = lambda a, p: float(np.sort(a)[int(p * (len(a) - 1))])
print(f"  a wave's two samples, start to end: mean {busy.mean() / 2100:6.2f}  p50 {q(busy, .5) / 2100:6.2f}  p90 {q(busy, .9) / 2100:6.2f}  "
      f"p99 {q(busy, .99) / 2100:6.2f}  max {busy.max() / 2100:6.2f} us   (the kernel: ~85 us incl. weight staging)")
print("  gap between a wave's samples (mean us):", float((np.mod(second[:, 7] - first[:, 7], 2 ** 24) - first[:, :6].sum(1)).mean()) / 2100)
pro = first[:, 8]
print(f"  kernel entry -> first sample starts (weight staging + barrier): mean {pro.mean() / 2100:6.2f}  p90 {q(pro, .9) / 2100:6.2f}  max {pro.max() / 2100:6.2f} us")
done = first[:, 8] + busy
print(f"  kernel entry -> wave done: mean {done.mean() / 2100:6.2f}  p99 {q(done, .99) / 2100:6.2f}  max {done.max() / 2100:6.2f} us")
