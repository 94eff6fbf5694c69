import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import ref_pipeline as rp
from recamd.pipeline import MinMaxScaler
rng = np.random.default_rng(700); M, N = 700, 13
x = (rng.integers(-50, 100000, size=(M, N)) + rng.random((M, N)) * 0.9).astype(np.float32)
x[:, -1] = 3.7
sc = MinMaxScaler(); got = sc.fit_transform(torch.from_numpy(x).cuda()).cpu().numpy()
mn, mx = rp.minmax_fit(x); e = rp.minmax_transform(x, mn, mx)
bad = np.argwhere(got != e)
print(len(bad), "mismatches")
for i, j in bad[:8]:
    print(i, j, x[i, j], got[i, j].view(np.uint32) if False else repr(got[i, j]), repr(e[i, j]), mn[j], mx[j])
