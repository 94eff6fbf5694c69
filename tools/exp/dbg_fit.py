import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import ref_train as rt
from recamd import train as tr
from tests.test_training_gpu import _setup, tr_weights
dev = torch.device("cuda:0")
rng = np.random.default_rng(21)
m, okind, kw, inputs, y = _setup("deepfm", dev, rng, B=400, scale=0.05)
w_init = m.get_weights()
m.set_weights({k: (np.zeros_like(v) if k.endswith("moving_mean") else np.ones_like(v)) for k, v in w_init.items() if "moving_" in k})
m.set_weights({k: (v * 0.1).astype(np.float32) for k, v in w_init.items() if k.endswith("embeddings")})
W = {k: v.astype(np.float64) for k, v in tr_weights(m).items()}
print("before: infer diff", np.abs(m(inputs).cpu().numpy().reshape(-1) - rt.predict(okind, W, inputs)).max())
l2 = tr.default_l2(m)
opt = tr.Adam(m, 5e-3, l2=l2); st = tr.TrainState(m); oo = rt.AdamOracle(lr=5e-3)
tr.train_step(m, opt, st, inputs, y); rt.train_step(okind, W, oo, inputs, y, l2)
got = tr_weights(m)
for k in W:
    d = np.abs(got[k] - W[k]).max()
    if d > 1e-5: print("weight diff", k, d)
print("after: infer diff", np.abs(m(inputs).cpu().numpy().reshape(-1) - rt.predict(okind, W, inputs)).max())
sub = [inputs[0][320:], inputs[1][320:]]
print("after: infer diff on slice", np.abs(m(sub).cpu().numpy().reshape(-1) - rt.predict(okind, W, sub)).max())
