"""per-kernel split of one training step (run under rocprofv3 --kernel-trace --stats): python tools/exp/train_prof.py autoint|sasrec|dlrm"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
from recamd import train as tr
rng = np.random.default_rng(0)
which = sys.argv[1]
V = 100_000
if which == "autoint":
    from ctr.autoint.model import AutoInt
    F, nd, B = 26, 13, 4096
    m = AutoInt([[{'feat': f'I{i}'} for i in range(nd)], [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': 16} for i in range(F)]],
                att_hidden_units=16, head_num=2, att_layer_num=3, use_res=True)
    inputs, y = [rng.random((B, nd)).astype(np.float32), rng.integers(0, V, size=(B, F)).astype(np.int32)], (rng.random(B) < 0.3).astype(np.float32)
elif which == "dlrm":
    from ctr.dlrm.model import DLRM
    F, nd, B = 26, 13, 8192
    m = DLRM([[{'feat': f'I{i}'} for i in range(nd)], [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': 128} for i in range(F)]],
             [512, 256, 128], [1024, 512, 256], interaction='dot')
    inputs, y = [rng.random((B, nd)).astype(np.float32), rng.integers(0, V, size=(B, F)).astype(np.int32)], (rng.random(B) < 0.3).astype(np.float32)
else:
    from match.sasrec.model import SASRec
    S, n_neg, B, d = 200, 100, 512, 64
    cols = [{'feat': k, 'feat_num': V, 'feat_len': n, 'embed_dim': d} for k, n in (('seq_item', S), ('pos_item', 1), ('neg_item', n_neg))]
    m = SASRec(cols, [], att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n_neg)
    seq = rng.integers(1, V, size=(B, S)).astype(np.int32)
    inputs, y = [seq, rng.integers(1, V, size=(B, 1)).astype(np.int32), rng.integers(1, V, size=(B, n_neg)).astype(np.int32)], None
m(inputs)
opt, state = tr.Adam(m, 1e-3, l2=tr.default_l2(m)), tr.TrainState(m)
for _ in range(4):
    tr.train_step(m, opt, state, inputs, y)
torch.cuda.synchronize()
