import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from ctr.dlrm.model import DLRM
from ctr.deep_fm.model import DeepFM
dev = torch.device("cuda:0")
B, F, D, V, ND = 65536, 26, 128, 200_000, 13
sparse = [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': D} for i in range(F)]
dense_c = [{'feat': f'I{i}'} for i in range(ND)]
ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
dense = torch.rand((B, ND), device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "dlrm"
m = DLRM([dense_c, sparse], [512, 256, 128], [1024, 1024, 512, 256], interaction='dot') if which == "dlrm" else DeepFM([dense_c, sparse], hidden_units=(256, 128, 64))
for _ in range(12):
    m([dense, ids])
torch.cuda.synchronize()
