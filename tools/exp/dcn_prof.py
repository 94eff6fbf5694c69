import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from ctr.dcn.model import DCN
dev = torch.device("cuda:0")
B, F, D, V = 65536, 26, 128, 200_000
sparse = [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': D} for i in range(F)]
ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
m = DCN(sparse, hidden_units=(256, 128, 64))
for _ in range(12):
    m(ids)
torch.cuda.synchronize()
