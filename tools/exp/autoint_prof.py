import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from ctr.autoint.model import AutoInt
dev = torch.device("cuda:0")
B, F, nd, D = 4096, 26, 13, 16
fc = [[{'feat': f'I{i}'} for i in range(nd)], [{'feat': f'C{i}', 'feat_num': 100_000, 'embed_dim': D} for i in range(F)]]
m = AutoInt(fc, att_hidden_units=16, head_num=2, att_layer_num=3, use_res=True)
dense = torch.rand((B, nd), device=dev)
ids = torch.randint(0, 100_000, (B, F), device=dev, dtype=torch.int32)
for _ in range(20):
    m([dense, ids])
torch.cuda.synchronize()
