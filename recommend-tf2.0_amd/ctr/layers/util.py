"""src/ctr/layers/util.py: attention helpers.  `modules.py` imports them but never uses them
(src/ctr/layers/modules.py:13); kept for surface completeness on top of the HIP row-mask kernel."""
import torch

from recamd import ops


def split_heads(x, seq_len, num_heads, depth):
    """(B, S, H*depth) -> (B, H, S, depth) view (src/ctr/layers/util.py:38-49)."""
    return x.reshape(-1, seq_len, num_heads, depth).permute(0, 2, 1, 3)


def scaled_dot_product_attention(q, k, v, mask=None):
    """src/ctr/layers/util.py:12-35 on (B, H, S, dk) tensors.  mask=None replaces EVERY logit by
    the padding value (:27-30) => uniform attention; a mask (B,H,S,1) masks whole query rows."""
    B, H, Sq, dk = q.shape
    Sk = k.shape[2]
    merge = lambda t: t.permute(0, 2, 1, 3).reshape(B, t.shape[2], H * dk).contiguous()  # noqa: E731
    if mask is None:
        m = torch.zeros((B, Sq), dtype=torch.float32, device=q.device)
    else:
        m = mask[:, 0, :, 0].to(torch.float32).contiguous()
    out = ops.mha_rowmask(merge(q), merge(k), merge(v), m, H)
    return out.reshape(B, Sq, H, dk).permute(0, 2, 1, 3)
