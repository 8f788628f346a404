"""TEST INFRASTRUCTURE: deterministic synthetic attention maps with clear spatial structure (for pinning the
get_token_maps port against the reference function without shipping 20 MB of real maps)."""
import torch


def synthetic_attention_maps(seed=0, n_self=3, n_cross=4):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij")
    # 5 blobs => 5 natural segments
    centers = torch.tensor([[6.0, 6.0], [6.0, 24.0], [25.0, 8.0], [24.0, 25.0], [15.0, 16.0]])
    d = ((yy[None] - centers[:, 0, None, None]) ** 2 + (xx[None] - centers[:, 1, None, None]) ** 2)
    label = d.argmin(0).reshape(-1)                                   # [1024]
    same = (label[:, None] == label[None, :]).float()
    selfm, crossm = {}, {}
    for i in range(n_self):
        a = same * 4.0 + torch.rand(1024, 1024, generator=g)
        selfm[f"self{i}"] = torch.softmax(a, -1)[None] * 3.0          # "accumulated over 3 steps"
    for i in range(n_cross):
        logits = torch.rand(1024, 77, generator=g)
        logits[:, 2] += (label == 0).float() * 3.0                    # token 2 -> blob 0
        logits[:, 3] += (label == 1).float() * 3.0
        logits[:, 6] += (label == 3).float() * 3.0
        crossm[f"cross{i}"] = torch.softmax(logits, -1)[None] * 3.0
    return selfm, crossm
