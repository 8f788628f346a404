"""CLIP text encoders on the engine's operators against transformers' CLIPTextModel / CLIPTextModelWithProjection (same random
weights): the call surface the reference uses (rd.py:53-66 `text_encoder(ids)[0]`; xl.py:330-356 `out[0]`, `out.hidden_states[-2]`)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()


@pytest.mark.parametrize("hidden,heads,layers,inter,act,proj", [(64, 2, 3, 128, "quick_gelu", None), (128, 4, 2, 256, "gelu", 96),
                                                                (768, 12, 2, 3072, "quick_gelu", None), (1280, 20, 2, 5120, "gelu", 1280)])
def test_text_encoder_matches_transformers(hidden, heads, layers, inter, act, proj):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from rich_text_to_image_amd.clip_text_encoder import HipCLIPTextEncoder
    torch.manual_seed(0)
    eos = 999
    cfg = CLIPTextConfig(vocab_size=1000, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                         max_position_embeddings=77, hidden_act=act, eos_token_id=eos, bos_token_id=998, pad_token_id=eos,
                         projection_dim=proj or 512)
    ref = (CLIPTextModelWithProjection if proj else CLIPTextModel)(cfg).eval()
    with torch.no_grad():                                      # transformers' default init is tiny: scale up so the test means something
        for n, p in ref.named_parameters():
            if p.dim() >= 2 and "embedding" not in n:
                p.mul_(3.0)
    ids = torch.randint(0, 990, (3, 77))
    ids[:, 0] = 998
    for b, n in enumerate((5, 20, 76)):
        ids[b, n:] = eos
    with torch.no_grad():
        r = ref(ids, output_hidden_states=True)
    enc = HipCLIPTextEncoder(ref.state_dict(), cfg, device=0, with_projection=bool(proj))
    o = enc(ids, output_hidden_states=True)
    e_first, e_pen = rel_l2(o[0], r[0]), rel_l2(o.hidden_states[-2], r.hidden_states[-2])
    e_last = rel_l2(o.last_hidden_state, r.last_hidden_state)
    print(f"CLIP text encoder C={hidden} L={layers} {act}: out[0] {e_first:.3e}  hidden[-2] {e_pen:.3e}  last {e_last:.3e}")
    assert len(o.hidden_states) == layers + 1 and o[0].shape == r[0].shape
    assert e_first < 2e-2 and e_pen < 2e-2 and e_last < 2e-2          # bf16 operands, fp32 accumulation / residual stream
