"""LoRA merge at load (f4): the three key layouts against the explicit low-rank arithmetic, on the tiny UNet's state dict."""
import pytest
import torch

from oracle.unet import TINY_SD_CONFIG, random_state_dict


def _pair(w, rank, g):
    o, i = w.shape[0], w.shape[1]
    return torch.randn(rank, i, generator=g) * 0.1, torch.randn(o, rank, generator=g) * 0.1


def test_merge_lora_layouts_and_errors():
    from rich_text_to_image_amd.lora import merge_lora
    sd = random_state_dict(TINY_SD_CONFIG, seed=2)
    g = torch.Generator().manual_seed(0)
    t1 = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"
    t2 = "mid_block.attentions.0.transformer_blocks.0.attn2.to_out.0"
    t3 = "up_blocks.1.attentions.1.transformer_blocks.0.ff.net.2"
    t4 = "down_blocks.1.attentions.0.proj_in"                                   # 1x1 conv in SD-v1.5 (use_linear_projection False)
    d1, u1 = _pair(sd[t1 + ".weight"], 4, g); d2, u2 = _pair(sd[t2 + ".weight"], 8, g); d3, u3 = _pair(sd[t3 + ".weight"], 2, g)
    w4 = sd[t4 + ".weight"]
    d4, u4 = torch.randn(4, w4.shape[1], 1, 1, generator=g) * 0.1, torch.randn(w4.shape[0], 4, 1, 1, generator=g) * 0.1
    lora = {
        "lora_unet_" + t1.replace(".", "_") + ".lora_down.weight": d1, "lora_unet_" + t1.replace(".", "_") + ".lora_up.weight": u1,
        "lora_unet_" + t1.replace(".", "_") + ".alpha": torch.tensor(2.0),                                   # kohya, alpha != rank
        "unet." + t2[:-len(".to_out.0")] + ".processor.to_out_lora.down.weight": d2,                         # diffusers 0.18
        "unet." + t2[:-len(".to_out.0")] + ".processor.to_out_lora.up.weight": u2,
        "unet." + t3 + ".lora_A.weight": d3, "unet." + t3 + ".lora_B.weight": u3,                            # peft
        "lora_unet_" + t4.replace(".", "_") + ".lora_down.weight": d4, "lora_unet_" + t4.replace(".", "_") + ".lora_up.weight": u4,
        "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight": torch.zeros(2, 8),                  # reported, not merged
    }
    out, rep = merge_lora(sd, lora, scale=0.7)
    assert torch.allclose(out[t1 + ".weight"], sd[t1 + ".weight"] + 0.7 * (2.0 / 4) * (u1 @ d1), atol=1e-6)
    assert torch.allclose(out[t2 + ".weight"], sd[t2 + ".weight"] + 0.7 * (u2 @ d2), atol=1e-6)
    assert torch.allclose(out[t3 + ".weight"], sd[t3 + ".weight"] + 0.7 * (u3 @ d3), atol=1e-6)
    assert torch.allclose(out[t4 + ".weight"], w4 + 0.7 * (u4.flatten(1) @ d4.flatten(1))[:, :, None, None], atol=1e-6)
    assert sorted(rep["merged"]) == sorted(t + ".weight" for t in (t1, t2, t3, t4)) and len(rep["text_encoder"]) == 1
    assert set(rep["alpha_default"]) == {t2, t3, t4}
    untouched = [k for k in sd if k not in rep["merged"]]
    assert all(out[k] is sd[k] for k in untouched)
    # the adapted projection really computes W x + s * up(down(x))
    x = torch.randn(5, sd[t1 + ".weight"].shape[1], generator=g)
    assert torch.allclose(x @ out[t1 + ".weight"].t(), x @ sd[t1 + ".weight"].t() + 0.7 * 0.5 * (x @ d1.t()) @ u1.t(), atol=1e-5)
    with pytest.raises(KeyError):
        merge_lora(sd, {"lora_unet_no_such_module.lora_down.weight": d1, "lora_unet_no_such_module.lora_up.weight": u1})
    with pytest.raises(KeyError):
        merge_lora(sd, {"lora_unet_" + t1.replace(".", "_") + ".lora_down.weight": d1})
