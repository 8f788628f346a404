"""LoRA checkpoints as a weight merge at load time (the reference's `lora` branch / diffusers' LoraLoaderMixin):
W' = W + scale * (alpha / rank) * up @ down for every adapted UNet projection, so the engine runs the merged model at full speed
and needs no LoRA processors.  Key layouts understood (UNet part; text-encoder adapters are reported, not merged):
  kohya / civitai     lora_unet_<module path with '_'>.lora_down.weight | .lora_up.weight | .alpha
  diffusers 0.18      unet.<path>.attn1.processor.to_q_lora.down.weight | .up.weight      (to_out_lora -> to_out.0)
  diffusers >= 0.20   unet.<path>.to_q.lora.down.weight | .lora.up.weight,  peft: .lora_A.weight | .lora_B.weight
Pure tensor plumbing on the checkpoint (runs once, before `rt_bind_weight`); no arithmetic on the sampling path.
"""
import re

import torch


def _targets(unet_sd):
    """module path (without '.weight') -> key, plus the kohya spelling of every path."""
    paths = {k[:-7]: k for k in unet_sd if k.endswith(".weight") and unet_sd[k].dim() in (2, 4)}
    kohya = {"lora_unet_" + p.replace(".", "_"): p for p in paths}
    return paths, kohya


def _split(key):
    """-> (module id, role) with role in {'down', 'up', 'alpha'} or None when the key is not a LoRA tensor."""
    for suf, role in ((".lora_down.weight", "down"), (".lora_up.weight", "up"), (".alpha", "alpha"), (".lora.down.weight", "down"),
                      (".lora.up.weight", "up"), (".lora_A.weight", "down"), (".lora_B.weight", "up"), ("_lora.down.weight", "down"),
                      ("_lora.up.weight", "up")):
        if key.endswith(suf):
            return key[:-len(suf)], role
    return None, None


def merge_lora(unet_state_dict, lora_state_dict, scale=1.0):
    """Returns (merged_state_dict, report).  report = dict(merged=[...], text_encoder=[...], alpha_default=[...])."""
    paths, kohya = _targets(unet_state_dict)
    groups = {}
    report = dict(merged=[], text_encoder=[], alpha_default=[])
    for key, t in lora_state_dict.items():
        mod, role = _split(key)
        if mod is None:
            raise KeyError(f"not a LoRA tensor: {key}")
        if mod.startswith("lora_te") or mod.startswith("text_encoder"):
            report["text_encoder"].append(key)
            continue
        if mod in kohya:
            path = kohya[mod]
        else:
            path = mod[5:] if mod.startswith("unet.") else mod
            path = re.sub(r"\.processor\.(to_[qkv])$", r".\1", path)                 # diffusers 0.18 attention-processor layout
            path = re.sub(r"\.processor\.to_out$", ".to_out.0", path)
            if path.endswith(".to_out") and path + ".0" in paths:
                path += ".0"
        if path not in paths:
            raise KeyError(f"LoRA tensor {key} targets '{path}', which is not a weight of this UNet")
        groups.setdefault(path, {})[role] = t
    out = dict(unet_state_dict)
    for path, g in groups.items():
        if "down" not in g or "up" not in g:
            raise KeyError(f"LoRA pair for '{path}' is incomplete: {sorted(g)}")
        W = unet_state_dict[paths[path]]
        down, up = g["down"].float(), g["up"].float()
        rank = down.shape[0]
        if "alpha" in g:
            alpha = float(g["alpha"])
        else:
            alpha = float(rank); report["alpha_default"].append(path)
        if down.dim() == 4 or up.dim() == 4:
            if down.shape[2:] != (1, 1) and up.shape[2:] != (1, 1):
                raise NotImplementedError(f"LoRA on a {tuple(down.shape[2:])} convolution ({path}) is not supported")
            if down.shape[2:] != (1, 1):                                               # 3x3 down, 1x1 up: delta is a 3x3 kernel
                delta = torch.einsum("or,rikl->oikl", up.flatten(1), down)
            else:
                delta = (up.flatten(1) @ down.flatten(1)).reshape(up.shape[0], down.shape[1], *([1, 1] if W.dim() == 4 else []))
        else:
            delta = up @ down
        if W.dim() == 4 and delta.dim() == 2:
            delta = delta[:, :, None, None]
        if delta.shape != W.shape:
            raise ValueError(f"LoRA delta {tuple(delta.shape)} does not fit {paths[path]} {tuple(W.shape)}")
        out[paths[path]] = (W.float() + scale * (alpha / rank) * delta).to(W.dtype)
        report["merged"].append(paths[path])
    return out, report
