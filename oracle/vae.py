"""ORACLE (test infrastructure): AutoencoderKL *decoder* restated in functional fp32 torch.

The reference calls `self.vae.decode(...)` inside the colour-guidance step (models/region_diffusion.py:151-168,
models/region_diffusion_sdxl.py:849-867) and for the final image (rd.py:227-236, xl.py:916-944).  `AutoencoderKL`
is diffusers 0.18.2 code (environment.yaml:17) that is NOT on disk here, so this restates the published
architecture (`models/autoencoder_kl.py`, `models/vae.py:Decoder`, `unet_2d_blocks.py:UNetMidBlock2D / UpDecoderBlock2D`)
from memory => PARITY UNPINNED against diffusers for everything in this file; the HIP path is pinned against THIS
restatement (forward and, through torch autograd, the input gradient).

State-dict key names follow diffusers 0.18.2 (`decoder.mid_block.attentions.0.to_q` ...; the pre-0.18
`query/key/value/proj_attn` names of old checkpoints are accepted as aliases by the engine).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

SD_VAE_CONFIG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, latent_channels=4,
                     out_channels=3, scaling_factor=0.18215)
SDXL_VAE_CONFIG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, latent_channels=4,
                       out_channels=3, scaling_factor=0.13025)
TINY_VAE_CONFIG = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_num_groups=8, latent_channels=4,
                       out_channels=3, scaling_factor=0.18215)


def vae_decoder_shapes(cfg):
    s = OrderedDict()
    boc = cfg["block_out_channels"]
    lc = cfg["latent_channels"]

    def conv(n, i, o, k):
        s[n + ".weight"] = (o, i, k, k); s[n + ".bias"] = (o,)

    def norm(n, c):
        s[n + ".weight"] = (c,); s[n + ".bias"] = (c,)

    def lin(n, i, o):
        s[n + ".weight"] = (o, i); s[n + ".bias"] = (o,)

    def resnet(n, i, o):
        norm(n + ".norm1", i); conv(n + ".conv1", i, o, 3); norm(n + ".norm2", o); conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1)
    conv("post_quant_conv", lc, lc, 1)
    top = boc[-1]
    conv("decoder.conv_in", lc, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    lin(a + ".to_q", top, top); lin(a + ".to_k", top, top); lin(a + ".to_v", top, top); lin(a + ".to_out.0", top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, c in enumerate(rev):
        prev, out_c = out_c, c
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", boc[0], cfg["out_channels"], 3)
    return s


def random_vae_state_dict(cfg, seed=0):
    import math
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in vae_decoder_shapes(cfg).items():
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        elif name.endswith(".weight"):
            sd[name] = 1.0 + 0.1 * (torch.rand(shape, generator=g) * 2 - 1)
        else:
            sd[name] = 0.05 * (torch.rand(shape, generator=g) * 2 - 1)
    return sd


class OracleVAEDecoder:
    def __init__(self, cfg, state_dict):
        self.cfg = dict(cfg)
        self.sd = {k: v.detach().float() for k, v in state_dict.items()}
        self.G = cfg["norm_num_groups"]

    def _conv(self, x, n, padding=1):
        return F.conv2d(x, self.sd[n + ".weight"], self.sd[n + ".bias"], padding=padding)

    def _gn(self, x, n):
        return F.group_norm(x, self.G, self.sd[n + ".weight"], self.sd[n + ".bias"], 1e-6)

    def _resnet(self, x, n):
        h = self._conv(F.silu(self._gn(x, n + ".norm1")), n + ".conv1")
        h = self._conv(F.silu(self._gn(h, n + ".norm2")), n + ".conv2")
        if (n + ".conv_shortcut.weight") in self.sd:
            x = self._conv(x, n + ".conv_shortcut", padding=0)
        return x + h

    def _attn(self, x, n):
        B, C, H, W = x.shape
        h = self._gn(x, n + ".group_norm").reshape(B, C, H * W).transpose(1, 2)
        q = F.linear(h, self.sd[n + ".to_q.weight"], self.sd[n + ".to_q.bias"])
        k = F.linear(h, self.sd[n + ".to_k.weight"], self.sd[n + ".to_k.bias"])
        v = F.linear(h, self.sd[n + ".to_v.weight"], self.sd[n + ".to_v.bias"])
        p = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1)          # one head of dim C
        o = F.linear(p @ v, self.sd[n + ".to_out.0.weight"], self.sd[n + ".to_out.0.bias"])
        return x + o.transpose(1, 2).reshape(B, C, H, W)

    def decode(self, z):
        """z = latents / scaling_factor, [B,4,h,w] -> image in [-1,1], [B,3,8h,8w]"""
        x = self._conv(z.float(), "post_quant_conv", padding=0)
        x = self._conv(x, "decoder.conv_in")
        x = self._resnet(x, "decoder.mid_block.resnets.0")
        x = self._attn(x, "decoder.mid_block.attentions.0")
        x = self._resnet(x, "decoder.mid_block.resnets.1")
        n = len(self.cfg["block_out_channels"])
        for i in range(n):
            for j in range(self.cfg["layers_per_block"] + 1):
                x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{j}")
            if i != n - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self._conv(x, f"decoder.up_blocks.{i}.upsamplers.0.conv")
        x = F.silu(self._gn(x, "decoder.conv_norm_out"))
        return self._conv(x, "decoder.conv_out")


def color_guidance_update(vae, latents, noise_pred, alpha_t, scaling, color_obj_atten, target_rgb, weight, color_obj_atten_all):
    """One guidance update exactly as rd.py:151-168 / xl.py:849-867 (torch autograd through the oracle decoder).
    color_obj_atten: list of [1,4,H,W] image-resolution masks; target_rgb: list of [1,3,1,1]."""
    lat = latents.detach().clone().requires_grad_(True)
    a = torch.as_tensor(alpha_t, dtype=torch.float32)
    with torch.enable_grad():
        x0 = (lat - noise_pred * torch.sqrt(1 - a)) / torch.sqrt(a)
        imgs = (vae.decode(x0 / scaling) / 2 + 0.5).clamp(0, 1)
        loss_total = 0.0
        for m, rgb in zip(color_obj_atten, target_rgb):
            avg = (imgs * m[:, 0]).sum(2).sum(2) / m[:, 0].sum()
            loss_total = loss_total + F.mse_loss(avg, rgb[:, :, 0, 0]) * 100
        loss_total.backward()
    return (lat - lat.grad * weight * color_obj_atten_all).detach().clone(), lat.grad.detach().clone(), float(loss_total)
