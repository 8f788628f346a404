"""Timesteps / TimestepEmbedding restated from diffusers 0.18.2 models/embeddings.py
(published algorithm; the source is not on disk here => [memory], parity unpinned)."""
import math
import torch
from torch import nn

def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False,
                           downscale_freq_shift=1, scale=1, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32,
                                                     device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1, 0, 0))
    return emb

class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift
    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels,
                                      flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)

class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None,
                 post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.cond_proj = None
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)
        self.post_act = None
    def forward(self, sample, condition=None):
        sample = self.linear_1(sample)
        sample = self.act(sample)
        sample = self.linear_2(sample)
        return sample

class _P:
    def __init__(self, *a, **k):
        raise RuntimeError("import-only placeholder")
class GaussianFourierProjection(_P): pass
class ImageHintTimeEmbedding(_P): pass
class ImageProjection(_P): pass
class ImageTimeEmbedding(_P): pass
class TextImageProjection(_P): pass
class TextImageTimeEmbedding(_P): pass
class TextTimeEmbedding(_P): pass
class CombinedTimestepLabelEmbeddings(_P): pass
class ImagePositionalEmbeddings(_P): pass
class PatchEmbed(_P): pass
