import torch.nn as nn
def get_activation(act_fn):
    if act_fn in ("swish", "silu"):
        return nn.SiLU()
    if act_fn == "mish":
        return nn.Mish()
    if act_fn == "gelu":
        return nn.GELU()
    raise ValueError(act_fn)
