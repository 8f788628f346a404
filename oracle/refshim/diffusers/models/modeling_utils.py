import torch
class ModelMixin(torch.nn.Module):
    pass
