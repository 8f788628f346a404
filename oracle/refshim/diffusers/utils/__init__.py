import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields, is_dataclass
import torch

class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)
logging = _Logging()

def maybe_allow_in_graph(cls):
    return cls

def deprecate(*args, **kwargs):
    return None

def is_torch_version(op, ver):
    return True

def is_accelerate_available():
    return False

def is_accelerate_version(*a, **k):
    return False

def replace_example_docstring(doc):
    def deco(fn):
        return fn
    return deco

def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(shape, generator=generator, dtype=dtype).to(device)

class BaseOutput(OrderedDict):
    def __post_init__(self):
        if is_dataclass(self):
            for f in fields(self):
                v = getattr(self, f.name)
                if v is not None:
                    self[f.name] = v
    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return tuple(self.values())[k]
    def to_tuple(self):
        return tuple(self.values())
