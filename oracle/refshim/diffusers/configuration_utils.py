import functools, inspect

class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

class ConfigMixin:
    config_name = None
    def register_to_config(self, **kwargs):
        cfg = dict(getattr(self, "_internal_dict", {}))
        cfg.update(kwargs)
        self._internal_dict = FrozenDict(cfg)
    @property
    def config(self):
        return self._internal_dict

def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = list(sig.parameters.items())[1:]
        cfg = {n: p.default for n, p in params if p.default is not inspect.Parameter.empty}
        for (n, _), a in zip(params, args):
            cfg[n] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        self.register_to_config(**cfg)
        init(self, *args, **kwargs)
    return inner
