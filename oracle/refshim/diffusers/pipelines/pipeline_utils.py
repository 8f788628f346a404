import contextlib
from ..configuration_utils import ConfigMixin

class _Bar:
    def update(self, *a, **k): pass

class DiffusionPipeline(ConfigMixin):
    @contextlib.contextmanager
    def _bar(self):
        yield _Bar()
    def progress_bar(self, iterable=None, total=None):
        return self._bar()
    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
    @property
    def _execution_device(self):
        return self.device
