"""Import shim: the product package lives in the directory `rich-text-to-image_amd/` (the name the
project brief fixes), which is not a valid Python identifier.  `import rich_text_to_image_amd`
resolves here and re-exports that directory as this package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rich-text-to-image_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
