"""Import shim: the package directory is `pytorch-gan_amd/` (not a valid identifier), so this module
loads it and registers it as `pytorch_gan_amd`.  `import pytorch_gan_amd` from the repo root just works."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pytorch-gan_amd")
_spec = importlib.util.spec_from_file_location("pytorch_gan_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pytorch_gan_amd"] = _mod
_spec.loader.exec_module(_mod)
