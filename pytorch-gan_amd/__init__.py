"""pytorch_gan_amd — MI355X-native (gfx950) hot path for the PyTorch-GAN training loop.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd, torch.distributed); every layer
forward/backward, loss and optimiser update on the path runs in hand-written HIP kernels from
csrc/libmigan.so behind the C ABI in include/migan.h.  Importing the package loads that library and
raises if it is missing — there is no CPU / ATen fallback.

The directory is named `pytorch-gan_amd/`; import it as `pytorch_gan_amd` (the repo-root shim
`pytorch_gan_amd.py` registers it under that name).
"""
from . import _lib  # noqa: F401  (loads libmigan.so; raises when unavailable)
from . import functional, nn, optim  # noqa: F401
from .nn import swap, set_fusion, dropout_masks, manual_seed  # noqa: F401
from ._lib import version as lib_version  # noqa: F401

__all__ = ["functional", "nn", "optim", "swap", "set_fusion", "dropout_masks", "manual_seed", "lib_version"]
