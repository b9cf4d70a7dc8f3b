"""TEST INFRASTRUCTURE: host-side execution model for the HIP kernels (see include/hip/hip_runtime.h and build_emu.py).

    from hipemu import load
    emu = load()            # ctypes library with the C ABI of include/migan.h, kernels running as fibers on the host
    emu.migan_conv2d_fwd_ws(x.data_ptr(), ...)   # pointers are HOST pointers (torch CPU tensors / numpy arrays)

Only tests import this package; the product (pytorch-gan_amd/) never does, and has no CPU path of its own."""
import ctypes
import os


_LIB = None


def load(only=None):
    """Build (if stale) and load the emulation library; argument types come from the product's own signature table, so a
    drift between include/migan.h, _lib.py and the sources shows up here exactly as it does for the gfx950 library."""
    global _LIB
    if _LIB is not None and only is None:
        return _LIB
    from . import build_emu

    # persistent kernels run one OS thread per workgroup in the model: keep their grids small.  Set HERE (the emulation library
    # reads it once, at its first persistent launch) and not at import: pytest imports this package while COLLECTING on the GPU
    # box too, where the real library must keep its default grid.
    path = build_emu.build(only=only)
    lib = ctypes.CDLL(path)
    from pytorch_gan_amd import _lib as product

    for name, (res, args) in product._SIGS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if only is None:
                raise AttributeError("emulation library lacks %s" % name)
            continue
        fn.restype = res
        fn.argtypes = args
    lib.hipemu_last_message.restype = ctypes.c_char_p
    lib.hipemu_set_coresident.argtypes = [ctypes.c_int]
    lib.hipemu_set_threads.argtypes = [ctypes.c_int]
    lib.hipemu_launch_count.argtypes = [ctypes.c_char_p]
    lib.hipemu_add_coresident_kernel.argtypes = [ctypes.c_char_p]
    lib.hipemu_launch_count.restype = ctypes.c_long
    lib.hipemu_set_wave_schedule.argtypes = [ctypes.c_int, ctypes.c_uint]
    # HIPEMU_SCHED = fwd | rev | rand[:seed]: the order in which a workgroup's waves run between synchronisation points
    sched = os.environ.get("HIPEMU_SCHED", "fwd").split(":")
    lib.hipemu_set_wave_schedule({"fwd": 0, "rev": 1, "rand": 2}[sched[0]], int(sched[1]) if len(sched) > 1 else 1)
    lib.hipemu_count_grids.argtypes = [ctypes.c_int]
    if only is None:
        _LIB = lib
    return lib


def available():
    from . import build_emu

    return os.path.exists(build_emu.CXX) and os.uname().machine == "x86_64"
