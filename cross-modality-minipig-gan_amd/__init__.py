"""mpgan_amd -- MI355X-native GAN training hot path (T1->T2 CasNet generator +
conv discriminator) behind the reference's module API.

Python here is host plumbing only: every FLOP of the hot path runs in
hand-written HIP kernels (libmpgan_hip.so, built by `__graft_entry__.build()`).
Importing the compute modules without that library raises -- there is no CPU
or eager-PyTorch fallback.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the .so until first use)


def load_library():
    """Load libmpgan_hip.so now (raises RuntimeError if it is missing)."""
    return _lib.lib()
