"""circkit_amd -- MI355X (gfx950) drop-in for circkit's canonicalize / uniq hot path.

csrc/      hand-written HIP kernels + the C ABI (include/circkit.h) + the C++ FASTA host
api.py     ctypes mirror of the reference's lib-crate API over that C ABI
"""
from .api import (CirckitError, Context, canonicalize, default_context, lmsr, lmsr_index, load_library, normalize,  # noqa: F401
                  xxh3_64)
from . import uniq  # noqa: F401,E402
