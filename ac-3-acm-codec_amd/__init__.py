"""ac-3-acm-codec_amd — MI355X-native AC-3 block-transform engine (host-side Python mirror).

The product is ``libac3mi.so`` (hand-written HIP for gfx950 behind the C-ABI in
``include/ac3mi.h``).  This package only loads it through ctypes and mirrors the
reference's call surface for tests and bench.py; PyTorch is used for device
memory and nothing else.  There is no CPU fallback: importing works without a
GPU (so the symbol table can be checked), creating an engine does not.

The directory name contains hyphens, so import it with
``importlib.import_module("ac-3-acm-codec_amd")``.
"""
from .capi import LIB_PATH, load_library, declared_symbols, AC3MIError  # noqa: F401
from .engine import Engine, XformDesc, DecodeDesc, EncodeDesc, syncinfo  # noqa: F401
from . import flags  # noqa: F401
from . import sharding  # noqa: F401
