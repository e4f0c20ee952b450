"""bulletproofspp_amd — MI355X-native hot path of Bulletproofs++ (Liam-Eagen/BulletproofsPP).

The product is the C-ABI shared library `lib/libbppp_hip.so` (hand-written HIP for gfx950,
declared in include/bppp.h).  This package is the thin Python host binding over it used by the
tests and bench; there is no CPU fallback: importing `capi` without the built library, or using
it without a GPU, fails loudly.
"""
from .capi import Bppp, BpppError, lib_path  # noqa: F401

__all__ = ["Bppp", "BpppError", "lib_path"]
