"""floria_amd — MI355X-native per-block read->haplotype clustering (the hot path of bluenote-1577/floria).

The product is libfloria_hip.so (floria_amd/csrc, C ABI in include/floria_hip.h); this package is the
thin Python host mirror used by tests and bench.py.  There is no CPU fallback.
"""
from .pileup import Pileup, reads_in_interval  # noqa: F401

__version__ = "0.1.0"
