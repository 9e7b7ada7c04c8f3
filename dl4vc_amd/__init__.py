"""dl4vc_amd -- MI355X-native DAN inference forward for DL4VC (hot path only).

Host side is plain Python; the compute path is hand-written HIP for gfx950 behind the
C ABI declared in ``include/dl4vc_dan.h`` (``dl4vc_amd/csrc/libdl4vc_dan.so``).
"""
from .config import DanConfig  # noqa: F401

__version__ = "0.1.0"
