"""verbatim-rag hot path, MI355X-native (gfx950).

Drop-in implementations of the reference's provider interfaces for ONE path:
ModernBERT span extraction + SPLADE / dense embedding + dot-product top-k
(SURVEY.md section 8).  Python here is host glue that mirrors the reference's
plug points; all device arithmetic is hand-written HIP behind the C ABI in
include/vrag_amd.h (libvrag_amd.so).  There is no CPU fallback: constructing
an engine without the library or without a GPU raises.
"""
from .version import __version__  # noqa: F401

__all__ = ["__version__"]
