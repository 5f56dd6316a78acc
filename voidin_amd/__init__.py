"""voidin_amd — MI355X (gfx950) implementation of voidin's GPU-driven visibility path
(frustum cull -> emit_draws -> ordered compaction; SAH BLAS build; TLAS build/refit; TLAS/BLAS
traversal) behind the C ABI of include/voidin_abi.h.

The compute lives in voidin_amd/csrc/libvoidin_hip.so (hand-written HIP for CDNA4).  This
package is the thin host-side mirror used by tests and bench; it has no CPU fallback.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
__version__ = "0.1.0"
