"""libbsc_amd — MI355X-native block-sorting hot path behind the libbsc C API.

Python here is plumbing (ctypes over the C ABI in include/, torch for device memory and
torch.distributed); the product is libbsc_amd/lib/libbsc_mi355x.so.
"""
from .gpu import GpuContext, GpuError  # noqa: F401

__all__ = ["GpuContext", "GpuError"]
