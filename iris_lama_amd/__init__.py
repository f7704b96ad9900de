"""iris_lama_amd -- MI355X-native particle-filter scan-matching path behind the LaMa (iris_lama) API.

The product is two in-tree shared libraries:
  lib/liblama_hip.so   hand-written HIP kernels for gfx950 behind the C-ABI of include/lama_hip.h
  lib/liblama_hip_wide.so   the same sources built for distance maps that reach 128 .. 255 cells (a 4-byte distance plane)
  lib/liblama_host.so  C++ host-side mirror of the reference interface (lama::PFSlam2D ...) + workload generator
This package is only the thin ctypes plumbing over them (plus the torch.distributed orchestration for
multi-GPU sharding).  There is no CPU fallback: loading fails loudly if a library is missing.
"""
from . import ffi  # noqa: F401
