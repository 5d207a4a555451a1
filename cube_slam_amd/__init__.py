"""cube_slam_amd -- MI355X-native kernels for CubeSLAM's per-frame hot path.

The compute lives in libcubeslam_hip.so (hand-written HIP for gfx950, C-ABI in include/cubeslam_hip.h); this
package is the thin host-side mirror of the reference interfaces used by tests and bench.py.
"""
from . import _lib  # noqa: F401
