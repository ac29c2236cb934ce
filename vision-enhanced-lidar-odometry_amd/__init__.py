"""MI355X-native scan-matching core for VELO's frame-to-frame registration loop.

The directory name carries a hyphen (it mirrors the upstream repository name), so the package is
imported through the root-level shim as ``velo_amd``:

    import velo_amd
    from velo_amd import api, synth

Contents: ``csrc/`` (HIP kernels + the C-ABI shared library, see include/velo_hip.h),
``api.py`` (ctypes mirror of that C-ABI -- plumbing for tests/bench, no compute),
``synth.py`` (seeded HDL-64E / KITTI-layout inputs), ``shard.py`` (multi-GPU host logic), ``odometry.py`` (pose
hand-off loop around the path), ``build.py`` (hipcc recipe).
"""
__all__ = ["api", "synth", "build", "shard", "odometry"]
