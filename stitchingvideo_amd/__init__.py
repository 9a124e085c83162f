"""stitchingvideo_amd -- MI355X-native per-frame panorama compositing engine.

The product is `libmi355stitch.so` (hand-written HIP kernels for gfx950 behind the C ABI of
include/mi355_stitch.h); this package is the thin host-side mirror of the reference's
warper / exposure-compensator / frame-loop interfaces plus the synthetic-rig input producers.
"""
from . import _capi as capi  # noqa: F401
from .engine import (PinnedArray, VoronoiSeamFinder, DpSeamFinder, BlocksGainCompensator, CylindricalWarper, GainCompensator, PanoramaEngine,  # noqa: F401
                     PlaneWarper, RotationWarper, SphericalWarper, compose_lens_maps, cvt_nv12_bgr, init_undistort_rectify_map, load_calibration, save_calibration, multiband_geometry,
                     remap_fixed, synth_frame)

__all__ = ["capi", "PanoramaEngine", "RotationWarper", "PlaneWarper", "SphericalWarper", "CylindricalWarper",
           "GainCompensator", "BlocksGainCompensator", "multiband_geometry", "synth_frame",
           "init_undistort_rectify_map", "compose_lens_maps", "cvt_nv12_bgr", "remap_fixed", "save_calibration", "load_calibration"]
