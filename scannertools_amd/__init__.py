"""scannertools hot path (Histogram -> ShotBoundaries, Farneback OpticalFlow) for MI355X.

``scannertools_amd.hip``            device ops over the C ABI (include/scannertools_hip.h)
``scannertools_amd.shot_detection`` the ShotBoundaries python op
``scannertools_amd.types``          wire-format readers
``scannertools_amd.engine``         in-process stand-in for the Scanner graph API of the path
``scannertools_amd.imgproc``        loads the op library (mirror of scannertools.imgproc)
``scannertools_amd.optical_flow``   compute_flow (mirror of scannertools/old/optical_flow.py)
``scannertools_amd.histograms``     compute_histograms / _hsv_ / _flow_ (mirror of old/histograms.py)
``scannertools_amd.vis``            the DrawFlow python op (mirror of scannertools/vis.py)
"""
from .shot_detection import shot_boundaries, WINDOW_SIZE, BOUNDARY_BATCH  # noqa: F401

__all__ = ["shot_boundaries", "WINDOW_SIZE", "BOUNDARY_BATCH"]
