"""Loads the op library, like ``scannertools/imgproc/__init__.py`` of the reference
(`_register_module(<pkg dir>, "scannertools_imgproc")`, /root/reference/scannertools/scannertools/
imgproc/__init__.py:1-3).  Importing this module makes the ops Histogram, OpticalFlow,
FlowHistogram, Blur, Resize and ConvertColor available: inside a Scanner deployment through
``scannertools_infra._register_module`` (Scanner then dlopens libscannertools_imgproc.so, whose
static initialisers run REGISTER_OP / REGISTER_KERNEL); standalone through the in-process engine
(`scannertools_amd.engine`), which loads the same library."""
import os

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lib")
LIB_NAME = "scannertools_imgproc"

try:
    from scannertools_infra import _register_module
    # the library sits in <package>/lib rather than <package>/imgproc/build
    _register_module(os.path.join(LIB_DIR, "x"), LIB_NAME)
except ImportError:
    from .. import engine as _engine
    _engine._imgproc()  # raises if libscannertools_imgproc.so has not been built
