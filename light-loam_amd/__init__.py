"""lightloam_amd -- MI355X-native hot path of Light-LOAM (host-side Python plumbing).

The product is the C-ABI shared library built from ``csrc/`` (see include/lightloam_hip.h);
this package only loads it with ctypes for tests, smoke and bench.  Importable as
``lightloam_amd`` through the shim ``/lightloam_amd.py`` at the repo root (the directory name
``light-loam_amd`` is not a valid Python identifier).
"""
from .build import build_all, lib_path, synth_lib_path  # noqa: F401
