"""MI355X-native SRUKF predict/update hot path of CV-MonoSLAM (gfx950 HIP kernels behind a C-ABI).

The directory name carries a hyphen (it mirrors the reference's name), so import it through
``__graft_entry__.load_package()`` which registers it as ``cv_monoslam_amd``.

Contents
  csrc/          hand-written HIP kernels + the C-ABI of include/srukf.h  -> libsrukf_hip.so
  host/          CSLAM-shaped C++ facade over the C-ABI (the drop-in for the MFC host)
  srukf.py       ctypes binding of the C-ABI (plumbing for tests / bench)
  synth.py       seeded synthetic scene generator (inputs only)
"""
from . import synth  # noqa: F401
from . import srukf  # noqa: F401

__all__ = ["synth", "srukf"]
