"""poreover_amd — MI355X (gfx950) CTC / pair-consensus decoding engine.

Drop-in for PoreOver's native decode path (poreover/decoding + poreover/align): the Python
modules under this package mirror the reference's names and signatures and call hand-written
HIP kernels through the C-ABI in include/poreover_hip.h.  No CPU fallback exists.
"""
__version__ = "0.1.0"
