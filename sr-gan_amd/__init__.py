"""sr-gan on MI355X: the SRGAN / SGAN training step (reference srgan.py / sgan.py) on hand-written HIP
kernels for gfx950 behind a C ABI (include/srgan_hip.h), orchestrated from Python on PyTorch-ROCm.

Importable as ``srgan_amd`` through the shim at the repository root."""
__version__ = '0.1.0'
