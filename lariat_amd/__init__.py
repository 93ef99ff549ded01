"""lariat_amd — MI355X-native replacement for the per-barcode align loop of 10XGenomics/lariat.

The product is ``lariat_amd/_build/liblariat_hip.so`` (hand-written HIP for gfx950 behind the C-ABI in
``include/lariat_hip.h``).  The Python here is plumbing: ctypes bindings (capi), the host-side mirror of
the reference's Go interface (gobwa, inference) used by tests and bench.py, and the synthetic-data generator.
"""
__version__ = "0.1.0"
