"""lariat_amd — MI355X-native replacement for the per-barcode align loop of 10XGenomics/lariat.

The product is ``lariat_amd/_build/liblariat_hip.so`` (hand-written HIP for gfx950 behind the C-ABI in
``include/lariat_hip.h``).  The Python here is plumbing for tests and bench.py: ctypes bindings of that header (capi), barcode-range sharding
(shard), the synthetic genomes and reads of BASELINE.json's configs (synth, workload) and the -simulated accounting (simulated).
"""
__version__ = "0.1.0"
