"""practical ceiling of independent random reads (lh_diag_random_read) at the sizes of the resident tables:
64 MiB (Infinity-Cache resident), 3 GiB (hg38 occurrence table), 48 GiB (hg38 dense suffix array / inverse array)"""
import os, sys
sys.path.insert(0, os.getcwd())
from lariat_amd import capi
lib = capi.load_library()
for table_mb in (64, 3072, 49152):
    for gran in (16, 32, 64):
        n = (1 << 33) // gran // 4
        g, ms = lib.diag_random_read(table_mb << 20, gran, n)
        print("table %6d MiB  granule %3d B : %8.1f GB/s requested  %.1f G acc/s (%.2f ms)" % (table_mb, gran, g, g / gran, ms), flush=True)
