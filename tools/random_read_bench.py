"""practical ceiling for random 64-B / 128-B block reads on this GPU; prints GB/s of requested bytes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lariat_amd import capi
lib = capi.load_library()
for table_mb in (64, 200, 4096, 16384):
    for gran in (64, 128):
        g, ms = lib.diag_random_read(table_mb << 20, gran, 1 << 28 if gran == 64 else 1 << 27)
        print("table %6d MiB  granule %3d B : %8.1f GB/s requested  (%.2f ms)" % (table_mb, gran, g, ms), flush=True)
