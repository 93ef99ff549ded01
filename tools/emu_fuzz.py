"""the differential fuzzer's cases (tests/checkers/fuzz_gpu.py) through the kernel sources under the CPU emulator: python tools/emu_fuzz.py <seconds> <first seed>  (make -C tests/hipemu first)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'checkers'))
import fuzz_gpu, oracle_py
from lariat_amd import capi
emu = capi.Library(os.path.join(ROOT, 'tests', '_build', 'liblariat_emu.so'))
oracle = oracle_py.load()
t_end = time.time() + float(sys.argv[1]); seed = int(sys.argv[2]); n = 0
while time.time() < t_end:
    try:
        fuzz_gpu.run_case(emu, oracle, seed)
    except AssertionError as e:
        print("DIFF", str(e)[:500], flush=True); sys.exit(1)
    seed += 1; n += 1
print("emu fuzz ok: %d cases up to seed %d" % (n, seed), flush=True)
