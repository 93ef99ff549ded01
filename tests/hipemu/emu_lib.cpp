// emu_lib.cpp — TEST INFRASTRUCTURE ONLY.  Compiles the product's host pipeline and kernel SOURCES against the SPMD
// emulator (hip_emu.h) into tests/_build/liblariat_emu.so so that the CPU test-suite can diff kernel logic against the
// oracle without a GPU.  Never loaded by the lariat_amd package.
#define LH_EMU 1
#include "../../lariat_amd/csrc/lh_host.inc"
