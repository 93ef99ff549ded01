// lariat_hip.hip — the product: liblariat_hip.so for gfx950 (MI355X).  All kernels are hand-written HIP in the k_*.h headers;
// lh_host.inc sequences them behind the C-ABI of include/lariat_hip.h.
#include "lh_host.inc"
