"""flashfry_amd -- MI355X-native implementation of FlashFry's `discover` off-target scan and CFD / Hsu2013
aggregation.  The compute path lives in csrc/ (hand-written HIP for gfx950) behind the C ABI declared in
include/flashfry_hip.h; this package only holds the thin Python plumbing around it (ctypes binding, synthetic
data, multi-GPU sharding).  There is no CPU fallback: importing `flashfry_amd.capi` fails loudly when the HIP
library has not been built."""

__version__ = "0.1.0"
