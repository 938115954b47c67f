"""Quick timing of the 1-D kernels through the host-buffer entry points; kernel time from the engine's HIP events."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from poreover_amd import batch, _lib
from poreover_amd.synth import synth_pair
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ff = len(sys.argv) > 2 and sys.argv[2] == "flipflop"
base = [synth_pair(i, T=4000, flipflop=ff)[0] for i in range(32)]
reads = [base[i % 32] for i in range(n)]
model = "ctc_flipflop" if ff else "ctc"
runs = [("viterbi", _lib.K_VITERBI, lambda: batch.viterbi_batch(reads, "flipflop" if ff else "poreover")),
        ("beam1d W=10", _lib.K_BEAM1D, lambda: batch.beam_search_batch(reads, 10, model=model)),
        ("beam1d W=25", _lib.K_BEAM1D, lambda: batch.beam_search_batch(reads, 25, model=model))]
for name, kid, fn in runs:
    fn()
    lib.po_profile_enable(1); lib.po_profile_reset()
    t0 = time.time(); out = fn(); dt = time.time() - t0
    ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(kid, C.byref(ms), C.byref(cnt))
    bases = sum(len(s) for s in out)
    print("%-12s %s n=%d  call %.3f s; kernel %.2f ms -> %.0f reads/s kernel-only, %.3f Mbases/s" % (
        name, model, n, dt, ms.value, n / (ms.value / 1e3), bases / (ms.value / 1e3) / 1e6))
