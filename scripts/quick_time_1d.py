"""Quick device-side timing of the 1-D kernels (host-buffer entry points, so H2D included)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreover_amd import batch, _lib
from poreover_amd.synth import synth_pair
_lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = [synth_pair(i, T=4000)[0] for i in range(32)]
reads = [base[i % 32] for i in range(n)]
for name, fn in (("viterbi", lambda: batch.viterbi_batch(reads)),
                 ("beam1d W=10", lambda: batch.beam_search_batch(reads, 10)),
                 ("beam1d W=25", lambda: batch.beam_search_batch(reads, 25))):
    fn()
    t0 = time.time(); out = fn(); dt = time.time() - t0
    bases = sum(len(s) for s in out)
    print("%-12s n=%d  %.3f s  %.1f reads/s  %.3f Mbases/s (host-buffer call, H2D included)" % (name, n, dt, n / dt, bases / dt / 1e6))
