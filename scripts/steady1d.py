import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreover_amd import batch, _lib
from poreover_amd.synth import log_softmax
lib = _lib.load()
rng = np.random.default_rng(1)
T = 4000
def mk(kind):
    logits = rng.normal(0, 1.0, (T, 5)).astype(np.float32)
    if kind == "blank": logits[:, 4] += 12.0
    return log_softmax(logits)
for kind in ("blank",):
    ys = [mk(kind) for _ in range(16)] * 63
    for rep in range(3):
        lib.po_profile_enable(1); lib.po_profile_reset()
        out = batch.beam_search_batch(ys, 10, model="ctc") if hasattr(batch, "beam_search_batch") else None
        ms = C.c_double(); cnt = C.c_int64(); lib.po_profile_get(_lib.K_BEAM1D, C.byref(ms), C.byref(cnt))
        print(kind, len(ys), "reads: kernel %.2f ms -> %.3f us per frame" % (ms.value, ms.value * 1e3 / T), "len", len(out[0]) if out else None)
