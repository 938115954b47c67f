"""One end-to-end job (host float32 logits -> strings) for a timeline trace:
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 scripts/e2e_trace.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreover_amd import batch
from poreover_amd.synth import synth_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
base = [synth_pair(i, T=4000) for i in range(64)]
l1 = [base[i % 64][0].astype(np.float32) for i in range(n)]
l2 = [base[i % 64][1].astype(np.float32) for i in range(n)]
batch.pair_decode_stream(l1[:1500], l2[:1500], "poreover", 5, "row_col")
for rep in range(3):
    st = {}
    t0 = time.perf_counter()
    res = batch.pair_decode_stream(l1, l2, "poreover", 5, "row_col", stats=st)
    dt = time.perf_counter() - t0
    print("rep %d: %.4f s, %.0f pairs/s, %s" % (rep, dt, n / dt, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
    print("MARK %d %.6f %.6f" % (rep, t0, t0 + dt), flush=True)
