import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreover_amd import batch as eng, _lib
from poreover_amd.synth import synth_pair
from oracle import po_oracle as oracle
_lib.load()
def q(y): return np.log((np.clip(np.rint(np.exp(y) * 255), 0, 255) + 1e-7) / (255 + 1e-7))
rng = np.random.default_rng(5)
Ts = [int(rng.integers(40, 200)) for _ in range(128)]
y1, y2 = synth_pair(20127, T=Ts[127]); y1, y2 = q(y1), q(y2)
U, V = len(y1), len(y2)
env = np.array([(max(0, int(u * V / U) - 8), min(V, int(u * V / U) + 9)) for u in range(U)])
want = oracle.cpp_beam_search_2d(y1, y2, env, 8, model_="ctc_merge_repeats", method_="row")
# other work that dirties the workspace in between
o1, o2 = synth_pair(7100, T=150); oenv = oracle.pair_decode(o1, o2, "poreover", 5, "row_col")["envelope"]
bad = 0; N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for it in range(N):
    if it % 3 == 1: eng.beam_search_2d_batch([o1] * 6, [o2] * 6, [oenv] * 6, 5)
    if it % 3 == 2: eng.beam_search_2d_batch([o1] * 3, [o2] * 3, [oenv] * 3, 9, model="ctc_merge_repeats")
    got = eng.beam_search_2d_batch([y1], [y2], [env], 8, model="ctc_merge_repeats", method="row")
    if got != [want]:
        bad += 1
        if bad <= 3: print("iter", it, "MISMATCH", got[0][:60], want[:60])
print("lib", os.environ.get("POREOVER_HIP_LIB", "default"), "mismatches", bad, "of", N)
