"""Ad-hoc probe (GPU): grid-method mismatches against the oracle by beam width / batch position."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiprocessing as mp
from oracle import po_oracle as O
from poreover_amd.synth import synth_pair

def one(a):
    y1, y2, env, W = a
    try:
        return O.cpp_beam_search_2d(y1, y2, env, W, model_="ctc", method_="grid")
    except Exception as e:
        return "ERR " + str(e)

def main():
    pool = mp.get_context("fork").Pool(16)
    from poreover_amd import batch
    rng = np.random.default_rng(11)
    cases = []
    for i in range(48):
        T = int(rng.integers(60, 260))
        seed = int(rng.integers(1 << 30))
        y1, y2 = synth_pair(seed, T=T)
        U, V = len(y1), len(y2)
        hw = int(rng.integers(3, 30))
        if i % 4 == 3:
            y2 = y2[: max(2, (2 * len(y2)) // 3)]; V = len(y2)
        if i % 2 == 0:
            env = np.asarray(O.diagonal_envelope(U, V, hw))
        else:
            try:
                env = np.asarray(O.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"])
            except Exception:
                env = np.asarray(O.diagonal_envelope(U, V, 12))
            if env is None or len(env) != U:
                env = np.asarray(O.diagonal_envelope(U, V, 12))
        cases.append((seed, T, hw, y1, y2, env))
    for W in (5, 6, 7, 9, 10, 12, 13, 16, 25):
        want = pool.map(one, [(c[3], c[4], c[5], W) for c in cases])
        got, st = batch.beam_search_2d_batch([c[3] for c in cases], [c[4] for c in cases], [c[5] for c in cases], W, method="grid", return_status=True)
        bad = [i for i in range(len(cases)) if st[i] == 0 and got[i] != want[i]]
        print("   refused", [i for i in range(len(cases)) if st[i] != 0])
        print("W", W, "batch mismatches", bad, flush=True)
        for i in bad[:3]:
            c = cases[i]
            alone = batch.beam_search_2d_batch([c[3]], [c[4]], [c[5]], W, method="grid", return_status=True)[0][0]
            np.savez("gpurun_out/grid_case_W%d_%d.npz" % (W, i), y1=c[3], y2=c[4], env=c[5], W=W, want=want[i], got=got[i])
            print("   case", i, "seed", c[0], "T", c[1], "hw", c[2], "U,V", len(c[3]), len(c[4]), "alone ok:", alone == want[i],
                  "batch==alone:", alone == got[i], "| want", want[i][:30], "| got", got[i][:30], flush=True)
    pool.terminate()

if __name__ == "__main__":
    main()
