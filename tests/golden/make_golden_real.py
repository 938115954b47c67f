#!/usr/bin/env python3
"""Generate tests/golden/real_{inputs.npz,golden.json} by RUNNING the Python/Cython reference on its own
sample signal (data/reads/read1.npy, read2.npy: float32 logits of shape (155, 400, 5) / (189, 400, 5), i.e.
62 000 / 75 600 frames — the pair of SURVEY.md §8(c) item 8).

Runs only in the build container (needs /root/reference); fails loudly on any error.  What is committed is
data: the two logit arrays (the reference's sample input, 2.7 MB of float32) and the outputs the reference
produced for them — digests of load_logits' float64 log-probabilities, 1-D Viterbi basecalls (forward and
reverse-complemented), 1-D beam search, the banded alignment, identity, the full 62 000-row envelope and the
pair-decode consensus for `--reverse_complement` with row_col / row at beam width 5.

    python3 tests/golden/make_golden_real.py
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (build_reference / import_reference: scratch build of the reference)

REF = MG.REF


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    MG.build_reference()
    decoding, align = MG.import_reference()
    from poreover.decoding import pair_decode as ref_pd, decode as ref_decode
    from poreover.decoding import envelope as ref_env

    p1 = os.path.join(REF, "data/reads/read1.npy")
    p2 = os.path.join(REF, "data/reads/read2.npy")
    l1, l2 = np.load(p1), np.load(p2)
    assert l1.dtype == np.float32 and l1.shape[1:] == (400, 5) and l2.shape[1:] == (400, 5)
    inputs = {"read1_logits": l1, "read2_logits": l2}
    G = {"source": "reference data/reads/read1.npy, read2.npy through the reference's own decode / pair_decode"}

    # ---- ingest: decode.load_logits(flatten=True) (decode.py:41-51)
    m1 = ref_decode.model_from_trace(p1, "poreover")
    m2 = ref_decode.model_from_trace(p2, "poreover")
    y1, y2 = np.array(m1.log_prob), np.array(m2.log_prob)
    G["log_prob_dtype"] = str(y1.dtype)
    G["log_prob_shape"] = [list(y1.shape), list(y2.shape)]
    G["log_prob_sha256"] = [sha(y1), sha(y2)]
    # a few rows in the clear so that a digest mismatch can be localised
    G["log_prob_rows"] = {"read1_0": [MG.jf(x) for x in y1[0]], "read1_last": [MG.jf(x) for x in y1[-1]],
                          "read2_0": [MG.jf(x) for x in y2[0]], "read2_31337": [MG.jf(x) for x in y2[31337]]}

    # ---- 1-D (transducer.py:72-73; decoding_cpp.pyx:88)
    s1, pth1 = m1.viterbi_decode(return_path=True)
    s2, pth2 = m2.viterbi_decode(return_path=True)
    G["viterbi1"], G["viterbi2_forward"] = s1, s2
    G["path_sha256"] = [sha(np.asarray(pth1, dtype=np.int8)), sha(np.asarray(pth2, dtype=np.int8))]
    t0 = time.time()
    G["beam1d_read1"] = {str(W): decoding.cpp_beam_search(y1, beam_width_=W) for W in (5, 10, 25)}
    G["beam1d_read1_seconds"] = round(time.time() - t0, 2)
    seg = slice(20000, 24000)   # one T = 4000 stretch of real signal for the other 1-D entry points
    G["segment"] = [seg.start, seg.stop]
    vseg = ref_decode.transducer.poreover(y1[seg]).viterbi_decode()
    G["segment_viterbi"] = vseg
    G["segment_forward_viterbi"] = MG.jf(decoding.cpp_forward(y1[seg], vseg))
    G["segment_beam_merge_w10"] = decoding.cpp_beam_search(y1[seg], beam_width_=10, model_="ctc_merge_repeats")

    # ---- pair decode with --reverse_complement (pair_decode.py:305-529)
    import argparse as _ap

    def ns(**kw):
        d = dict(dir=os.path.join(REF, "data/reads"), basecaller="poreover", reverse_complement=True, out="out",
                 threads=1, method="envelope", single="viterbi", logging="info", debug=False, algorithm="beam",
                 alignment="banded", beam_width=5, debug_envelope=False, diagonal_envelope=False,
                 diagonal_width=50, padding=5, skip_matches=False, skip_threshold=10,
                 beam_search_method="row_col", window=200)
        d.update(kw)
        n = _ap.Namespace(**d)
        setattr(n, "in", ["read1.npy", "read2.npy"])
        return n

    captured = {}
    real_build = ref_env.build_envelope

    def spy_build(*a, **k):
        e = real_build(*a, **k)
        captured["env"] = np.array(e)
        return e

    ref_env.build_envelope = spy_build
    real_b = align.global_pair_banded

    def spy_b(*a, **k):
        r = real_b(*a, **k)
        captured["aln"] = ["".join(r[0]), "".join(r[1])]
        return r

    ref_pd.align.global_pair_banded = spy_b

    runs = {}
    for method in ("row_col", "row"):
        captured.clear()
        t0 = time.time()
        out = ref_pd.pair_decode_helper(ns(beam_search_method=method))
        assert len(out) == 3, "the reference skipped its own sample pair"
        r = {"fasta_1d": out[0], "fasta_2d": out[1],
             "summary": {k: (MG.jf(v) if isinstance(v, float) else v) for k, v in out[2].items()},
             "alignment": captured["aln"], "seconds": round(time.time() - t0, 2)}
        env = captured["env"].astype(np.int32)
        assert env.shape == (len(y1), 2)
        if "envelope" in inputs:
            assert np.array_equal(inputs["envelope"], env)
        inputs["envelope"] = env
        cons = "".join(out[1].split("\n")[1:])
        r["consensus_len"] = len(cons)
        r["consensus_md5_12"] = hashlib.md5(cons.encode()).hexdigest()[:12]
        runs[method + "_w5"] = r
    G["pair_revcomp"] = runs
    # reverse-complemented read 2 as the pair path sees it (transducer.py:68-70)
    m2.reverse_complement()
    G["viterbi2_revcomp"] = m2.viterbi_decode()
    G["log_prob_revcomp_sha256"] = sha(np.ascontiguousarray(m2.log_prob))
    env = inputs["envelope"]
    G["envelope_cells"] = int((env[:, 1] - env[:, 0]).sum())
    G["envelope_max_band"] = int((env[:, 1] - env[:, 0]).max())
    G["survey_hashes"] = {"read1_viterbi": hashlib.md5(s1.encode()).hexdigest()[:12],
                          "read2_viterbi_forward": hashlib.md5(s2.encode()).hexdigest()[:12]}

    with open(os.path.join(HERE, "real_golden.json"), "w") as f:
        json.dump(G, f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "real_inputs.npz"), **inputs)
    print("wrote real_golden.json (%d bytes), real_inputs.npz (%d bytes)" % (
        os.path.getsize(os.path.join(HERE, "real_golden.json")), os.path.getsize(os.path.join(HERE, "real_inputs.npz"))))
    print({k: G[k] for k in ("survey_hashes", "envelope_cells", "envelope_max_band")},
          {k: (v["consensus_len"], v["consensus_md5_12"], v["seconds"]) for k, v in runs.items()})
    if not args.keep:
        shutil.rmtree(MG.SCRATCH, ignore_errors=True)


if __name__ == "__main__":
    main()
