#!/usr/bin/env python3
"""Generate tests/golden/trace_golden.json: what the reference's `decode` produces for its two flip-flop trace
files (data/flappie_trace.hdf5, data/guppy_flipflop.fast5 — copied next to this script as DATA fixtures).

The reference opens them with h5py, which this image lacks; the generator therefore hands the reference a shim
module named h5py whose File is poreover_amd.decoding.hdf5_lite.File, and everything else — model_from_trace, the
trace scaling, transducer.flipflop.viterbi_decode, cpp_beam_search — is the reference's own code.  The reader itself
is checked independently: the Guppy file carries Guppy's own basecall (Fastq), which the Viterbi decode of the Trace
dataset must reproduce almost base for base.

    python3 tests/golden/make_golden_trace.py
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

sys.path.insert(0, MG.REPO)


def identity(a, b):
    """matches / columns of a plain global alignment (unit costs)"""
    n, m = len(a), len(b)
    prev = np.arange(m + 1)
    for i in range(1, n + 1):
        cur = np.empty(m + 1, dtype=np.int64)
        cur[0] = i
        sub = prev[:-1] + (np.frombuffer(b.encode(), dtype=np.uint8) != ord(a[i - 1]))
        cur[1:] = np.minimum(sub, prev[1:] + 1)
        for j in range(1, m + 1):
            if cur[j - 1] + 1 < cur[j]:
                cur[j] = cur[j - 1] + 1
        prev = cur
    return 1.0 - prev[m] / max(n, m)


def main():
    MG.build_reference()
    from poreover_amd.decoding import hdf5_lite
    shim = types.ModuleType("h5py")
    shim.File = hdf5_lite.File
    sys.modules["h5py"] = shim
    decoding, align = MG.import_reference()
    from poreover.decoding import decode as ref_decode
    ref_decode.h5py = shim
    G = {}
    for name, path, basecaller in (("flappie", "flappie_trace.hdf5", "flappie"), ("guppy", "guppy_flipflop.fast5", "guppy")):
        model = ref_decode.model_from_trace(os.path.join(HERE, path), basecaller)
        assert model.kind == "flipflop"
        y = np.array(model.log_prob)
        vit, vpath = model.viterbi_decode(return_path=True)
        rec = {"file": path, "basecaller": basecaller, "shape": list(y.shape),
               "log_prob_sha256": hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest(),
               "viterbi": vit, "path_sha256": hashlib.sha256(np.asarray(vpath, dtype=np.int8).tobytes()).hexdigest()}
        seg = y[8000:12000]
        rec["segment"] = [8000, 12000]
        rec["segment_beam_w10"] = decoding.cpp_beam_search(seg, beam_width_=10, model_="ctc_flipflop")
        rec["segment_beam_w5"] = decoding.cpp_beam_search(seg, beam_width_=5, model_="ctc_flipflop")
        rec["beam_w5"] = decoding.cpp_beam_search(y, beam_width_=5, model_="ctc_flipflop")
        G[name] = rec
        print(name, y.shape, len(vit), len(rec["beam_w5"]))
    # independent check of the container reader: Guppy's own basecall sits in the same file
    f = hdf5_lite.File(os.path.join(HERE, "guppy_flipflop.fast5"))
    fq = bytes(np.array(f["/Analyses/Basecall_1D_000/BaseCalled_template/Fastq"]).tobytes()).decode().split("\n")[1]
    idn = identity(G["guppy"]["viterbi"][:3000], fq[:3000])
    print("guppy Fastq length", len(fq), "viterbi length", len(G["guppy"]["viterbi"]), "identity of the first 3000 bases", idn)
    assert idn > 0.9
    G["guppy"]["fastq_len"] = len(fq)
    G["guppy"]["fastq_identity_first3000"] = idn
    with open(os.path.join(HERE, "trace_golden.json"), "w") as fo:
        json.dump(G, fo, indent=0, sort_keys=True)
    shutil.rmtree(MG.SCRATCH, ignore_errors=True)


if __name__ == "__main__":
    main()
