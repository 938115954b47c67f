"""GPU parity, 1-D path: HIP engine (through the C-ABI) vs the CPU oracle and the golden vectors.
Bar: bit-exact paths / strings for argmax-Viterbi; identical strings for beam search on these
inputs (the documented tolerance for beam search is <= 0.1 % edit distance)."""
import numpy as np
import pytest

from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu

MODEL_OF_KIND = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


def _reads(n, T, ff=False, base=1000):
    out = []
    for i in range(n):
        a, b = synth_pair(base + i, T=T, flipflop=ff)
        out += [a, b]
    return out


@pytest.mark.parametrize("kind", ["poreover", "bonito", "flipflop"])
def test_viterbi_matches_oracle(eng, oracle, kind):
    reads = _reads(6, 700, ff=(kind == "flipflop")) + [_reads(1, 3000, ff=(kind == "flipflop"))[0]]
    reads.append(reads[0][:1])      # T = 1
    reads.append(reads[1][:257])    # one frame past a tile boundary
    seqs, paths, maps, st = eng.viterbi_batch(reads, kind, return_path=True, return_map=True)
    for i, y in enumerate(reads):
        s, p = oracle.viterbi_decode(y, kind)
        assert seqs[i] == s, (kind, i)
        assert np.array_equal(paths[i], p), (kind, i)
        m = oracle.get_sequence_mapping(p, kind)
        if len(m) == len(s):
            assert st[i] == 0 and np.array_equal(maps[i], m), (kind, i)
        else:  # bonito wrap-around quirk: the reference's assert fires (pair_decode.py:379)
            assert st[i] != 0


def test_flipflop_viterbi_block_boundaries(eng, oracle):
    """flipflop_dp_kernel requests its rows 16 frames ahead and stores a block's back-pointers a block late (round 6): read
    lengths around the block size and the back-trace's 512-frame chunks, ragged within a wave (8 reads per wave share the loop)"""
    base = _reads(3, 1100, ff=True, base=7300)
    lens = [1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49, 511, 512, 513, 527, 528, 529, 1023, 1024, 1025, 1040, 1041]
    reads = [base[i % 3][:L] for i, L in enumerate(lens)]
    seqs, paths, maps, st = eng.viterbi_batch(reads, "flipflop", return_path=True, return_map=True)
    for i, y in enumerate(reads):
        s, p = oracle.viterbi_decode(y, "flipflop")
        assert seqs[i] == s, (lens[i], seqs[i][:20], s[:20])
        assert np.array_equal(paths[i], p), lens[i]
        assert st[i] == 0 and np.array_equal(maps[i], oracle.get_sequence_mapping(p, "flipflop")), lens[i]


def test_viterbi_golden(eng, golden, golden_inputs):
    y = np.log(golden_inputs["poreover_csv_prob"])
    seqs, paths = eng.viterbi_batch([y], "poreover", return_path=True)
    assert seqs[0] == golden["csv"]["viterbi"]
    assert paths[0].tolist() == golden["csv"]["viterbi_path"]
    for rec in golden["pairs"]:
        y1 = golden_inputs["pair%d_y1" % rec["index"]]
        y2 = golden_inputs["pair%d_y2" % rec["index"]]
        seqs, paths, maps, st = eng.viterbi_batch([y1, y2], rec["kind"], return_path=True, return_map=True)
        assert seqs == [rec["viterbi1"], rec["viterbi2"]]
        assert paths[0].tolist() == rec["path1"] and paths[1].tolist() == rec["path2"]
        assert maps[0].tolist() == rec["map1"] and maps[1].tolist() == rec["map2"]


def test_viterbi_toy_alphabet(eng, golden):
    t1 = np.array(golden["toy_prob"]["t1"])
    assert eng.viterbi_batch([t1], "poreover", alphabet="AB")[0] == golden["toy"]["viterbi_t1"]


@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
@pytest.mark.parametrize("W", [1, 3, 5, 10, 25])
def test_beam1d_matches_oracle(eng, oracle, model, ff, W):
    reads = _reads(4, 400, ff=ff, base=2000) + [_reads(1, 150, ff=ff, base=2100)[0][:1]]
    got = eng.beam_search_batch(reads, W, model=model)
    for i, y in enumerate(reads):
        assert got[i] == oracle.cpp_beam_search(y, W, model_=model), (model, W, i)


def test_beam1d_golden(eng, golden, golden_inputs):
    from poreover_amd.decoding import cpp_beam_search
    toy, g = golden["toy_prob"], golden["toy"]
    assert cpp_beam_search(np.log(np.array(toy["t1"])), alphabet_="AB") == g["beam1d_t1"]
    assert cpp_beam_search(np.log(np.array(toy["t2"])), alphabet_="AB") == g["beam1d_t2"]
    ff = np.log(np.array(toy["ff"], dtype=np.float32))
    assert cpp_beam_search(ff, alphabet_="AB", model_="ctc_flipflop") == g["ff_beam1d"]
    y = np.log(golden_inputs["poreover_csv_prob"])
    for W in (5, 10, 25):
        assert cpp_beam_search(y, beam_width_=W) == golden["csv"]["beam_w%d" % W]
    assert cpp_beam_search(y, beam_width_=10, model_="ctc_merge_repeats") == golden["csv"]["beam_merge_w10"]
    for rec in golden["pairs"]:
        y1 = golden_inputs["pair%d_y1" % rec["index"]]
        for W in (5, 10):
            assert cpp_beam_search(y1, W, model_=MODEL_OF_KIND[rec["kind"]]) == rec["beam1d_w%d" % W]


def test_beam1d_full_size_batch(eng, oracle):
    """BASELINE config 2 shape (T = 4000, W = 10) on a batch; oracle on every read."""
    reads = [synth_pair(3000 + i, T=4000)[0] for i in range(24)]
    got = eng.beam_search_batch(reads, 10)
    want = [oracle.cpp_beam_search(y, 10) for y in reads]
    assert got == want


def test_beam1d_flipflop_full_size_batch(eng, oracle):
    """BASELINE config 5 shape: flip-flop (T x 8 state) traces, T = 4000, W = 10, 64 reads; oracle on every read,
    Viterbi and beam search."""
    reads = [synth_pair(3500 + i, T=4000, flipflop=True)[0] for i in range(64)]
    got = eng.beam_search_batch(reads, 10, model="ctc_flipflop")
    seqs = eng.viterbi_batch(reads, "flipflop")
    for i, y in enumerate(reads):
        assert got[i] == oracle.cpp_beam_search(y, 10, model_="ctc_flipflop"), i
        assert seqs[i] == oracle.viterbi_decode(y, "flipflop")[0], i


def test_beam1d_ragged_and_errors(eng):
    from poreover_amd import _lib
    reads = [synth_pair(10, T=t)[0] for t in (5, 64, 65, 300, 2)]
    got = eng.beam_search_batch(reads, 5)
    assert all(isinstance(s, str) for s in got) and len(got) == 5
    with pytest.raises((ValueError, _lib.EngineError)):
        eng.beam_search_batch([np.zeros((10, 7))], 5)      # C does not match the model
    with pytest.raises(_lib.EngineError):
        eng.beam_search_batch(reads[:1], 5, alphabet="ACGTN")


@pytest.mark.parametrize("model,kind", [("ctc", "poreover"), ("ctc_merge_repeats", "bonito"), ("ctc_flipflop", "flipflop")])
@pytest.mark.parametrize("W", [3, 5, 10, 12, 25])
def test_beam1d_last_frame_opens_a_y_block(oracle, model, kind, W):
    """reads whose LAST frame is the first of a 32-frame y block (T - 1 a multiple of 32) and their neighbours: the steady run of
    beam1d_wave_kernel must have the block in LDS before the general path ranks that frame (round 6: the open-ended 1-D fuzz found
    61 of 51 806 reads wrong on a loop that left the run before committing the block; the seeded tests had no such length)"""
    from poreover_amd import batch
    from poreover_amd.synth import synth_pair
    Ts = [32, 33, 34, 64, 65, 66, 97, 129, 1025, 1536, 1537, 1538, 2049]
    reads = [synth_pair(91000 + i, T=max(T, 40), flipflop=(kind == "flipflop"))[0][:T] for i, T in enumerate(Ts)]
    got = batch.beam_search_batch(reads, W, model=model)
    for T, y, g in zip(Ts, reads, got):
        assert g == oracle.cpp_beam_search(y, W, model_=model), (model, W, T)
