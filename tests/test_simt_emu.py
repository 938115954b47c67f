"""CPU: the pair-beam kernels' LOGIC without a GPU.  tools/simt_emu compiles the product's own kernel sources
(po_beam2d_pre.h, po_beam2d_reg.hip) with g++ against a lane-by-lane emulation of the HIP execution
model (one fibre per lane, cross-lane operations as rendezvous points) and runs them on small pairs; the strings must be the
oracle's.  Both lane schedules (ascending / descending) must agree: a result that changes with the schedule means a kernel
relies on lockstep execution across an LDS hand-over without a fence.  This is test infrastructure around the SAME sources
the GPU library is built from — the `-m gpu` suite stays the parity gate on real hardware."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(REPO, "tools", "simt_emu")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU])
    return os.path.join(EMU, "_build", "libemu_pair_beam.so")


@pytest.mark.parametrize("model,W", [("ctc", "0"), ("merge", "0"), ("flipflop", "5"), ("ctc", "10"), ("merge", "12")])
@pytest.mark.parametrize("sched", ["0", "1"])
def test_emulated_kernel_matches_oracle(emu_lib, oracle, model, W, sched):
    """every tree model, both lane layouts (W <= 6: lane = (read, slot); 7 <= W <= 12: lane = slot, reads in sequence)"""
    kernel = "reg"
    env = dict(os.environ, EMU_SCHED=sched)
    out = subprocess.run([sys.executable, os.path.join(EMU, "check_emu.py"), "--n", "18", "--T", "320", "--W", W, "--model", model, "--seed", "31",
                          "--procs", "4", "--kernel", kernel, "--styles", "pipeline,stairs,wobble", "--lib", emu_lib],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "0 mismatches" in out.stdout


@pytest.mark.parametrize("mode,W", [("1", "0"), ("1", "10"), ("2", "5")])
def test_emulated_closed_form_chains(emu_lib, oracle, mode, W):
    """po_set_chain_mode(PO_CHAIN_CLOSED_FORM): the new elements' windows in closed form (one exp, a prefix sum, one log per time;
    both lane layouts), and with its guard at 3 nats (mode 2: most steps hand over to the general scan).  The values differ from
    the serial chain's in the last bits; the strings of these small cases are the oracle's."""
    env = dict(os.environ, EMU_CHAIN_SCAN=mode)
    out = subprocess.run([sys.executable, os.path.join(EMU, "check_emu.py"), "--n", "16", "--T", "320", "--W", W, "--model", "ctc", "--seed", "77",
                          "--procs", "4", "--styles", "pipeline,stairs,wobble", "--lib", emu_lib],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "0 mismatches" in out.stdout


def test_emulated_kernel_hands_a_wide_window_on(emu_lib, oracle):
    """the fuzz case that faulted the GPU in round 4 (a 311-frame window at W = 1: the packed walk records keep a window
    length in 8 bits): the register-state kernel must DEFER the pair (beam2d_kernel decodes it), not decode it wrong"""
    import ctypes as C

    import numpy as np
    d = np.load(os.path.join(REPO, "tests", "golden", "fuzz_cases", "seed411_pipeline_W1_window311.npz"))
    y1 = np.ascontiguousarray(d["y1"], dtype=np.float64); y2 = np.ascontiguousarray(d["y2"], dtype=np.float64)
    env = np.ascontiguousarray(d["env"], dtype=np.int32)
    assert int((env[:, 1] - env[:, 0]).max()) < 255      # (rows are narrow: the wide window is a COLUMN's)
    lib = C.CDLL(emu_lib)
    o1 = np.array([0, len(y1)], dtype=np.int64); o2 = np.array([0, len(y2)], dtype=np.int64)
    cap = len(y1) + len(y2) + 8
    seq = np.zeros(cap, dtype=np.uint8); so = np.array([0, cap], dtype=np.int64)
    sl = np.zeros(1, dtype=np.int32); st = np.zeros(1, dtype=np.int32); upd = np.zeros(2, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    deferred = lib.emu_ring_pair_beam(p(y1), p(o1), p(y2), p(o2), p(env), 1, 5, 4, C.c_uint32(int.from_bytes(b"ACGT", "little")),
                                      int(d["W"]), p(seq), p(so), p(sl), p(st), 1, p(upd), 1, 0)
    assert deferred == 1 and int(st[0]) == -100


def _emu_one(lib_path, d, model_code):
    """one saved fuzz case (tests/golden/fuzz_cases/*.npz) through the emulated kernel: (string, status, deferred)"""
    import ctypes as C

    import numpy as np
    y1 = np.ascontiguousarray(d["y1"], dtype=np.float64); y2 = np.ascontiguousarray(d["y2"], dtype=np.float64)
    env = np.ascontiguousarray(d["env"], dtype=np.int32)
    lib = C.CDLL(lib_path)
    o1 = np.array([0, len(y1)], dtype=np.int64); o2 = np.array([0, len(y2)], dtype=np.int64)
    cap = len(y1) + len(y2) + 8
    seq = np.zeros(cap, dtype=np.uint8); so = np.array([0, cap], dtype=np.int64)
    sl = np.zeros(1, dtype=np.int32); st = np.zeros(1, dtype=np.int32); upd = np.zeros(2, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    deferred = lib.emu_ring_pair_beam(p(y1), p(o1), p(y2), p(o2), p(env), 1, y1.shape[1], 4, C.c_uint32(int.from_bytes(b"ACGT", "little")),
                                      int(d["W"]), p(seq), p(so), p(sl), p(st), 1, p(upd), 1, model_code)
    return bytes(seq[: sl[0]]).decode(), int(st[0]), deferred


def test_emulated_kernel_catchup_read_of_a_reentered_parent(emu_lib, oracle):
    """round 5's fuzz find (seed 701, a flip-flop pair at W = 5): during catch-up steps a beam node reads the stored values of
    its parent, which is an element AGAIN (a child slot) and has not computed since — what it stored before it left is there,
    up to its row's header.  With the tag-free entries that read was answered "absent" and one base of 291 came out wrong."""
    import numpy as np
    d = np.load(os.path.join(REPO, "tests", "golden", "fuzz_cases", "seed701_flipflop_W5_catchup_reads_reentered_parent.npz"))
    assert str(d["model"]) == "ctc_flipflop"
    want = oracle.cpp_beam_search_2d(d["y1"], d["y2"], d["env"], int(d["W"]), model_="ctc_flipflop", method_="row_col")
    assert want == str(d["want"])
    got, st, deferred = _emu_one(emu_lib, d, 2)
    assert st == 0 and deferred == 0
    assert got == want


def test_tag_free_store_answers_what_tags_would(oracle):
    """-DPO_EMU_SHADOW: the emulator keeps, beside the value store, the {pair, node, time} TAG rounds 1 - 4 kept in it, and
    every read of the kernel checks its own presence bookkeeping (element lanes' v_done, row headers) against a tag lookup.
    A disagreement is a bug even where the decoded string happens to survive it — the check that found the cause of the case
    above in one run."""
    out_dir = os.path.join(EMU, "_build_shadow")
    subprocess.check_call(["make", "-s", "-C", EMU, "OUT=" + out_dir, "EMU_DEFS=-DPO_EMU_SHADOW"])
    lib = os.path.join(out_dir, "libemu_pair_beam.so")
    for model, W, T in [("flipflop", "5", "700"), ("ctc", "5", "700"), ("merge", "10", "400")]:
        out = subprocess.run([sys.executable, os.path.join(EMU, "check_emu.py"), "--n", "16", "--T", T, "--W", W, "--model", model, "--seed", "77",
                              "--procs", "4", "--kernel", "reg", "--lib", lib], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        assert "0 mismatches" in out.stdout
        assert "SHADOW" not in out.stdout and "SHADOW" not in out.stderr, (out.stdout + out.stderr)[-2000:]
    # ... and eight pairs one after the other through ONE slice of the store: nothing an earlier pair left may be read as present
    out = subprocess.run([sys.executable, os.path.join(EMU, "check_emu.py"), "--n", "16", "--T", "400", "--W", "5", "--model", "flipflop", "--seed", "78",
                          "--procs", "2", "--kernel", "reg", "--batch", "8", "--slots", "1", "--lib", lib], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "0 mismatches" in out.stdout
    assert "SHADOW" not in out.stdout and "SHADOW" not in out.stderr, (out.stdout + out.stderr)[-2000:]
