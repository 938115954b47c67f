"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/poreover_hip.h declares (no compute calls without a GPU), and the product fails loudly
instead of falling back when no GPU is present."""
import os
import re

import numpy as np
import pytest

from conftest import REPO


def _have_lib():
    from poreover_amd import _lib
    return os.path.exists(_lib.LIB_PATH)


def _ensure_built():
    if not _have_lib():
        from poreover_amd import build
        build.build()


def test_header_symbols_all_exported():
    _ensure_built()
    from poreover_amd import _lib
    lib = _lib.load(require_gpu=False)
    hdr = open(os.path.join(REPO, "include", "poreover_hip.h")).read()
    declared = set(re.findall(r"\b(po_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), "libporeover_hip.so does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    assert lib.po_version() >= 100


def test_build_objects_cover_every_source():
    """poreover_amd/build.py: every HIP source under csrc/ is an object of the library, exactly once — but po_beam2d_reg.hip,
    which is compiled as two objects (PO_REG_TU = 1: the 32-slot kernels and the C entry points; 2: the 64-slot kernels), so
    that each gets its own scheduler option; a variant object built without PO_REG_TU holds both (the tools, the emulator)."""
    import glob
    from poreover_amd import build
    on_disk = sorted(os.path.basename(f) for f in glob.glob(os.path.join(os.path.dirname(build.__file__), "csrc", "*.hip")))
    assert sorted(build.SOURCES) == on_disk
    by_src = {}
    for oname, sname, extra in build.OBJECTS:
        by_src.setdefault(sname, []).append((oname, extra))
    assert sorted(by_src) == on_disk
    assert len({o for o, _, _ in build.OBJECTS}) == len(build.OBJECTS)      # (object names are distinct)
    for sname, objs in by_src.items():
        if sname == "po_beam2d_reg.hip":
            tus = sorted(x for _, extra in objs for x in extra if x.startswith("-DPO_REG_TU="))
            assert tus == ["-DPO_REG_TU=1", "-DPO_REG_TU=2"], objs
        else:
            assert len(objs) == 1 and not any(x.startswith("-DPO_REG_TU") for x in objs[0][1]), (sname, objs)
    src = open(os.path.join(os.path.dirname(build.__file__), "csrc", "po_beam2d_reg.hip")).read()
    assert "PO_REG_TU == 1" in src and "PO_REG_TU == 2" in src and "po_reg_wide_launch" in src


def test_product_never_imports_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "poreover_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "po_oracle" in txt or "libporef" in txt:
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_fails_loudly_without_gpu():
    _ensure_built()
    from poreover_amd import _lib, batch
    if _lib.load(require_gpu=False).po_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.EngineUnavailable):
        batch.beam_search_batch([np.zeros((4, 5))], 5)
    with pytest.raises(_lib.EngineUnavailable):
        batch.viterbi_batch([np.zeros((4, 5))])


def test_pack_rows():
    from poreover_amd.batch import pack_rows
    y, off, C = pack_rows([np.zeros((3, 5)), np.ones((2, 5), dtype=np.float32)])
    assert y.shape == (5, 5) and y.dtype == np.float64 and off.tolist() == [0, 3, 5] and C == 5
    with pytest.raises(ValueError):
        pack_rows([np.zeros((3, 5)), np.zeros((3, 8))])


def test_workspace_bounded_for_long_reads():
    """the pair-decode workspace must not grow as resident workgroups x read length without bound —
    thousands of pairs of the reference's own sample lengths (62 000 / 75 600 frames) must plan within the board's
    288 GB (fewer resident workgroups instead)."""
    _ensure_built()
    import ctypes as C
    from poreover_amd import _lib
    lib = _lib.load(require_gpu=False)
    n = 8192
    for W, method in ((5, "row_col"), (10, "row_col"), (25, "row"), (5, "row")):
        opt = _lib.PairOptions(W, _lib.MODELS["ctc"], _lib.METHODS[method], 5, 0, 0, 50)
        ws = lib.po_pair_decode_workspace_bytes(n, n * 62000, n * 75600, 62000, 75600, 5, C.byref(opt))
        fixed = 4 * n * (62000 + 75600) + 8 * n * 62000      # frame maps + envelope: proportional to the input itself
        assert ws - fixed < 150 * 2**30, (W, method, ws / 2**30)
    opt = _lib.PairOptions(5, _lib.MODELS["ctc"], _lib.METHODS["row_col"], 5, 0, 0, 50)
    # ultra-long reads through the direct API (the pipeline bounds its waves itself) — the aligner's slices stay
    # within 1/16 of the board whatever the CU count says (a floor of one slice per CU was 27 GB at 4e5 frames, more beyond)
    var = {}
    for n_, mr in ((256, 1000000), (512, 1000000), (512, 400000)):
        ws = lib.po_pair_decode_workspace_bytes(n_, n_ * mr, n_ * mr, mr, mr, 5, C.byref(opt))
        var[(n_, mr)] = ws - (4 * n_ * 2 * mr + 8 * n_ * mr)
        assert var[(n_, mr)] < 72 * 2**30, (n_, mr, var[(n_, mr)] / 2**30)
    assert var[(512, 1000000)] < var[(256, 1000000)] + 24 * 2**30      # (twice the pairs: the per-pair arrays, not twice the slices)
    small = lib.po_pair_decode_workspace_bytes(10000, 40000000, 40000000, 4000, 4400, 5, C.byref(opt))
    assert small < 48 * 2**30      # (the bench configuration keeps its full occupancy: ~16 GB of DP slices + ~15 GB for the pair beam)


def test_wave_plan_single_device_short_first_and_last_wave():
    """the plan po_pipeline_pair_decode makes by itself for a job of several waves: a short first wave (the device starts
    early), doubling up to the full size, and a short LAST one (what follows it — download, copies, the caller's records — is
    serial)"""
    from poreover_amd import _lib
    lib = _lib.load(require_gpu=False)
    n = 10000
    r1 = np.full(n, 4000, dtype=np.int64); r2 = np.full(n, 3900, dtype=np.int64)
    first = np.zeros(32, dtype=np.int32); count = np.zeros(32, dtype=np.int32)
    k = lib.po_wave_plan(r1.ctypes.data_as(_lib._i64p), r2.ctypes.data_as(_lib._i64p), n, 0, 0, 1,
                         first.ctypes.data_as(_lib._i32p), count.ctypes.data_as(_lib._i32p), 32)
    c = count[:k].tolist()
    assert sum(c) == n and first[:k].tolist() == np.concatenate([[0], np.cumsum(c)[:-1]]).tolist()
    assert c[0] == 1250 and c[1] == 2500 and max(c) <= 3334 and c[-1] <= 800 and k == 5, c
    # a job of one wave is left alone
    k = lib.po_wave_plan(r1.ctypes.data_as(_lib._i64p), r2.ctypes.data_as(_lib._i64p), 4000, 0, 0, 1,
                         first.ctypes.data_as(_lib._i32p), count.ctypes.data_as(_lib._i32p), 32)
    assert k == 1 and count[0] == 4000


def test_wave_plan_covers_every_pair_in_order():
    """the planner of the multi-device pipeline (po_wave_plan: what po_multi_pair_decode hands out): every pair in
    exactly one wave, waves in input order, no wave beyond the pair / frame limits, and with several devices at
    least two waves per device so that each has one decoding while the next uploads"""
    from poreover_amd import _lib
    lib = _lib.load(require_gpu=False)
    rng = np.random.default_rng(5)
    for n, ndev, wp, wr in ((10000, 8, 4096, 0), (10000, 1, 4096, 0), (37, 2, 7, 0), (5, 8, 4096, 0), (1000, 4, 100, 50000), (0, 2, 0, 0)):
        r1 = rng.integers(50, 4500, size=max(n, 1)).astype(np.int64)
        r2 = rng.integers(50, 4500, size=max(n, 1)).astype(np.int64)
        cap = n + 8
        first = np.zeros(cap, dtype=np.int32)
        count = np.zeros(cap, dtype=np.int32)
        k = lib.po_wave_plan(r1.ctypes.data_as(_lib._i64p), r2.ctypes.data_as(_lib._i64p), n, wp, wr, ndev,
                             first.ctypes.data_as(_lib._i32p), count.ctypes.data_as(_lib._i32p), cap)
        assert k >= 0
        assert int(count[:k].sum()) == n
        pos = 0
        for j in range(k):
            assert first[j] == pos and count[j] >= 1
            limit = wp or 4096
            assert count[j] <= limit
            if wr:
                tot = int(r1[pos:pos + count[j]].sum() + r2[pos:pos + count[j]].sum())
                assert count[j] == 1 or tot <= wr
            pos += count[j]
        if ndev > 1 and n >= 2 * ndev:
            assert k >= 2 * ndev, (n, ndev, k)
            # near-equal waves: no device is handed a wave more than one pair larger than another's (frame limit aside)
            if not wr:
                assert count[:k].max() - count[:k - 1].min() <= 1


def test_pipeline_device_follows_set_device(monkeypatch):
    """batch._pipeline() asks the library layer for the process's device (under torchrun every rank's
    pipeline sat on GPU 0 because only POREOVER_DEVICE was looked at)"""
    from poreover_amd import _lib
    monkeypatch.delenv("POREOVER_DEVICE", raising=False)
    monkeypatch.setattr(_lib, "_CURRENT_DEVICE", [None])
    assert _lib.current_device() == 0
    monkeypatch.setenv("POREOVER_DEVICE", "3")
    assert _lib.current_device() == 3
    monkeypatch.setattr(_lib, "_CURRENT_DEVICE", [5])      # what set_device(5) records
    assert _lib.current_device() == 5


def test_devices_are_bound_through_set_device_only():
    """po_set_device must be reached through _lib.set_device, which also records the device for the cached pipelines
    (batch._pipeline): a direct call leaves every rank's pipeline on GPU 0 (bench.py's strong-scaling leg
    repeated it in round 3)"""
    offenders = []
    for root in ("poreover_amd", "."):
        base = os.path.join(REPO, root)
        for dirpath, _dirs, files in os.walk(base):
            if root == "." and dirpath != base:
                continue
            for f in files:
                if f.endswith(".py") and f != "_lib.py":
                    with open(os.path.join(dirpath, f)) as fh:
                        if "po_set_device(" in fh.read():
                            offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
