"""GPU: the diagnostic counters of the pair beam search (po_profile_update_counter: update_prob evaluations the reference's
schedule makes for the decoded pairs, and those the kernel executed).  With a counter attached the register-state kernel
runs an instantiation of its own (beam2d_reg_kernel<1, true>, round 4): it must decode the same strings as the product
path, and what it counts must make sense."""
import ctypes as C

import numpy as np
import pytest

from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu


def _hip():
    """the HIP runtime the engine itself is linked against, for sixteen bytes of device memory (the C-ABI allocates none)"""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    return hip


def test_counting_instantiation_decodes_the_same(oracle):
    from poreover_amd import _lib, batch
    lib = _lib.load()
    hip = _hip()
    y1s, y2s, envs = [], [], []
    for i in range(24):
        a, b = synth_pair(7300 + i, T=500 + 90 * (i % 7))
        y1s.append(a); y2s.append(b)
        envs.append(np.asarray(oracle.pair_decode(a, b, "poreover", 5, "row_col")["envelope"]))
    want = [oracle.cpp_beam_search_2d(a, b, e, 5, model_="ctc", method_="row_col") for a, b, e in zip(y1s, y2s, envs)]
    plain = batch.beam_search_2d_batch(y1s, y2s, envs, 5, model="ctc", method="row_col")
    assert plain == want
    counts = {}
    for route in ("reg", "legacy"):
        _lib.set_pair_route(route)
        d_upd = C.c_void_p()
        assert hip.hipMalloc(C.byref(d_upd), 16) == 0 and hip.hipMemset(d_upd, 0, 16) == 0
        host = (C.c_uint64 * 2)()
        try:
            assert lib.po_profile_update_counter(d_upd) == 0
            counted = batch.beam_search_2d_batch(y1s, y2s, envs, 5, model="ctc", method="row_col")
            assert hip.hipDeviceSynchronize() == 0
            assert hip.hipMemcpy(host, d_upd, 16, 2) == 0     # hipMemcpyDeviceToHost
        finally:
            lib.po_profile_update_counter(None)
            _lib.set_pair_route("auto")
            hip.hipFree(d_upd)
        assert counted == want, route
        counts[route] = [int(host[0]), int(host[1])]
    for route, (ref, exe) in counts.items():
        assert 0 < exe <= ref, (route, ref, exe)
    # both kernels count the schedule of the reference (to a few no-op catch-ups); the register-state kernel executes less of it
    assert abs(counts["reg"][0] - counts["legacy"][0]) <= 0.02 * counts["legacy"][0], counts
    assert counts["reg"][1] <= counts["legacy"][1], counts


def test_reg_pool_prewarm_and_release(oracle):
    """po_reg_pool_prewarm / po_reg_pool_release: the register-state kernel's slice pool made ahead of the first
    launch, given back, and made again by the next call's size query — results identical before and after"""
    from poreover_amd import _lib, batch
    from poreover_amd.synth import synth_pair
    _lib.load()
    pairs = [synth_pair(880 + i, T=700) for i in range(6)]
    want = [oracle.pair_decode(a, b, "poreover", 5, "row_col")["consensus"] for a, b in pairs]
    _lib.reg_pool_prewarm("ctc", 5)
    _lib.reg_pool_prewarm("ctc", 10)
    got = batch.pair_decode_batch([p[0] for p in pairs], [p[1] for p in pairs], "poreover", 5, "row_col")
    assert [g["consensus"] for g in got] == want
    _lib.reg_pool_release()
    got = batch.pair_decode_batch([p[0] for p in pairs], [p[1] for p in pairs], "poreover", 5, "row_col")   # (the pool is made again)
    assert [g["consensus"] for g in got] == want
    with pytest.raises(_lib.EngineError):
        _lib.check(_lib.load().po_reg_pool_prewarm(0, 13), "po_reg_pool_prewarm")   # (W > 12 is beam2d_kernel's)
