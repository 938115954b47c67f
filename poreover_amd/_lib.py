"""ctypes binding of libporeover_hip.so (include/poreover_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or no GPU is visible,
calls raise EngineUnavailable.  (The CPU restatement under oracle/ is test infrastructure and
is never imported from here.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# POREOVER_HIP_LIB selects an alternative build of the same library (A/B timing builds)
LIB_PATH = os.environ.get("POREOVER_HIP_LIB") or os.path.join(HERE, "libporeover_hip.so")

OK = 0
E_CAP, E_ARG, E_ENVELOPE, E_NOMEM, E_DIVERGE, E_UNSUPPORTED, E_HIP = -1, -2, -3, -4, -5, -6, -7
SKIP_LENGTH, SKIP_IDENTITY = -10, -11
MODELS = {"ctc": 0, "ctc_merge_repeats": 1, "ctc_flipflop": 2}
METHODS = {"row": 0, "row_col": 1, "grid": 2}
KINDS = {"poreover": 0, "bonito": 1, "flipflop": 2}
K_VITERBI, K_BEAM1D, K_BEAM2D, K_ALIGN, K_ENVELOPE, K_BEAM2D_MAIN = range(6)
_CODE_NAMES = {E_CAP: "PO_E_CAP (buffer too small)", E_ARG: "PO_E_ARG (bad argument)",
               E_ENVELOPE: "PO_E_ENVELOPE (envelope undefined for the reference)",
               E_NOMEM: "PO_E_NOMEM (node arena / band capacity exceeded)",
               E_DIVERGE: "PO_E_DIVERGE (the reference never terminates on this input)",
               E_UNSUPPORTED: "PO_E_UNSUPPORTED", E_HIP: "PO_E_HIP"}


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    def __init__(self, code, what, detail=""):
        super().__init__("%s: %s%s" % (what, _CODE_NAMES.get(code, "code %d" % code),
                                       (" — " + detail) if detail else ""))
        self.code = code


class PairOptions(C.Structure):
    _fields_ = [("beam_width", C.c_int), ("model", C.c_int), ("method", C.c_int), ("padding", C.c_int),
                ("full_alignment", C.c_int), ("diagonal_envelope", C.c_int), ("diagonal_width", C.c_int)]


_vp, _i64p, _i32p, _dp, _cp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
# every symbol include/poreover_hip.h declares: (restype, argtypes)
PROTOTYPES = {
    "po_version": (C.c_int, []),
    "po_device_count": (C.c_int, []),
    "po_set_device": (C.c_int, [C.c_int]),
    "po_last_error": (C.c_char_p, []),
    "po_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_size_t)]),
    "po_set_pair_route": (C.c_int, [C.c_int, C.c_int]),
    "po_set_chain_mode": (C.c_int, [C.c_int]),
    "po_get_chain_mode": (C.c_int, []),
    "po_reg_pool_prewarm": (C.c_int, [C.c_int, C.c_int]),
    "po_reg_pool_release": (C.c_int, []),
    "po_debug_deferred_pairs": (C.c_longlong, [C.c_int]),
    "po_set_align_route": (C.c_int, [C.c_int]),
    "po_ingest_batch": (C.c_int, [_vp, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _dp, _vp]),
    "po_ingest_batch_h": (C.c_int, [_vp, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _dp]),
    "po_viterbi_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "po_viterbi_batch": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _vp, _cp, _i64p, _i32p,
                                   _i32p, _i32p, _vp, C.c_size_t, _vp]),
    "po_beam1d_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "po_beam1d_batch": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, _cp, _i64p, _i32p,
                                  _i32p, _vp, C.c_size_t, _vp]),
    "po_beam2d_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                               C.c_int, C.c_int, C.c_int]),
    "po_beam2d_batch": (C.c_int, [_dp, _i64p, _dp, _i64p, _i32p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                  C.c_int, _cp, _i64p, _i32p, _i32p, _vp, C.c_size_t, _vp]),
    "po_forward_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int]),
    "po_forward_batch": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _cp, _i64p, _dp, _i32p, _vp,
                                   C.c_size_t, _vp]),
    "po_viterbi_acceptor_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64]),
    "po_viterbi_acceptor_batch": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _cp, _i64p, _i32p,
                                            _i32p, _vp, C.c_size_t, _vp]),
    "po_forward_batch_h": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _cp, _i64p, _dp, _i32p]),
    "po_viterbi_acceptor_batch_h": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _cp, _i64p, _i32p,
                                              _i32p]),
    "po_prefix_search_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "po_prefix_search_batch": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, _cp, _i64p, _i32p, _dp, _i32p, _vp,
                                         C.c_size_t, _vp]),
    "po_prefix_search_batch_h": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, _cp, _i64p, _i32p, _dp, _i32p]),
    "po_align_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64, C.c_int]),
    "po_align_batch": (C.c_int, [_cp, _i64p, C.c_int, C.c_int, _cp, _cp, _i64p, _i32p, _i32p, _vp, C.c_size_t, _vp]),
    "po_align_batch_h": (C.c_int, [_cp, _i64p, C.c_int, C.c_int, _cp, _cp, _i64p, _i32p, _i32p]),
    "po_align_scores_batch_h": (C.c_int, [_cp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _cp, _cp, _i64p, _i32p, _i32p]),
    "po_nw_matrix_batch": (C.c_int, [_cp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i64p, _i32p, _vp]),
    "po_nw_matrix_batch_h": (C.c_int, [_cp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i64p, _i32p]),
    "po_envelope_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "po_envelope_batch": (C.c_int, [_cp, _cp, _i64p, _i32p, C.c_int, _i32p, _i64p, _i32p, _i64p, _i32p, _i32p, C.c_int,
                                    _i32p, _i64p, _i32p, _vp, C.c_size_t, _vp]),
    "po_envelope_batch_h": (C.c_int, [_cp, _cp, _i64p, _i32p, C.c_int, _i32p, _i64p, _i32p, _i64p, _i32p, _i32p, C.c_int,
                                      _i32p, _i64p, _i32p]),
    "po_pair_gamma_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64, C.c_int64]),
    "po_pair_gamma_batch": (C.c_int, [_dp, _i64p, _dp, _i64p, _i32p, _i64p, C.c_int, C.c_int, C.c_int, C.c_int64, _dp, _dp,
                                      _i64p, _i32p, _vp, C.c_size_t, _vp]),
    "po_pair_gamma_batch_h": (C.c_int, [_dp, _i64p, _dp, _i64p, _i32p, _i64p, C.c_int, C.c_int, C.c_int, _dp, _dp, _i64p,
                                        _i32p]),
    "po_pair_decode_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                                    C.POINTER(PairOptions)]),
    "po_pair_decode_batch": (C.c_int, [_dp, _i64p, _dp, _i64p, C.c_int, C.c_int, C.POINTER(PairOptions), _cp,
                                       _i64p, _i32p, _i32p, _dp, _i32p, _cp, _i64p, _i32p, _i32p, _vp,
                                       C.c_size_t, _vp]),
    "po_decode_1d_batch_h": (C.c_int, [_vp, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int,
                                       C.c_int, C.c_int, _cp, _i64p, _i32p, _i32p]),
    "po_viterbi_batch_h": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _vp, _cp, _i64p, _i32p,
                                     _i32p, _i32p]),
    "po_beam1d_batch_h": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, _cp, _i64p,
                                    _i32p, _i32p]),
    "po_beam2d_batch_h": (C.c_int, [_dp, _i64p, _dp, _i64p, _i32p, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                    C.c_int, C.c_int, _cp, _i64p, _i32p, _i32p]),
    "po_pair_decode_batch_h": (C.c_int, [_dp, _i64p, _dp, _i64p, C.c_int, C.c_int, C.POINTER(PairOptions), _cp,
                                         _i64p, _i32p, _i32p, _dp, _i32p, _cp, _i64p, _i32p, _i32p]),
    "po_pair_decode_from_1d_batch_h": (C.c_int, [_dp, _i64p, _dp, _i64p, C.c_int, C.c_int, C.POINTER(PairOptions), _cp,
                                                 _i64p, _i32p, _i32p, _i32p, _i32p, _dp, _i32p, _cp, _i64p, _i32p, _i32p]),
    "po_profile_update_counter": (C.c_int, [C.c_void_p]),
    "po_lae_peak": (C.c_int, [C.c_int, _dp, C.c_void_p]),
    "po_pair_prefix_search_batch_h": (C.c_int, [_dp, _i64p, _dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _cp, _i64p, _i32p,
                                                _dp, _i32p]),
    "po_pair_prefix_search_env_batch_h": (C.c_int, [_dp, _i64p, _dp, _i64p, _i32p, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _cp,
                                                    _i64p, _i32p, _dp, _i32p]),
    "po_forward_vec_batch_h": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "po_viterbi_acceptor_cy_batch_h": (C.c_int, [_dp, _i64p, C.c_int, C.c_int, C.c_char_p, C.c_int, _cp, _i64p, _i32p,
                                              _i32p]),
    "po_pipeline_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int64, C.c_int]),
    "po_pipeline_destroy": (None, [C.c_void_p]),
    "po_pipeline_pair_decode": (C.c_int, [C.c_void_p, _vp, _i64p, _vp, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), C.c_int, C.POINTER(PairOptions), _cp, _i64p, _i32p, _i32p, _dp,
                                          _i32p, _cp, _i64p, _i32p, _i32p]),
    "po_multi_create": (C.c_void_p, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int64, C.c_int]),
    "po_multi_destroy": (None, [C.c_void_p]),
    "po_multi_devices": (C.c_int, [C.c_void_p]),
    "po_multi_pair_decode": (C.c_int, [C.c_void_p, _vp, _i64p, _vp, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.c_int, C.POINTER(PairOptions), _cp, _i64p, _i32p, _i32p, _dp,
                                       _i32p, _cp, _i64p, _i32p, _i32p]),
    "po_wave_plan": (C.c_int, [_i64p, _i64p, C.c_int, C.c_int, C.c_int64, C.c_int, _i32p, _i32p, C.c_int]),
    "po_multi_stats": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "po_pipeline_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int)]),
    "po_event_create": (C.c_void_p, []),
    "po_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "po_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "po_event_destroy": (None, [C.c_void_p]),
    "po_profile_enable": (None, [C.c_int]),
    "po_profile_reset": (None, []),
    "po_profile_get": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def load(require_gpu=True):
    """Load the HIP library (once).  Raises EngineUnavailable instead of falling back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                "libporeover_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError here means the .so is stale vs the header
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    if require_gpu and _lib.po_device_count() < 1:
        raise EngineUnavailable("no HIP device visible: the decoding engine needs an MI355X (gfx950) GPU; "
                                "there is no CPU fallback")
    return _lib


_CURRENT_DEVICE = [None]


def set_device(device):
    """Bind this process (its calling thread, as HIP does) to `device` for every later engine call, and remember it:
    the cached pipelines of batch.pair_decode_stream are per device and ask current_device()."""
    check(load().po_set_device(int(device)), "po_set_device(%d)" % int(device))
    _CURRENT_DEVICE[0] = int(device)


def current_device():
    """The device set_device() chose, else POREOVER_DEVICE (what dist.run_sharded gives its workers), else 0."""
    if _CURRENT_DEVICE[0] is not None:
        return _CURRENT_DEVICE[0]
    return int(os.environ.get("POREOVER_DEVICE", "0") or 0)


ROUTES = {"auto": 0, "legacy": 2, "reg": 4}


def set_pair_route(route="auto", defer_odd=False, starve=0):
    """test / tuning hook (po_set_pair_route): which kernel serves the pair beam search; defer_odd: every odd pair is handed
    to beam2d_kernel; starve: 1 = the register-state kernel runs with a dozen row groups, 2 = with a tiny tree arena (pairs run
    out of them and are handed on)"""
    check(load(False).po_set_pair_route(ROUTES[route], (1 if defer_odd else 0) | ((int(starve) & 3) << 1)), "po_set_pair_route")


CHAIN_MODES = {"serial": 0, "closed_form": 1, "closed_guard3": 2}


def set_chain_mode(mode="serial"):
    """po_set_chain_mode: how the register-state pair kernel computes a new element's window — "serial" (default): the
    reference's logaddexp chain; "closed_form": one exp, a prefix sum, one log per time (values within ~1e-12 of the
    reference's, strings inside its edit tolerance; measured slower, opt-in); "closed_guard3": the tests' way to its hand-over"""
    check(load(False).po_set_chain_mode(CHAIN_MODES[mode]), "po_set_chain_mode")


def get_chain_mode():
    return {v: k for k, v in CHAIN_MODES.items()}[int(load(False).po_get_chain_mode())]


def reg_pool_prewarm(model="ctc", beam_width=5):
    """po_reg_pool_prewarm: make the register-state pair kernel's slice pool for (model, beam_width) on the current device now"""
    check(load().po_reg_pool_prewarm(MODELS[model], int(beam_width)), "po_reg_pool_prewarm")


def reg_pool_release():
    """po_reg_pool_release: wait for the device and free every slice pool of the current device"""
    check(load().po_reg_pool_release(), "po_reg_pool_release")


def deferred_pairs(reset=False):
    """test hook (po_debug_deferred_pairs): pairs handed from the register-state kernel to beam2d_kernel since the last reset"""
    return int(load(False).po_debug_deferred_pairs(1 if reset else 0))


def set_align_route(legacy=False):
    """test / tuning hook (po_set_align_route): the banded aligner on its row-at-a-time kernel (True) or the skewed wavefront"""
    check(load(False).po_set_align_route(1 if legacy else 0), "po_set_align_route")


def check(rc, what):
    if rc != OK:
        detail = load(False).po_last_error()
        raise EngineError(rc, what, detail.decode() if detail else "")
