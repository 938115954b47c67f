"""Batched numpy front end of the engine: lists of (T, C) arrays in, lists of strings out.

These are what the drivers (decode / pair-decode) call instead of the reference's
multiprocessing.Pool fan-out (decode.py:158-162, pair_decode.py:292-297): one launch per batch.
Host buffers go through the *_h entry points of the C-ABI (which copy to the device, launch,
and copy back); device-resident callers use the device-pointer forms directly (see bench.py).
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L

__all__ = ["viterbi_batch", "beam_search_batch", "beam_search_2d_batch", "pair_decode_batch", "pair_decode_batch_sharded", "pair_decode_stream", "decode_1d_batch", "pack_rows",
           "forward_batch", "viterbi_acceptor_batch", "prefix_search_batch", "pair_prefix_search_batch", "forward_vec_batch", "align_batch", "envelope_batch", "ingest_batch", "pair_gamma_batch"]


def pack_rows(arrays, C_expected=None):
    """Concatenate (T_i, C) arrays into one float64 C-contiguous matrix + int64 row offsets."""
    mats = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
    for m in mats:
        if m.ndim != 2:
            raise ValueError("expected (T, C) matrices")
    Cc = mats[0].shape[1] if mats else (C_expected or 5)
    if any(m.shape[1] != Cc for m in mats):
        raise ValueError("all matrices of a batch must have the same number of columns")
    off = np.zeros(len(mats) + 1, dtype=np.int64)
    np.cumsum([m.shape[0] for m in mats], out=off[1:])
    y = np.concatenate(mats, axis=0) if mats else np.zeros((0, Cc))
    return np.ascontiguousarray(y), off, Cc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_SCRATCH = None          # threading.local: a thread's buffers die with it
_SCRATCH_KEEP = 1 << 30  # bytes a thread keeps between calls; a larger buffer is dropped by release_scratch()


def _scratch(name, nbytes):
    """A uint8 buffer of at least nbytes that survives the call (grow-only, one per thread and name; freed with the
    thread, or by release_scratch())."""
    import threading
    global _SCRATCH
    if _SCRATCH is None:
        _SCRATCH = threading.local()
    bufs = _SCRATCH.__dict__.setdefault("bufs", {})
    buf = bufs.get(name)
    if buf is None or buf.size < nbytes:
        buf = np.empty(int(nbytes * 1.25) + 4096, dtype=np.uint8)
        bufs[name] = buf
    return buf[:nbytes]


def release_scratch(keep_bytes=0):
    """Drop this thread's text buffers larger than keep_bytes (a long-running process after one large job)."""
    if _SCRATCH is None:
        return
    bufs = _SCRATCH.__dict__.get("bufs", {})
    for k in [k for k, b in bufs.items() if b.size > keep_bytes]:
        del bufs[k]


_PENDING = -(2 ** 31)   # a status no decode returns: "not written yet" (pair_decode_stream)


def _addresses(arrays):
    """Data pointers of a list of C-contiguous arrays as uint64.  ctypes' from_buffer + addressof is three times faster
    than __array_interface__ (no dict per array) — 20 000 arrays per call sit on the end-to-end clock — but wants a
    writable buffer; read-only arrays (memory maps) take the slow way."""
    fb, ao = C.c_char.from_buffer, C.addressof
    try:
        return np.fromiter((ao(fb(a)) for a in arrays), dtype=np.uint64, count=len(arrays))
    except (TypeError, ValueError, BufferError):
        return np.fromiter((a.__array_interface__["data"][0] for a in arrays), dtype=np.uint64, count=len(arrays))


def _strings(buf, off, lens):
    raw = buf.tobytes()
    return [raw[off[i]:off[i] + lens[i]].decode("ascii") for i in range(len(lens))]


def viterbi_batch(arrays, kind="poreover", alphabet="ACGT", return_path=False, return_map=False):
    """transducer.*.viterbi_decode for a batch.  Returns a list of sequences (and paths / maps)."""
    lib = L.load()
    y, off, Cc = pack_rows(arrays)
    n = len(arrays)
    rows = int(off[-1])
    seq = np.zeros(max(rows, 1), dtype=np.uint8)
    lens = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    path = np.zeros(max(rows, 1), dtype=np.int8)
    mp = np.zeros(max(rows, 1), dtype=np.int32) if return_map else None
    L.check(lib.po_viterbi_batch_h(_ptr(y), _ptr(off), n, Cc, alphabet.encode(), L.KINDS[kind], _ptr(path),
                                   _ptr(seq), _ptr(off), _ptr(lens), _ptr(mp), _ptr(st)), "po_viterbi_batch_h")
    for i in range(n):
        if st[i] != 0 and not (return_map and st[i] == L.E_ARG):
            raise L.EngineError(int(st[i]), "viterbi decode of read %d" % i)
    out = [_strings(seq, off, lens)]
    if return_path:
        out.append([path[off[i]:off[i + 1]].astype(np.int64) for i in range(n)])
    if return_map:
        out.append([mp[off[i]:off[i] + lens[i]].astype(np.int64) for i in range(n)])
        out.append(st[:n].copy())
    return out[0] if len(out) == 1 else tuple(out)


def beam_search_batch(arrays, beam_width=25, alphabet="ACGT", model="ctc"):
    """decoding_cpp.cpp_beam_search for a batch of reads."""
    lib = L.load()
    y, off, Cc = pack_rows(arrays)
    n = len(arrays)
    seq = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    lens = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    L.check(lib.po_beam1d_batch_h(_ptr(y), _ptr(off), n, Cc, alphabet.encode(), int(beam_width), L.MODELS[model],
                                  _ptr(seq), _ptr(off), _ptr(lens), _ptr(st)), "po_beam1d_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "beam search of read %d" % i)
    return _strings(seq, off, lens)


def decode_1d_batch(arrays, kind="poreover", algorithm="viterbi", beam_width=25, alphabet="ACGT", perm=None, reverse=False):
    """`poreover decode` for a batch of reads in ONE engine call (po_decode_1d_batch_h): the arrays are the basecaller's
    own output — float32 logits, uint8 flip-flop traces or float64 log-probabilities, all of one dtype — and are
    uploaded as they are; log-softmax / trace scaling / column order run on the device, then Viterbi or the 1-D beam
    search.  Returns the sequences."""
    lib = L.load()
    n = len(arrays)
    if n == 0:
        return []
    arrs = [np.ascontiguousarray(a) for a in arrays]
    dt = arrs[0].dtype
    mode = INGEST_MODES.get(np.dtype(dt))
    if mode is None or any(a.dtype != dt or a.ndim != 2 for a in arrs):
        raise ValueError("decode_1d_batch takes 2-D float32 logits, uint8 traces or float64 log-probabilities of one dtype")
    Cc = arrs[0].shape[1]
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(a) for a in arrs], out=off[1:])
    src = np.ascontiguousarray(np.concatenate(arrs, axis=0))
    seq = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    st = np.zeros(n, dtype=np.int32)
    model = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}[kind]
    pm = (C.c_int * Cc)(*perm) if perm is not None else None
    L.check(lib.po_decode_1d_batch_h(_ptr(src), _ptr(off), n, Cc, mode, pm, 1 if reverse else 0, alphabet.encode(), L.KINDS[kind],
                                     int(beam_width) if algorithm == "beam" else 0, L.MODELS[model], _ptr(seq), _ptr(off),
                                     _ptr(lens), _ptr(st)), "po_decode_1d_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "decode of read %d" % i)
    return _strings(seq, off, lens)


def beam_search_2d_batch(arrays1, arrays2, envelopes, beam_width=25, alphabet="ACGT", model="ctc",
                         method="row", return_status=False):
    """decoding_cpp.cpp_beam_search_2d for a batch of pairs; envelopes: list of (U_i, 2) or None."""
    lib = L.load()
    y1, o1, Cc = pack_rows(arrays1)
    y2, o2, _ = pack_rows(arrays2, Cc)
    n = len(arrays1)
    env = None
    if envelopes is not None:
        es = [np.ascontiguousarray(e, dtype=np.int32) for e in envelopes]
        for e, a in zip(es, arrays1):
            if e.ndim != 2 or e.shape[1] != 2 or e.shape[0] < len(a):
                raise ValueError("envelope must be (U, 2)")
        env = np.ascontiguousarray(np.concatenate([e[:len(a)] for e, a in zip(es, arrays1)], axis=0))
    so = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(a) + len(b) for a, b in zip(arrays1, arrays2)], out=so[1:])
    seq = np.zeros(max(int(so[-1]), 1), dtype=np.uint8)
    lens = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    L.check(lib.po_beam2d_batch_h(_ptr(y1), _ptr(o1), _ptr(y2), _ptr(o2), _ptr(env), n, Cc, alphabet.encode(),
                                  int(beam_width), L.MODELS[model], L.METHODS[method], _ptr(seq), _ptr(so),
                                  _ptr(lens), _ptr(st)), "po_beam2d_batch_h")
    if not return_status:
        for i in range(n):
            if st[i] != 0:
                raise L.EngineError(int(st[i]), "pair beam search of pair %d" % i)
    seqs = _strings(seq, so, lens)
    return (seqs, st[:n].copy()) if return_status else seqs


def pair_decode_batch(arrays1, arrays2, kind="poreover", beam_width=5, method="row_col", padding=5,
                      alignment="banded", diagonal_envelope=False, diagonal_width=50, single="viterbi"):
    """pair_decode_helper stage chain (pair_decode.py:305-529) for a batch of pairs, all on the GPU.
    single="viterbi" (default): 1-D basecalls by argmax; single="beam": by cpp_beam_search (W = 25) with
    frame maps from cpp_viterbi_acceptor (band 1000), as pair_decode.py:363-370.
    Returns a list of dicts: seq1, seq2, consensus (None if skipped), length1, length2,
    sequence_identity, skipped, status, envelope."""
    lib = L.load()
    y1, o1, Cc = pack_rows(arrays1)
    y2, o2, _ = pack_rows(arrays2, Cc)
    n = len(arrays1)
    model = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}[kind]
    opt = L.PairOptions(int(beam_width), L.MODELS[model], L.METHODS[method], int(padding),
                        1 if alignment == "full" else 0, 1 if diagonal_envelope else 0, int(diagonal_width))
    s1o = np.zeros(2 * n + 1, dtype=np.int64)
    caps = []
    for a, b in zip(arrays1, arrays2):
        caps += [len(a), len(b)]
    np.cumsum(caps, out=s1o[1:])
    so = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(a) + len(b) for a, b in zip(arrays1, arrays2)], out=so[1:])
    seq1d = np.zeros(max(int(s1o[-1]), 1), dtype=np.uint8)
    seq = np.zeros(max(int(so[-1]), 1), dtype=np.uint8)
    l1, l2, lens, st = (np.zeros(max(n, 1), dtype=np.int32) for _ in range(4))
    ident = np.zeros(max(n, 1), dtype=np.float64)
    env = np.zeros((max(int(o1[-1]), 1), 2), dtype=np.int32)
    if single == "beam" and not diagonal_envelope:
        if kind != "poreover":
            raise L.EngineError(L.E_UNSUPPORTED, "pair decode --single beam", "only for the poreover (ctc) kind, as the "
                                "reference's acceptor is")
        # pair_decode.py:363-370 calls cpp_beam_search / cpp_viterbi_acceptor with their defaults
        b1, b2 = beam_search_batch(arrays1, 25), beam_search_batch(arrays2, 25)
        p1, p2 = viterbi_acceptor_batch(arrays1, b1, 1000), viterbi_acceptor_batch(arrays2, b2, 1000)
        map1 = np.zeros(max(int(o1[-1]), 1), dtype=np.int32)
        map2 = np.zeros(max(int(o2[-1]), 1), dtype=np.int32)
        for i in range(n):   # get_sequence_mapping('poreover'): frames whose state is a base
            for mp, off, path, bs, ln, slot in ((map1, o1, p1[i], b1[i], l1, 2 * i), (map2, o2, p2[i], b2[i], l2, 2 * i + 1)):
                fr = np.nonzero(path < 4)[0]
                if len(fr) != len(bs):
                    raise L.EngineError(L.E_ARG, "pair decode --single beam", "frame map and basecall lengths differ "
                                        "(the reference asserts here, pair_decode.py:379)")
                mp[off[i]:off[i] + len(fr)] = fr
                ln[i] = len(bs)
                seq1d[s1o[slot]:s1o[slot] + len(bs)] = np.frombuffer(bs.encode("ascii"), dtype=np.uint8)
        L.check(lib.po_pair_decode_from_1d_batch_h(_ptr(y1), _ptr(o1), _ptr(y2), _ptr(o2), n, Cc, C.byref(opt), _ptr(seq1d),
                                                   _ptr(s1o), _ptr(l1), _ptr(l2), _ptr(map1), _ptr(map2), _ptr(ident),
                                                   _ptr(env), _ptr(seq), _ptr(so), _ptr(lens), _ptr(st)),
                "po_pair_decode_from_1d_batch_h")
    elif single not in ("viterbi", "beam"):
        raise ValueError("single must be 'viterbi' or 'beam'")
    else:
        L.check(lib.po_pair_decode_batch_h(_ptr(y1), _ptr(o1), _ptr(y2), _ptr(o2), n, Cc, C.byref(opt), _ptr(seq1d),
                                           _ptr(s1o), _ptr(l1), _ptr(l2), _ptr(ident), _ptr(env), _ptr(seq), _ptr(so),
                                           _ptr(lens), _ptr(st)), "po_pair_decode_batch_h")
    raw1, raw = seq1d.tobytes(), seq.tobytes()
    out = []
    for i in range(n):
        code = int(st[i])
        if code not in (0, L.SKIP_LENGTH, L.SKIP_IDENTITY):
            raise L.EngineError(code, "pair decode of pair %d" % i)
        out.append({
            "seq1": raw1[s1o[2 * i]:s1o[2 * i] + l1[i]].decode("ascii"),
            "seq2": raw1[s1o[2 * i + 1]:s1o[2 * i + 1] + l2[i]].decode("ascii"),
            "consensus": raw[so[i]:so[i] + lens[i]].decode("ascii") if code == 0 else None,
            "length1": int(l1[i]), "length2": int(l2[i]),
            "sequence_identity": float(ident[i]) if code != L.SKIP_LENGTH else None,
            "skipped": 0 if code == 0 else 1, "status": code,
            "envelope": env[o1[i]:o1[i + 1]].astype(np.int64) if code == 0 else None})
    return out


_PIPELINES = {}
INGEST_MODES = {np.dtype(np.float32): 0, np.dtype(np.uint8): 1, np.dtype(np.float64): 2}


def _pipeline(wave_pairs=0, wave_rows=0, threads=0, device=None):
    """One po_pipeline per (process, device, geometry): its pinned staging buffers, device buffers and workspace
    are allocated once and reused by every call.  The device is the caller's (`device`), else the one the process was
    bound to (_lib.set_device — the torchrun branches of the drivers, dist.run_sharded's workers), else 0."""
    dev = int(device) if device is not None else L.current_device()
    key = (os.getpid(), dev, int(wave_pairs), int(wave_rows), int(threads))
    pl = _PIPELINES.get(key)
    if pl is None:
        lib = L.load()
        pl = lib.po_pipeline_create(dev, int(wave_pairs), int(wave_rows), int(threads))
        if not pl:
            raise L.EngineError(L.E_HIP, "po_pipeline_create", (lib.po_last_error() or b"").decode())
        _PIPELINES[key] = pl
    return pl


_MULTIS = {}


def _multi(devices, wave_pairs=0, wave_rows=0, threads=0):
    """One po_multi (a pipeline and a host thread per device, one wave planner) per (process, device list, geometry)."""
    devs = tuple(int(d) for d in devices)
    key = (os.getpid(), devs, int(wave_pairs), int(wave_rows), int(threads))
    m = _MULTIS.get(key)
    if m is None:
        lib = L.load()
        m = lib.po_multi_create((C.c_int * len(devs))(*devs), len(devs), int(wave_pairs), int(wave_rows), int(threads))
        if not m:
            raise L.EngineError(L.E_HIP, "po_multi_create", (lib.po_last_error() or b"").decode())
        _MULTIS[key] = m
    return m


_PIPE_LOCKS = {}


def _pipeline_lock(handle):
    import threading
    key = int(handle) if not isinstance(handle, int) else handle
    lk = _PIPE_LOCKS.get(key)
    if lk is None:
        lk = _PIPE_LOCKS.setdefault(key, threading.Lock())
    return lk


def pair_decode_stream(arrays1, arrays2, kind="poreover", beam_width=5, method="row_col", padding=5, alignment="banded",
                       diagonal_envelope=False, diagonal_width=50, perm1=None, perm2=None, reverse2=False,
                       return_envelope=False, wave_pairs=0, wave_rows=0, threads=0, strict=True, stats=None,
                       devices=None):
    """The pair-decode stage chain for a list of pairs, HOST ARRAYS IN -> STRINGS OUT, through the engine's
    pipelined host layer (po_pipeline_pair_decode): the arrays are uploaded as they are — float32 logits, uint8
    flip-flop traces or float64 log-probabilities, all of one dtype — in waves, log-softmax / trace scaling /
    column order (perm1, perm2: out[:, c] = in[:, perm[c]]) / time reversal of read 2 (reverse2; reverse_complement =
    reverse2 + perm2 [3,2,1,0,4]) run on the device, and wave k + 1 uploads while wave k decodes.
    Returns the same records as pair_decode_batch (envelope only with return_envelope).  strict=False: a per-pair
    engine error is left in the record's status instead of raising for the whole batch.
    devices: a list of device indices (an index may repeat) -> ONE process drives them all (po_multi_pair_decode: a
    pipeline and a host thread per device, waves dealt as devices become free, results written in input order);
    None -> the process's own device."""
    import time as _time
    _t0 = _time.perf_counter()
    lib = L.load()
    n = len(arrays1)
    if n == 0:
        return []
    # (marshalling 10^4 pairs is 2 x 10^4 small Python operations per line below: every one of them is on the
    #  end-to-end clock, hence the flags / __array_interface__ / tolist forms)
    a1 = [a if a.flags.c_contiguous else np.ascontiguousarray(a) for a in arrays1]
    a2 = [a if a.flags.c_contiguous else np.ascontiguousarray(a) for a in arrays2]
    dt = a1[0].dtype
    mode = INGEST_MODES.get(np.dtype(dt))
    Cc = a1[0].shape[1] if a1[0].ndim == 2 else -1
    ok = mode is not None and Cc > 0
    if ok:
        for a in a1:
            if a.dtype != dt or a.ndim != 2 or a.shape[1] != Cc:
                ok = False
                break
    if ok:
        for a in a2:
            if a.dtype != dt or a.ndim != 2 or a.shape[1] != Cc:
                ok = False
                break
    if not ok:
        raise ValueError("pair_decode_stream takes 2-D float32 logits, uint8 traces or float64 log-probabilities of one "
                         "dtype and one column count")
    model = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}[kind]
    opt = L.PairOptions(int(beam_width), L.MODELS[model], L.METHODS[method], int(padding),
                        1 if alignment == "full" else 0, 1 if diagonal_envelope else 0, int(diagonal_width))
    r1 = np.fromiter((a.shape[0] for a in a1), dtype=np.int64, count=n)
    r2 = np.fromiter((a.shape[0] for a in a2), dtype=np.int64, count=n)
    p1 = _addresses(a1)
    p2 = _addresses(a2)
    s1o = np.zeros(2 * n + 1, dtype=np.int64)
    caps = np.empty(2 * n, dtype=np.int64)
    caps[0::2], caps[1::2] = r1, r2
    np.cumsum(caps, out=s1o[1:])
    so = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(r1 + r2, out=so[1:])
    # (capacity-sized text buffers — a base per frame, ~18 x the text: 160 MB for the 10 000-pair job.  Fresh arrays would be
    #  page-faulted in while the engine copies results out and unmapped on return, ~10 ms each way; they are kept per thread)
    seq1d = _scratch("seq1d", max(int(s1o[-1]), 1))
    seq = _scratch("seq", max(int(so[-1]), 1))
    l1, l2, lens, st = (np.zeros(n, dtype=np.int32) for _ in range(4))
    ident = np.zeros(n, dtype=np.float64)
    env = np.zeros((max(int(r1.sum()), 1), 2), dtype=np.int32) if return_envelope else None
    pm1 = (C.c_int * Cc)(*perm1) if perm1 is not None else None
    pm2 = (C.c_int * Cc)(*perm2) if perm2 is not None else None
    multi = devices is not None and len(devices) > 1
    pl = _multi(devices, wave_pairs, wave_rows, threads) if multi else _pipeline(
        wave_pairs, wave_rows, threads, device=(devices[0] if devices else None))
    _t1 = _time.perf_counter()
    fn = lib.po_multi_pair_decode if multi else lib.po_pipeline_pair_decode
    what = "po_multi_pair_decode" if multi else "po_pipeline_pair_decode"

    def call():
        return fn(pl, _ptr(p1), _ptr(r1), _ptr(p2), _ptr(r2), n, Cc, mode, pm1, pm2, 1 if reverse2 else 0,
                  C.byref(opt), _ptr(seq1d), _ptr(s1o), _ptr(l1), _ptr(l2), _ptr(ident), _ptr(env),
                  _ptr(seq), _ptr(so), _ptr(lens), _ptr(st))
    # A job of several waves: the records of a finished wave are built while the later ones decode.  The engine writes a
    # pair's status LAST (after its strings, behind a release fence) and never writes _PENDING, so a status that has
    # changed means the pair's outputs are there.  The engine call runs on a helper thread (ctypes drops the GIL).
    overlap = n > 4096 and not multi and os.environ.get("PO_NO_OVERLAP_RECORDS") is None
    out = []
    raw1, raw = memoryview(seq1d), memoryview(seq)     # (the buffers are capacity-sized, ~18 x the text: no bulk copy)
    eo = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(r1, out=eo[1:])
    s1l, sol = s1o.tolist(), so.tolist()
    skip_len, ok_codes = L.SKIP_LENGTH, (0, L.SKIP_LENGTH, L.SKIP_IDENTITY)

    def records(lo, hi):
        l1l, l2l, lnl, stl, idl = l1[lo:hi].tolist(), l2[lo:hi].tolist(), lens[lo:hi].tolist(), st[lo:hi].tolist(), ident[lo:hi].tolist()
        for k in range(hi - lo):
            i = lo + k
            code = stl[k]
            if strict and code not in ok_codes:
                raise L.EngineError(code, "pair decode of pair %d" % i)
            b1, b2, b = s1l[2 * i], s1l[2 * i + 1], sol[i]
            out.append({
                "seq1": str(raw1[b1:b1 + l1l[k]], "ascii"), "seq2": str(raw1[b2:b2 + l2l[k]], "ascii"),
                "consensus": str(raw[b:b + lnl[k]], "ascii") if code == 0 else None,
                "length1": l1l[k], "length2": l2l[k],
                "sequence_identity": idl[k] if code != skip_len else None,
                "skipped": 0 if code == 0 else 1, "status": code,
                "envelope": env[eo[i]:eo[i + 1]].astype(np.int64) if (code == 0 and return_envelope) else None})

    done = 0
    # one engine call at a time per pipeline: the cached pipeline (slots, staging buffers) is shared by every caller of
    # this process that asks for the same geometry
    plock = _pipeline_lock(pl)
    plock.acquire()
    try:
        if overlap:
            import threading
            st.fill(_PENDING)
            box = []

            def run():   # (the engine keeps its error text per thread: read it where it was written)
                rc = call()
                box.append((rc, (lib.po_last_error() or b"").decode() if rc != L.OK else ""))
            th = threading.Thread(target=run)
            th.start()
            err = None
            try:
                while th.is_alive():
                    seg = st[done:]
                    pend = np.flatnonzero(seg == _PENDING)
                    k = int(pend[0]) if len(pend) else len(seg)
                    if k == 0 or err is not None:
                        _time.sleep(0.001)
                        continue
                    try:
                        records(done, done + k)
                    except L.EngineError as e:      # (strict: raised once the engine call has returned)
                        err = e
                    done += k
            finally:
                # whatever ends the loop (KeyboardInterrupt, MemoryError, a decode error in records()): the engine call is
                # still writing into this thread's buffers and driving the cached pipeline — wait for it before they can
                # be handed to another call
                th.join()
            rc, detail = box[0] if box else (L.E_HIP, "the engine call did not return")
            if rc != L.OK:
                raise L.EngineError(rc, what, detail)
            if err is not None:
                raise err
        else:
            L.check(call(), what)
    finally:
        plock.release()
    if stats is not None:
        pk, wt, tot, wv, np_ = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_int()
        if multi:
            per = []
            for i in range(len(devices)):
                lib.po_multi_stats(pl, i, C.byref(np_), C.byref(pk), C.byref(wt), C.byref(tot), C.byref(wv))
                per.append({"device": int(devices[i]), "pairs": np_.value, "pack_ms": pk.value, "wait_ms": wt.value,
                            "total_ms": tot.value, "waves": wv.value})
            stats.update(per_device=per, waves=sum(d["waves"] for d in per), pack_ms=max(d["pack_ms"] for d in per),
                         wait_ms=max(d["wait_ms"] for d in per), total_ms=max(d["total_ms"] for d in per))
        else:
            lib.po_pipeline_stats(pl, C.byref(pk), C.byref(wt), C.byref(tot), C.byref(wv))
            stats.update(pack_ms=pk.value, wait_ms=wt.value, total_ms=tot.value, waves=wv.value)
    _t2 = _time.perf_counter()
    records(done, n)
    if stats is not None:
        stats.update(py_in_ms=(_t1 - _t0) * 1e3, call_ms=(_t2 - _t1) * 1e3, py_out_ms=(_time.perf_counter() - _t2) * 1e3)
    del raw1, raw
    release_scratch(_SCRATCH_KEEP)   # (a thread keeps its text buffers between calls up to this size; one huge job does not pin them)
    return out


def pair_decode_batch_sharded(arrays1, arrays2, devices=None, keep_envelope=True, single="viterbi", **kw):
    """pair_decode_batch over several GPUs of one node (BASELINE config 4; the reference's Pool fan-out,
    pair_decode.py:292-297), IN THIS PROCESS: one pipeline and one host thread per device inside
    po_multi_pair_decode, waves of pairs dealt to whichever device is free, inputs uploaded as they are (float32
    logits / uint8 traces / float64 log-probabilities: no packed copy, no shared memory, no worker processes),
    results written in input order.  devices: list of device indices (default: every visible device; an index may
    repeat); with one device this is the single-device pipeline.  single="beam" (1-D beam search basecalls first) is
    not a pipeline stage: it runs pair_decode_batch on the first device."""
    from . import dist as podist
    n = len(arrays1)
    devs = podist.plan_devices(n, devices)
    if single != "viterbi":
        if devs and devs[0] != 0:
            L.set_device(devs[0])
        return pair_decode_batch(arrays1, arrays2, single=single, **kw)
    return pair_decode_stream(arrays1, arrays2, return_envelope=keep_envelope, devices=(devs if len(devs) > 1 else None) or
                              (devs[:1] if devs else None), **kw)


def _pack_labels(labels):
    enc = [l.encode("ascii") for l in labels]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    buf = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
    return buf, off


def forward_batch(arrays, labels, alphabet="ACGT", model="ctc"):
    """decoding_cpp.cpp_forward for a batch: log P(label_i | y_i)."""
    lib = L.load()
    y, off, Cc = pack_rows(arrays)
    n = len(arrays)
    lb, lo = _pack_labels(labels)
    out = np.zeros(max(n, 1), dtype=np.float64)
    st = np.zeros(max(n, 1), dtype=np.int32)
    L.check(lib.po_forward_batch_h(_ptr(y), _ptr(off), n, Cc, alphabet.encode(), L.MODELS[model], _ptr(lb), _ptr(lo),
                                   _ptr(out), _ptr(st)), "po_forward_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "forward of item %d" % i)
    return out[:n].copy()


def viterbi_acceptor_batch(arrays, labels, band_size=1000, alphabet="ACGT", flavor="cpp"):
    """decoding_cpp.cpp_viterbi_acceptor (flavor "cpp") / decoding_cy.viterbi_acceptor ("cy") for a batch:
    per-frame state paths (blank = len(alphabet))."""
    lib = L.load()
    y, off, Cc = pack_rows(arrays)
    n = len(arrays)
    lb, lo = _pack_labels(labels)
    path = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    fn = lib.po_viterbi_acceptor_cy_batch_h if flavor == "cy" else lib.po_viterbi_acceptor_batch_h
    L.check(fn(_ptr(y), _ptr(off), n, Cc, alphabet.encode(), int(band_size), _ptr(lb), _ptr(lo), _ptr(path), _ptr(st)),
            "po_viterbi_acceptor_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "viterbi acceptor of item %d" % i)
    return [path[off[i]:off[i + 1]].astype(np.int64) for i in range(n)]


def prefix_search_batch(y, offsets, alphabet="ACGT"):
    """prefix_search.prefix_search_log_cy on every row range [offsets[i], offsets[i+1]) of ONE (T, C)
    matrix (so consecutive windows of a read need no copies).  Returns [(label, logp), ...]."""
    lib = L.load()
    y = np.ascontiguousarray(y, dtype=np.float64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(off) - 1
    so = off - off[0]
    seq = np.zeros(max(int(so[-1]), 1), dtype=np.uint8)
    lens = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    lp = np.zeros(max(n, 1), dtype=np.float64)
    L.check(lib.po_prefix_search_batch_h(_ptr(y), _ptr(off), n, y.shape[1], alphabet.encode(), _ptr(seq), _ptr(so),
                                         _ptr(lens), _ptr(lp), _ptr(st)), "po_prefix_search_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "prefix search of window %d" % i)
    return list(zip(_strings(seq, so, lens), [float(x) for x in lp[:n]]))


def forward_vec_batch(arrays, s, i, previous=None, flavor="cy"):
    """decoding_cy.forward_vec_log (flavor "cy") / prefix_search.forward_vec_log ("py") for a batch: one forward row
    per item.  previous: list of rows (one per item) of the label without its last symbol; None for i == 0."""
    lib = L.load()
    y, off, Cc = pack_rows(arrays)
    n = len(arrays)
    out = np.zeros(max(int(off[-1]), 1), dtype=np.float64)
    pv = None
    if previous is not None:
        pv = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64) for p in previous] + [np.zeros(1)]))
    L.check(lib.po_forward_vec_batch_h(_ptr(y), _ptr(off), n, Cc, int(s), int(i), {"py": 0, "cy": 1}[flavor],
                                       _ptr(pv), _ptr(out)), "po_forward_vec_batch_h")
    return [out[off[k]:off[k + 1]].copy() for k in range(n)]


def pair_prefix_search_batch(arrays1, arrays2, alphabet="ACGT", flavor="cy", envelopes=None):
    """prefix_search.pair_prefix_search_log_cy (flavor "cy") / pair_prefix_search_log ("py") for a batch of
    small boxes (dense gamma on the device).  envelopes: list of (U_i + 1, 2) arrays with INCLUSIVE column ends
    (Gamma.h) — gamma then comes from the envelope DP, the working form of decoding_cpp.cpp_pair_prefix_search_log
    (PairPrefixSearch.cpp:79-229).  Returns [(label, log-probability), ...]."""
    lib = L.load()
    y1, o1, Cc = pack_rows(arrays1)
    y2, o2, _ = pack_rows(arrays2, Cc)
    n = len(arrays1)
    env = eo = None
    if envelopes is not None:
        es = [np.ascontiguousarray(e, dtype=np.int32) for e in envelopes]
        for e, a in zip(es, arrays1):
            if e.ndim != 2 or e.shape[1] != 2 or e.shape[0] < len(a) + 1:
                raise ValueError("gamma envelopes need U + 1 rows")
        env = np.ascontiguousarray(np.concatenate([e[:len(a) + 1] for e, a in zip(es, arrays1)], axis=0))
        eo = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(a) + 1 for a in arrays1], out=eo[1:])
    so = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([max(len(a), len(b)) + 2 for a, b in zip(arrays1, arrays2)], out=so[1:])
    seq = np.zeros(max(int(so[-1]), 1), dtype=np.uint8)
    lens = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    lp = np.zeros(max(n, 1), dtype=np.float64)
    L.check(lib.po_pair_prefix_search_env_batch_h(_ptr(y1), _ptr(o1), _ptr(y2), _ptr(o2), _ptr(env), _ptr(eo), n, Cc,
                                                  alphabet.encode(), {"py": 0, "cy": 1}[flavor], _ptr(seq), _ptr(so),
                                                  _ptr(lens), _ptr(lp), _ptr(st)), "po_pair_prefix_search_env_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "pair prefix search of box %d" % i)
    return list(zip(_strings(seq, so, lens), [float(x) for x in lp[:n]]))


def align_batch(pairs, band_width=500, match=2, mismatch=-1, gap_cost=-1):
    """align.global_pair_banded (band_width > 0) / align.global_pair (band_width <= 0) for a batch of
    (seq1, seq2) string pairs.  Returns [(align1, align2), ...] as strings of equal length."""
    lib = L.load()
    n = len(pairs)
    enc = []
    for a, b in pairs:
        enc += [a.encode("ascii"), b.encode("ascii")]
    so = np.zeros(2 * n + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=so[1:])
    buf = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
    ao = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(a) + len(b) + 8 for a, b in pairs], out=ao[1:])
    a1 = np.zeros(max(int(ao[-1]), 1), dtype=np.uint8)
    a2 = np.zeros(max(int(ao[-1]), 1), dtype=np.uint8)
    nc = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    L.check(lib.po_align_scores_batch_h(_ptr(buf), _ptr(so), n, int(band_width), int(match), int(mismatch), int(gap_cost),
                                        _ptr(a1), _ptr(a2), _ptr(ao), _ptr(nc), _ptr(st)), "po_align_scores_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "alignment of pair %d" % i)
    return list(zip(_strings(a1, ao, nc), _strings(a2, ao, nc)))


def nw_matrix_batch(pairs, match=2, mismatch=-1, gap_cost=-1):
    """the dense DP matrix align.global_pair returns as its third item (align.pyx:34-52,98), for a batch of (seq1, seq2)
    string pairs: a list of (len1 + 1, len2 + 1) int32 arrays"""
    lib = L.load()
    n = len(pairs)
    enc = []
    for a, b in pairs:
        enc += [a.encode("ascii"), b.encode("ascii")]
    so = np.zeros(2 * n + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=so[1:])
    buf = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
    do = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([(len(a) + 1) * (len(b) + 1) for a, b in pairs], out=do[1:])
    dp = np.zeros(max(int(do[-1]), 1), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    L.check(lib.po_nw_matrix_batch_h(_ptr(buf), _ptr(so), n, int(match), int(mismatch), int(gap_cost), _ptr(dp), _ptr(do), _ptr(st)),
            "po_nw_matrix_batch_h")
    return [dp[do[i]:do[i + 1]].reshape(len(a) + 1, len(b) + 1) for i, (a, b) in enumerate(pairs)]


def envelope_batch(alignments, maps1, maps2, Us, Vs, padding=150):
    """envelope.build_envelope for a batch: alignments = [(row1, row2) strings], maps = frame index of every
    base (get_sequence_mapping), Us / Vs = signal lengths.  Returns a list of (U_i, 2) int arrays."""
    lib = L.load()
    n = len(alignments)
    ao = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(a) for a, _ in alignments], out=ao[1:])
    a1 = np.frombuffer(("".join(a for a, _ in alignments)).encode("ascii") + b"\0", dtype=np.uint8).copy()
    a2 = np.frombuffer(("".join(b for _, b in alignments)).encode("ascii") + b"\0", dtype=np.uint8).copy()
    nc = np.array([len(a) for a, _ in alignments] or [0], dtype=np.int32)
    m1o = np.zeros(n + 1, dtype=np.int64); np.cumsum([len(m) for m in maps1], out=m1o[1:])
    m2o = np.zeros(n + 1, dtype=np.int64); np.cumsum([len(m) for m in maps2], out=m2o[1:])
    m1 = np.ascontiguousarray(np.concatenate([np.asarray(m, dtype=np.int32) for m in maps1] + [np.zeros(1, np.int32)]))
    m2 = np.ascontiguousarray(np.concatenate([np.asarray(m, dtype=np.int32) for m in maps2] + [np.zeros(1, np.int32)]))
    U = np.array(list(Us) or [0], dtype=np.int32)
    V = np.array(list(Vs) or [0], dtype=np.int32)
    eo = np.zeros(n + 1, dtype=np.int64); np.cumsum(list(Us), out=eo[1:])
    env = np.zeros((max(int(eo[-1]), 1), 2), dtype=np.int32)
    st = np.zeros(max(n, 1), dtype=np.int32)
    L.check(lib.po_envelope_batch_h(_ptr(a1), _ptr(a2), _ptr(ao), _ptr(nc), n, _ptr(m1), _ptr(m1o), _ptr(m2), _ptr(m2o),
                                    _ptr(U), _ptr(V), int(padding), _ptr(env), _ptr(eo), _ptr(st)), "po_envelope_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "envelope of pair %d" % i)
    return [env[eo[i]:eo[i + 1]].astype(np.int64) for i in range(n)]


def ingest_batch(arrays, perm=None, reverse=False):
    """Device ingest of basecaller outputs -> list of (T, C) float64 log-probability matrices.
    float32 (T, C) logits -> log-softmax (decode.py:34-39); uint8 traces -> log((x+1e-7)/(255+1e-7))
    (decode.py:92); float64 -> copied.  perm: column order (Bonito: [1,2,3,4,0]); reverse: time-reverse
    every item (reverse_complement = reverse + perm [3,2,1,0,4])."""
    lib = L.load()
    if not arrays:
        return []
    dt = arrays[0].dtype
    mode = {np.dtype(np.float32): 0, np.dtype(np.uint8): 1, np.dtype(np.float64): 2}.get(np.dtype(dt))
    if mode is None or any(a.dtype != dt or a.ndim != 2 for a in arrays):
        raise ValueError("ingest_batch takes 2-D float32 logits, uint8 traces or float64 matrices of one dtype")
    Cc = arrays[0].shape[1]
    off = np.zeros(len(arrays) + 1, dtype=np.int64)
    np.cumsum([len(a) for a in arrays], out=off[1:])
    src = np.ascontiguousarray(np.concatenate(arrays, axis=0))
    out = np.zeros((int(off[-1]), Cc), dtype=np.float64)
    p = (C.c_int * Cc)(*perm) if perm is not None else None
    L.check(lib.po_ingest_batch_h(_ptr(src), _ptr(off), len(arrays), Cc, mode, p, 1 if reverse else 0, _ptr(out)),
            "po_ingest_batch_h")
    return [out[off[i]:off[i + 1]] for i in range(len(arrays))]


def pair_gamma_batch(arrays1, arrays2, envelopes=None, flavor="cpp", return_matrix=False):
    """gamma(0,0) = log P(both reads emit the same label) for a batch of pairs.
    envelopes: list of (U_i + 1, 2) arrays with INCLUSIVE ends (Gamma.h), or None for the dense DP.
    flavor "cpp" = Gamma.h arithmetic, "cy" = decoding_cy.pair_gamma_log arithmetic (dense), "cy_env" =
    decoding_cy.pair_gamma_log_envelope (log(exp + exp), -inf defaults, every envelope cell with u < U, v < V computed).
    return_matrix: return the (U+1, V+1) gamma matrices instead of gamma(0,0) (-inf outside an envelope)."""
    lib = L.load()
    y1, o1, Cc = pack_rows(arrays1)
    y2, o2, _ = pack_rows(arrays2, Cc)
    n = len(arrays1)
    env = eo = None
    if envelopes is not None:
        es = [np.ascontiguousarray(e, dtype=np.int32) for e in envelopes]
        for e, a in zip(es, arrays1):
            if e.ndim != 2 or e.shape[1] != 2 or e.shape[0] < len(a) + 1:
                raise ValueError("gamma envelopes need U + 1 rows")
        env = np.ascontiguousarray(np.concatenate([e[:len(a) + 1] for e, a in zip(es, arrays1)], axis=0))
        eo = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(a) + 1 for a in arrays1], out=eo[1:])
    g0 = np.zeros(max(n, 1), dtype=np.float64)
    st = np.zeros(max(n, 1), dtype=np.int32)
    dn = dof = None
    if return_matrix:   # (with an envelope: -inf outside the stored ranges)
        dof = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([(len(a) + 1) * (len(b) + 1) for a, b in zip(arrays1, arrays2)], out=dof[1:])
        dn = np.zeros(max(int(dof[-1]), 1), dtype=np.float64)
    L.check(lib.po_pair_gamma_batch_h(_ptr(y1), _ptr(o1), _ptr(y2), _ptr(o2), _ptr(env), _ptr(eo), n, Cc,
                                      {"cpp": 0, "cy": 1, "cy_env": 2}[flavor], _ptr(g0), _ptr(dn), _ptr(dof), _ptr(st)),
            "po_pair_gamma_batch_h")
    for i in range(n):
        if st[i] != 0:
            raise L.EngineError(int(st[i]), "pair gamma of pair %d" % i)
    if return_matrix:
        return [dn[dof[i]:dof[i + 1]].reshape(len(arrays1[i]) + 1, len(arrays2[i]) + 1) for i in range(n)]
    return g0[:n].copy()
