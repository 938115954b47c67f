"""poreover.decoding.decode (reference decode.py:20-192) behind the same names: trace loading, FASTA formatting and
the `decode` driver.  Where the reference starts one process per file (decode.py:158-162) this driver loads every
trace and makes ONE batched engine call per input form (spread over the node's GPUs when there are several).
fasta_format, softmax, logit_to_log_likelihood and load_logits are short host helpers that restate the reference's
Python line for line — the output format and the host arithmetic are the interface (the batched drivers do the
log-softmax on the device instead; load_logits serves `.log_prob` and probability-valued files)."""
import glob
import logging
import os
import sys
from pathlib import Path

import numpy as np

from .. import batch as _batch
from .. import _lib
from . import transducer

MODEL_TYPE = {'poreover': 'ctc', 'bonito': 'ctc_merge_repeats', 'guppy': 'ctc_flipflop',
              'flappie': 'ctc_flipflop', 'flipflop': 'ctc_flipflop'}


def fasta_format(name, seq, width=60):
    """decode.py:20-27"""
    fasta = '>' + name + '\n'
    window = 0
    while window + width < len(seq):
        fasta += (seq[window:window + width] + '\n')
        window += width
    fasta += (seq[window:] + '\n')
    return fasta


try:  # the reference calls scipy.special.logsumexp (decode.py:7,39); use the same function when present
    from scipy.special import logsumexp as _logsumexp
except ImportError:  # pragma: no cover
    def _logsumexp(x, axis):
        m = np.max(x, axis=axis, keepdims=True)
        m = np.where(np.isfinite(m), m, 0)
        return (np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)) + m).squeeze(axis)


def softmax(logits):
    e = np.exp(logits)
    return (e.T / np.sum(e, axis=len(logits.shape) - 1).T).T


def logit_to_log_likelihood(logits):
    """decode.py:34-39: logits - logsumexp(logits, axis=2) (the reference requires a 3-D array)"""
    return (logits.T - _logsumexp(logits, axis=2).T).T


def load_logits(file_path, flatten=False):
    """decode.py:41-51"""
    read_reshape = np.load(file_path)
    if np.isclose(np.sum(read_reshape[0]), 1):
        read_reshape = np.log(read_reshape)
    else:
        read_reshape = logit_to_log_likelihood(read_reshape)
    if flatten and len(read_reshape.shape) > 2:
        return np.concatenate(read_reshape)
    return read_reshape


def _trace_from_hdf5(path, dataset):
    """decode.py:53-65: the uint8 trace of a Flappie .hdf5 (first read's 'trace') or a Guppy .fast5 (the named
    dataset).  The reference opens the file with h5py; so does this function when h5py is installed, and otherwise it
    uses the engine's own reader of the part of the HDF5 format these files use (hdf5_lite)."""
    try:
        import h5py
        opener = h5py.File
    except ImportError:
        from . import hdf5_lite
        opener = hdf5_lite.File
    with opener(path, 'r') as hdf:
        if dataset is None:
            read_id = list(hdf)[0]
            return np.array(hdf[read_id]['trace'])
        return np.array(hdf[dataset])


def _deferred_logits(file_path):
    """load_logits(flatten=True) with the log-softmax left to the engine's ingest kernel when the file holds
    float32 logits of shape (B, W, C) (what `poreover call` and the Bonito patch write); probabilities, other
    dtypes and the shapes on which the reference raises go through load_logits itself."""
    arr = np.load(file_path)
    if arr.ndim == 3 and arr.dtype == np.float32 and not np.isclose(np.sum(arr[0]), 1):
        return transducer.DeferredTrace(arr.reshape(-1, arr.shape[2]), 0,
                                        lambda a, _s=arr.shape: np.concatenate(logit_to_log_likelihood(a.reshape(_s))))
    if np.isclose(np.sum(arr[0]), 1):
        arr = np.log(arr)
    else:
        arr = logit_to_log_likelihood(arr)
    return np.concatenate(arr) if len(arr.shape) > 2 else arr


def _deferred_trace(trace):
    """uint8 flip-flop trace -> log((x + eps) / (255 + eps)) (decode.py:89-93,99-103), on the device when it can be"""
    eps = 0.0000001
    trace = np.asarray(trace)
    if trace.dtype == np.uint8 and trace.ndim == 2:
        return transducer.DeferredTrace(trace, 1, lambda a: np.log((a + eps) / (255 + eps)))
    return np.log((trace + eps) / (255 + eps))


def model_from_trace(f, basecaller=""):
    """decode.py:67-112.  float32 logits and uint8 traces stay as they are inside the returned object (deferred
    ingest: the batched drivers upload them in that form); .log_prob gives the reference's float64 table."""
    _, ext = os.path.splitext(f)
    if ext == '.npy' and basecaller == 'poreover':
        return transducer.poreover(_deferred_logits(f))
    if ext == '.npy' and basecaller == 'bonito':
        model = transducer.bonito(_deferred_logits(f))
        model._permute([1, 2, 3, 4, 0])          # trace[::, [1, 2, 3, 4, 0]] (decode.py:79)
        return model
    if ext == '.csv':
        trace = np.log(np.loadtxt(f, delimiter=',', skiprows=1))
        if trace.shape[1] == 5:
            return transducer.poreover(trace)
        if trace.shape[1] == 8:
            return transducer.flipflop(trace)
    if ext == '.hdf5' or basecaller == 'flappie':
        return transducer.flipflop(_deferred_trace(_trace_from_hdf5(f, None)))
    if ext == '.fast5' or basecaller == 'guppy':
        return transducer.flipflop(_deferred_trace(_trace_from_hdf5(f, '/Analyses/Basecall_1D_000/BaseCalled_template/Trace')))
    if basecaller == "":
        print("Problem loading the trace probabilities, please specify where they came from with "
              "--basecaller [poreover/guppy/flappie]")
    else:
        print("Problem loading the trace probabilities")
    sys.exit(1)


def decode_models(models, args):
    """Decode a list of transducer objects with ONE engine call per model kind; returns sequences."""
    out = [None] * len(models)
    by_kind = {}
    for i, m in enumerate(models):
        by_kind.setdefault(m.kind, []).append(i)
    groups = {}
    for kind, idx in by_kind.items():      # reads that can share an engine call: same kind, input form, pending permutation
        for i in idx:
            e = models[i].engine_input()
            groups.setdefault((kind, e[1], tuple(e[2]), e[3]), []).append(i)
    for (kind, mode, perm, rev), idx in groups.items():
        if args.algorithm in ('viterbi', 'beam'):
            # the traces go up as the basecaller wrote them (deferred ingest: log-softmax / scaling on the device)
            ident = list(range(len(perm)))
            seqs = _batch.decode_1d_batch([models[i].engine_input()[0] for i in idx], kind, args.algorithm, args.beam_width,
                                          "ACGT", None if list(perm) == ident else list(perm), rev)
            for i, s in zip(idx, seqs):
                out[i] = s
            continue
        ys = [models[i].log_prob for i in idx]
        if args.algorithm == 'prefix':
            assert kind == "poreover"
            seqs = []
            for yy in ys:   # consecutive windows of args.window frames (decode.py:182-188): offsets, no copies
                window, t_max = args.window, len(yy)
                # the reference: while i + window < t_max: [i, i+window); then the rest [i, t_max)
                offs, i = [0], 0
                while i + window < t_max:
                    i += window
                    offs.append(i)
                offs.append(t_max)
                seqs.append("".join(lab for lab, _ in _batch.prefix_search_batch(yy, offs)))
        else:
            raise ValueError("unknown algorithm %r" % args.algorithm)
        for i, s in zip(idx, seqs):
            out[i] = s
    return out


def decode_files_local(in_files, args):
    """Sequences of a list of trace files, decoded on THIS process's device (one batched engine call per kind)."""
    return decode_models([model_from_trace(p, args.basecaller) for p in in_files], args)


def decode_files(in_files, args, devices=None, decode_fn=None):
    """decode.py:158-162's Pool fan-out, over GPUs: files are split by size over the node's devices (one spawned
    worker per device, or the ranks of a torchrun launch) and the sequences come back in input order (None on ranks
    other than 0 of a torchrun launch)."""
    from .. import dist as podist
    fn = decode_fn or decode_files_local
    rank, local_rank, world = podist.env_rank_world()
    costs = []
    for p in in_files:
        try:
            costs.append(max(1, os.path.getsize(p)))
        except OSError:
            costs.append(1)
    if world > 1:
        if decode_fn is None:
            _lib.set_device(local_rank)
        return podist.decode_distributed(in_files, costs, fn, args)
    threads = getattr(args, 'threads', 1)
    devs = podist.plan_devices(len(in_files), devices, threads if threads and threads > 1 else None)
    if len(devs) <= 1:
        if devs and devs[0] != 0 and decode_fn is None:
            _lib.set_device(devs[0])
        return fn(in_files, args)
    return podist.run_sharded(in_files, costs, fn, devs, args, bind_device=decode_fn is None)


def decode_helper(in_path, args):
    """decode.py:169-192 for one file"""
    model = model_from_trace(in_path, args.basecaller)
    return fasta_format(Path(in_path).stem, decode_models([model], args)[0])


def decode(args):
    """decode.py:114-167.  Output: {out}.fasta with one record per input, in INPUT order (the
    reference writes records in process-completion order)."""
    logger = logging.getLogger("poreover_amd")
    in_path = getattr(args, 'in')
    in_files = in_path
    if len(in_path) == 1 and os.path.isdir(in_path[0]):
        file_ext = {'guppy': '.fast5', 'flappie': '.hdf5', 'bonito': '.npy', 'poreover': '.npy'}[args.basecaller]
        in_files = sorted(glob.glob("{}/*{}".format(in_path[0], file_ext)))
    if len(in_files) > 1:
        logger.info("found {} reads to decode".format(len(in_files)))
        seqs = decode_files(in_files, args)
        if seqs is None:   # a rank other than 0 of a torchrun launch: rank 0 writes the file
            return
        with open(args.out + '.fasta', 'w') as out_f:
            for p, s in zip(in_files, seqs):
                print(fasta_format(Path(p).stem, s), file=out_f)
    else:
        seqs = decode_helper(in_files[0], args)
        with open(args.out + '.fasta', 'w') as out_fasta:
            print(seqs, file=out_fasta)
