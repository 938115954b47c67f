"""Mirror of poreover.decoding.decode (reference decode.py:20-192): trace loading, FASTA formatting
and the `decode` driver.  Where the reference starts one process per file (decode.py:158-162) this
driver loads every trace and makes ONE batched engine call."""
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
    try:
        import h5py
    except ImportError as e:  # same dependency as the reference (decode.py:2)
        raise ImportError("reading %s needs h5py, as in the reference" % path) from e
    with h5py.File(path, 'r') as hdf:
        if dataset is None:
            read_id = list(hdf)[0]
            return np.array(hdf[read_id]['trace'])
        return np.array(hdf[dataset])


def model_from_trace(f, basecaller=""):
    """decode.py:67-112"""
    _, ext = os.path.splitext(f)
    if ext == '.npy' and basecaller == 'poreover':
        return transducer.poreover(load_logits(f, flatten=True))
    if ext == '.npy' and basecaller == 'bonito':
        trace = load_logits(f, flatten=True)
        return transducer.bonito(trace[::, [1, 2, 3, 4, 0]])
    if ext == '.csv':
        trace = np.log(np.loadtxt(f, delimiter=',', skiprows=1))
        if trace.shape[1] == 5:
            return transducer.poreover(trace)
        if trace.shape[1] == 8:
            return transducer.flipflop(trace)
    if ext == '.hdf5' or basecaller == 'flappie':
        eps = 0.0000001
        return transducer.flipflop(np.log((_trace_from_hdf5(f, None) + eps) / (255 + eps)))
    if ext == '.fast5' or basecaller == 'guppy':
        eps = 0.0000001
        trace = _trace_from_hdf5(f, '/Analyses/Basecall_1D_000/BaseCalled_template/Trace')
        return transducer.flipflop(np.log((trace + eps) / (255 + eps)))
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
    for kind, idx in by_kind.items():
        ys = [models[i].log_prob for i in idx]
        if args.algorithm == 'viterbi':
            seqs = _batch.viterbi_batch(ys, kind)
        elif args.algorithm == 'beam':
            seqs = _batch.beam_search_batch(ys, args.beam_width, "ACGT", MODEL_TYPE[kind])
        elif args.algorithm == 'prefix':
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
        models = [model_from_trace(p, args.basecaller) for p in in_files]
        seqs = decode_models(models, args)
        with open(args.out + '.fasta', 'w') as out_f:
            for p, s in zip(in_files, seqs):
                print(fasta_format(Path(p).stem, s), file=out_f)
    else:
        seqs = decode_helper(in_files[0], args)
        with open(args.out + '.fasta', 'w') as out_fasta:
            print(seqs, file=out_fasta)
