"""Mirror of poreover.decoding.prefix_search (reference prefix_search.py): the 1-D prefix search
entry points, on the GPU engine."""
from collections import OrderedDict

import numpy as np

from .. import batch as _batch

DNA_alphabet = OrderedDict([('A', 0), ('C', 1), ('G', 2), ('T', 3)])


def remove_gaps(a):
    """prefix_search.py:16-23"""
    return ''.join(i for i in a if i != '-')


def greedy_search(logits, alphabet=['A', 'C', 'G', 'T', '-']):
    """prefix_search.py:25-29: best symbol per frame, gaps removed (repeats kept)"""
    sym = "".join(a for a in alphabet if a != '-')
    return _batch.viterbi_batch([np.asarray(logits, dtype=np.float64)], "poreover", sym)[0]


def prefix_search_log_cy(y_, alphabet=DNA_alphabet, return_forward=False):
    """prefix_search.py:176-238 -> (label, log-probability of the label), or with return_forward (label, matrix):
    the transposed rows [0, len(label)) of upstream's per-symbol table prefix_forward[last symbol] — row j is the
    forward row of label[:j] + label[-1] (the row written for that symbol at search level j + 1)."""
    y = np.asarray(y_, dtype=np.float64)
    sym = "".join(alphabet.keys())
    label, logp = _batch.prefix_search_batch(y, [0, len(y)], sym)[0]
    if not return_forward:
        return label, logp
    if not label:
        return label, np.array([]).T
    last = alphabet[label[-1]]
    prev = _batch.forward_vec_batch([y], -1, 0, None, "cy")[0]
    rows = []
    for j in range(len(label)):
        rows.append(_batch.forward_vec_batch([y], last, j + 1, [prev], "cy")[0])
        if j + 1 < len(label):
            prev = _batch.forward_vec_batch([y], alphabet[label[j]], j + 1, [prev], "cy")[0]
    return label, np.array(rows).T


def _env_arg(envelope_ranges):
    return None if envelope_ranges is None else [np.asarray(envelope_ranges, dtype=np.int32)]


def pair_prefix_search_log_cy(y1_, y2_, alphabet=DNA_alphabet, envelope_ranges=None):
    """prefix_search.py:316-385 -> (label, log-probability): pair prefix search over the gamma matrix of two small
    boxes (decoding_cy arithmetic).  envelope_ranges ((U + 1, 2), inclusive column ends): gamma restricted to the
    envelope (Gamma.h:15-98) — what PairPrefixSearch.cpp:79-229 set out to do."""
    sym = "".join(alphabet.keys())
    return _batch.pair_prefix_search_batch([np.asarray(y1_, dtype=np.float64)], [np.asarray(y2_, dtype=np.float64)], sym, "cy",
                                           _env_arg(envelope_ranges))[0]


def pair_prefix_search_log(y1_, y2_, alphabet=DNA_alphabet, envelope_ranges=None):
    """prefix_search.py:247-314: the numpy twin (-inf instead of LOG_0, np.logaddexp); envelope_ranges as above"""
    sym = "".join(alphabet.keys())
    return _batch.pair_prefix_search_batch([np.asarray(y1_, dtype=np.float64)], [np.asarray(y2_, dtype=np.float64)], sym, "py",
                                           _env_arg(envelope_ranges))[0]


prefix_search_log = prefix_search_log_cy   # the pure-python twin computes the same quantity (prefix_search.py:115)
