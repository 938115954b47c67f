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
    """prefix_search.py:176-238 -> (label, log-probability of the label)"""
    if return_forward:
        raise NotImplementedError("return_forward is only used by the reference's deprecated box methods")
    y = np.asarray(y_, dtype=np.float64)
    sym = "".join(alphabet.keys())
    return _batch.prefix_search_batch(y, [0, len(y)], sym)[0]


def pair_prefix_search_log_cy(y1_, y2_, alphabet=DNA_alphabet):
    """prefix_search.py:316-385 -> (label, log-probability): pair prefix search over the dense gamma matrix
    of two small boxes (decoding_cy arithmetic)"""
    sym = "".join(alphabet.keys())
    return _batch.pair_prefix_search_batch([np.asarray(y1_, dtype=np.float64)], [np.asarray(y2_, dtype=np.float64)], sym, "cy")[0]


def pair_prefix_search_log(y1_, y2_, alphabet=DNA_alphabet):
    """prefix_search.py:247-314: the numpy twin (-inf instead of LOG_0, np.logaddexp)"""
    sym = "".join(alphabet.keys())
    return _batch.pair_prefix_search_batch([np.asarray(y1_, dtype=np.float64)], [np.asarray(y2_, dtype=np.float64)], sym, "py")[0]


prefix_search_log = prefix_search_log_cy   # the pure-python twin computes the same quantity (prefix_search.py:115)
