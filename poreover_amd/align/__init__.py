"""Mirror of poreover.align (reference align/__init__.py:1, align/align.pyx): global pairwise alignment
of two basecalls, constant gap penalty, on the GPU engine."""
from .align import global_pair, global_pair_banded
