"""align.global_pair / align.global_pair_banded (reference align/align.pyx:29-178)."""
from .. import batch as _batch

MATCH_DEFAULT, MISMATCH_DEFAULT, GAP_DEFAULT, BAND_DEFAULT = 2, -1, -1, 500


def global_pair(seq1, seq2, match=MATCH_DEFAULT, mismatch=MISMATCH_DEFAULT, gap_cost=GAP_DEFAULT):
    """Needleman-Wunsch (align.pyx:29-98).  Returns (align1, align2, dpMatrix) as the reference does: two lists of
    characters and the dense (len1 + 1, len2 + 1) int32 DP matrix."""
    a1, a2 = _batch.align_batch([(seq1, seq2)], 0, match, mismatch, gap_cost)[0]
    return list(a1), list(a2), _batch.nw_matrix_batch([(seq1, seq2)], match, mismatch, gap_cost)[0]


def global_pair_banded(seq1, seq2, band_width=BAND_DEFAULT, match=MATCH_DEFAULT, mismatch=MISMATCH_DEFAULT,
                       gap_cost=GAP_DEFAULT):
    """Banded Needleman-Wunsch exactly as written upstream (align.pyx:100-178)."""
    if band_width <= 0:
        raise ValueError("band_width must be positive")
    a1, a2 = _batch.align_batch([(seq1, seq2)], band_width, match, mismatch, gap_cost)[0]
    return list(a1), list(a2)
