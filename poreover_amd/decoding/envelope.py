"""poreover.decoding.envelope (reference envelope.py:5-103) behind the same names.  build_envelope runs on the GPU
engine.  The small host helpers add_block, check_envelope, offset_envelope and pad_envelope are a few lines each and ARE
the reference's Python restated line for line (their behaviour — in-place edits, clamping, return values — is the
interface); none of them is on the decode path."""
import numpy as np

from .. import batch as _batch


def add_block(b, envelope):
    """envelope.py:5-17: add one (sx, sy, ex, ey) block to a row-based envelope, in place"""
    (sx, sy, ex, ey) = b
    for i in range(sx, ex):
        if i < len(envelope):
            if sy < envelope[i, 0] or envelope[i, 0] < 0:
                envelope[i, 0] = sy
            if ey > envelope[i, 1] or envelope[i, 1] < 0:
                envelope[i, 1] = ey


def check_envelope(envelope, U, V):
    """envelope.py:19-24"""
    check_greater = all(envelope[:, 1] > envelope[:, 0])
    check_overlap = all(envelope[:-1, 1] - envelope[1:, 0])
    check_length = len(envelope) == U + 2
    check_range = all(envelope[:, 1] <= V)
    return check_greater and check_overlap and check_length and check_range


def get_alignment_columns(alignment):
    """envelope.py:26-44: (label, x_index, y_index) per column of a 2 x N character array"""
    x_index, y_index, out = -1, -1, []
    for (x, y) in np.asarray(alignment).T:
        if x != '-':
            x_index += 1
        if y != '-':
            y_index += 1
        out.append(('i' if x == '-' else ('d' if y == '-' else 'm'), x_index, y_index))
    return out


def build_envelope(y1, y2, alignment_col, sequence_to_signal1, sequence_to_signal2, padding=150):
    """envelope.py:46-87.  alignment_col as returned by get_alignment_columns."""
    # only "does this column advance read 1 / read 2" matters, which the labels encode
    row1 = "".join('-' if lab == 'i' else 'N' for lab, _, _ in alignment_col)
    row2 = "".join('-' if lab == 'd' else 'N' for lab, _, _ in alignment_col)
    return _batch.envelope_batch([(row1, row2)], [sequence_to_signal1], [sequence_to_signal2], [len(y1)], [len(y2)],
                                 padding)[0]


def offset_envelope(full_envelope, subset):
    """envelope.py:89-94"""
    (u1, u2, v1, v2) = subset
    subset_envelope = np.copy(full_envelope[u1:u2])
    subset_envelope[:, 0] = subset_envelope[:, 0] - v1
    subset_envelope[:, 1] = subset_envelope[:, 1] - v1
    return subset_envelope


def pad_envelope(envelope, U, V):
    """envelope.py:96-103"""
    new_envelope = np.concatenate((envelope, [envelope[-1], envelope[-1]]))
    for i, _ in enumerate(new_envelope):
        if new_envelope[i, 1] == V - 1:
            new_envelope[i, 1] = V
    new_envelope[U] = new_envelope[U - 1]
    new_envelope[U + 1] = new_envelope[U - 1]
    return new_envelope
