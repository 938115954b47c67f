"""Mirror of poreover.decoding (reference decoding/__init__.py:1-4) backed by the HIP engine."""
from . import decoding_cpp, decoding_cy, transducer, decode, pair_decode, prefix_search, envelope
from .decoding_cpp import (cpp_beam_search, cpp_beam_search_2d, cpp_forward, cpp_viterbi_acceptor,
                           cpp_pair_gamma_log_envelope, cpp_pair_prefix_search_log)
