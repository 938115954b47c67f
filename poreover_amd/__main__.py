"""`python -m poreover_amd decode|pair-decode ...` — the decode / pair-decode sub-commands of the
reference CLI (reference __main__.py:52-91) with the same flags and defaults, on the GPU engine.
(`train`, `call` and `benchmark` are outside this engine's scope: SURVEY.md §2.)"""
import argparse
import logging
import sys

from . import __version__


def build_parser():
    parser = argparse.ArgumentParser(prog="poreover_amd",
                                     description='PoreOver decoding on MI355X: consensus basecalling for nanopore sequencing')
    subparsers = parser.add_subparsers(dest="command")
    subparsers.required = True

    p = subparsers.add_parser('decode', help='Decode basecaller probabilities to a FASTA file')
    p.add_argument('in', nargs='+', help='Probabilities to decode (.npy from PoreOver/Bonito, .csv, or HDF5/FAST5 from Flappie/Guppy)')
    p.add_argument('--out', default='out', help='Prefix for FASTA sequence output')
    p.add_argument('--basecaller', choices=['poreover', 'flappie', 'guppy', 'bonito'], help='Basecaller used to generate probabilities')
    p.add_argument('--algorithm', default='viterbi', choices=['viterbi', 'beam', 'prefix'], help='')
    p.add_argument('--window', type=int, default=400, help='Use chunks of this size for prefix search')
    p.add_argument('--beam_width', type=int, default=25, help='Width for beam search')
    p.add_argument('--threads', type=int, default=1, help='Upper bound on the GPUs one call is spread over when > 1 (the reference: worker processes); batching replaces processes on each device')
    p.add_argument('-v', '--version', action='version', version=__version__)
    p.set_defaults(func="decode")

    p = subparsers.add_parser('pair-decode', help='1D2 consensus decoding of two output probabilities',
                              formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-v', '--version', action='version', version=__version__)
    p.add_argument('in', nargs='+', help='Softmax probabilities to decode or list of read pairs')
    p.add_argument('--dir', default='.', help='Base directory to look in for basecaller probabilities')
    p.add_argument('--basecaller', choices=['poreover', 'flappie', 'guppy', 'bonito'], help='Basecaller used to generate probabilities')
    p.add_argument('--reverse_complement', default=False, action='store_true', help='Whether to reverse complement the second sequence')
    p.add_argument('--out', default='out', help='Prefix for FASTA sequence output')
    p.add_argument('--threads', type=int, default=1, help='Upper bound on the GPUs one call is spread over when > 1 (the reference: worker processes); batching replaces processes on each device')
    p.add_argument('--method', choices=['align', 'split', 'envelope'], default='envelope', help=argparse.SUPPRESS)
    p.add_argument('--single', choices=['beam', 'viterbi'], default='viterbi', help='Algorithm for 1D basecalling (used to build alignment envelope)')
    p.add_argument('--logging', default="info", choices=['info', 'debug'], help='Level for logging')
    p.add_argument('--debug', default=False, action='store_true', help=argparse.SUPPRESS)
    p.add_argument('--algorithm', default='beam', choices=['prefix', 'beam'], help=argparse.SUPPRESS)
    p.add_argument('--alignment', default='banded', choices=['banded', 'full'], help='Do full Needleman-Wunsch alignment between 1D basecalls to build envelope')
    p.add_argument('--beam_width', type=int, default=5, help='Width for beam search')
    p.add_argument('--debug_envelope', action='store_true', help=argparse.SUPPRESS)
    p.add_argument('--diagonal_envelope', action='store_true', help='Use a simple diagonal band for the signal alignment envelope')
    p.add_argument('--diagonal_width', type=int, default=50, help='Width of diagonal band envelope')
    p.add_argument('--padding', type=int, default=5, help='Padding for building alignment envelope')
    p.add_argument('--skip_matches', action='store_true', help='Skip regions of sequence alignment with match columns greater than --skip_threshold')
    p.add_argument('--skip_threshold', type=int, default=10, help='Number of consecutive matches to use for --skip_matches')
    p.add_argument('--beam_search_method', choices=['row', 'row_col', 'grid'], default="row_col", help=argparse.SUPPRESS)
    p.add_argument('--window', type=int, default=200, help=argparse.SUPPRESS)
    p.set_defaults(func="pair-decode")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(format='%(message)s', level=logging.DEBUG if getattr(args, 'logging', 'info') == 'debug' else logging.INFO)
    from .decoding import decode as _decode, pair_decode as _pair
    if args.func == "decode":
        _decode.decode(args)
    else:
        _pair.pair_decode(args)
    print(args, file=sys.stderr)


if __name__ == "__main__":
    main()
