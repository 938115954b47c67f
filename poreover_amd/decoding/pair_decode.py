"""Mirror of poreover.decoding.pair_decode (reference pair_decode.py:230-529): the 1D^2 consensus
driver.  The whole stage chain of pair_decode_helper — 1-D basecalls, alignment, skips, envelope,
pair beam search — runs on the GPU (batch.pair_decode_batch -> po_pair_decode_batch); this module
loads the traces, applies --reverse_complement, and writes the reference's output files."""
import logging
import os
import sys
from pathlib import Path

from .. import batch as _batch
from .. import _lib
from . import decode

fasta_format = decode.fasta_format


def _check_supported(args):
    if getattr(args, 'method', 'envelope') != 'envelope':
        raise _lib.EngineError(_lib.E_UNSUPPORTED, "pair-decode --method %s" % args.method,
                               "only the envelope method (the reference's default) is built")
    if getattr(args, 'algorithm', 'beam') != 'beam':
        raise _lib.EngineError(_lib.E_UNSUPPORTED, "pair-decode --algorithm %s" % args.algorithm)
    if getattr(args, 'skip_matches', False):
        raise _lib.EngineError(_lib.E_UNSUPPORTED, "pair-decode --skip_matches")


def _load_pair(in_path, args):
    path1, path2 = Path(in_path[0]), Path(in_path[1])
    if path1.suffix == ".fast5":   # pairs files may list FAST5 names (pair_decode.py:316-319)
        path1 = path1.with_suffix(".npy")
    if path2.suffix == ".fast5":
        path2 = path2.with_suffix(".npy")
    model1 = decode.model_from_trace(os.path.join(args.dir, path1), args.basecaller)
    model2 = decode.model_from_trace(os.path.join(args.dir, path2), args.basecaller)
    if args.reverse_complement:
        model2.reverse_complement()
    assert model1.kind == model2.kind
    return path1, path2, model1, model2


def decode_pairs(in_paths, args):
    """pair_decode_helper for a LIST of pairs: returns a list of the reference's return tuples
    (1-, 2- or 3-tuples, pair_decode.py:375,398,526-529)."""
    _check_supported(args)
    loaded = [_load_pair(p, args) for p in in_paths]
    out = [None] * len(loaded)
    by_kind = {}
    for i, (_, _, m1, _) in enumerate(loaded):
        by_kind.setdefault(m1.kind, []).append(i)
    for kind, idx in by_kind.items():
        res = _batch.pair_decode_batch([loaded[i][2].log_prob for i in idx], [loaded[i][3].log_prob for i in idx],
                                       kind=kind, beam_width=args.beam_width, method=args.beam_search_method,
                                       padding=args.padding, alignment=args.alignment,
                                       diagonal_envelope=args.diagonal_envelope,
                                       diagonal_width=args.diagonal_width, single=getattr(args, 'single', 'viterbi'))
        for i, r in zip(idx, res):
            in_path = in_paths[i]
            path1, path2 = loaded[i][0], loaded[i][1]
            if args.diagonal_envelope:
                # no 1-D decoding to return; the header quirk of pair_decode.py:527 is kept
                out[i] = (fasta_format('consensus;{};{}'.format(args.method, path1.stem, path2.stem), r["consensus"]),
                          {'read1': in_path[0], 'read2': in_path[1]})
                continue
            summary = {'read1': in_path[0], 'read2': in_path[1], 'length1': r["length1"], 'length2': r["length2"]}
            if r["status"] == _lib.SKIP_LENGTH:
                summary['skipped'] = 1
                out[i] = [summary]
                continue
            summary['sequence_identity'] = r["sequence_identity"]
            if r["status"] == _lib.SKIP_IDENTITY:
                summary['skipped'] = 1
                out[i] = [summary]
                continue
            summary['skipped'] = 0
            out[i] = (fasta_format(in_path[0], r["seq1"]) + fasta_format(in_path[1], r["seq2"]),
                      fasta_format('consensus;{};{}'.format(path1.stem, path2.stem), r["consensus"]), summary)
    return out


def pair_decode_helper(args):
    """pair_decode.py:305-529 for the one pair named by args.in"""
    in_path = getattr(args, 'in')
    if len(in_path) != 2:
        raise ValueError("Exactly two reads are required")
    return decode_pairs([in_path], args)[0]


def pair_decode(args):
    """pair_decode.py:230-303.  One positional = a file of pairs -> {out}.1d.fasta, {out}.2d.fasta,
    {out}.log; two positionals = one pair -> {out}.fasta."""
    logger = logging.getLogger("poreover_amd")
    in_path = getattr(args, 'in')
    if len(in_path) == 1:
        with open(in_path[0], 'r') as read_pairs:
            pairs = [line.split() for line in read_pairs if line.split()]
        logger.info("found {} read pairs in {}".format(len(pairs), in_path[0]))
        results = decode_pairs(pairs, args)
        keys = ["read1", "read2", "length1", "length2", "sequence_identity", "skipped"]
        with open(args.out + '.1d.fasta', 'w') as out_1d_f, open(args.out + '.2d.fasta', 'w') as out_2d_f, \
                open(args.out + '.log', 'w', 1) as log_f:
            print('# PoreOver pair-decode', file=log_f)
            print('# ' + str(vars(args)), file=log_f)
            print('# ' + '\t'.join(keys), file=log_f)
            for x in results:   # input order (the reference: process-completion order)
                if len(x) == 3:
                    print(x[0], file=out_1d_f)
                    print(x[1], file=out_2d_f)
                    print('\t'.join(map(str, [x[2].get(k, "") for k in keys])), file=log_f)
                elif len(x) == 2:
                    print(x[0], file=out_2d_f)
                    print('\t'.join(map(str, [x[1].get(k, "") for k in ["read1", "read2"]])), file=log_f)
                elif len(x) == 1:
                    print('\t'.join(map(str, [x[0].get(k, "") for k in keys])), file=log_f)
    else:
        res = pair_decode_helper(args)
        if len(res) == 3:
            seqs_1d, seq_2d, summary = res
        elif len(res) == 2:
            seq_2d, summary = res
        else:
            print(res[0], file=sys.stderr)
            return
        print(summary, file=sys.stderr)
        with open(args.out + '.fasta', 'w') as out_fasta:
            print(seq_2d, file=out_fasta)
