"""Test helper: a CPU stand-in for the engine call of the pair-decode driver, built on the oracle (test
infrastructure).  The sharding tests inject it into pair_decode.decode_pairs so that the multi-process paths
(torchrun ranks over gloo, spawned per-device workers) run on machines without a GPU and decode REAL fixture pairs."""
import os
from pathlib import Path

import numpy as np


def oracle_decode_pairs(in_paths, args):
    """decode_pairs_local's contract (list of the reference's return tuples) for the default route, on the CPU"""
    from oracle import po_oracle as O
    from poreover_amd.decoding.decode import fasta_format
    out = []
    for in_path in in_paths:
        y1 = np.log(np.load(os.path.join(args.dir, in_path[0])))
        y2 = np.log(np.load(os.path.join(args.dir, in_path[1])))
        r = O.pair_decode(y1, y2, "poreover", args.beam_width, args.beam_search_method, args.padding, args.alignment)
        summary = {'read1': in_path[0], 'read2': in_path[1], 'length1': r["length1"], 'length2': r["length2"]}
        if r["status"] == O.SKIP_LENGTH:
            summary['skipped'] = 1
            out.append([summary])
            continue
        summary['sequence_identity'] = r["sequence_identity"]
        if r["status"] == O.SKIP_IDENTITY:
            summary['skipped'] = 1
            out.append([summary])
            continue
        summary['skipped'] = 0
        out.append((fasta_format(in_path[0], r["seq1"]) + fasta_format(in_path[1], r["seq2"]),
                    fasta_format('consensus;{};{}'.format(Path(in_path[0]).stem, Path(in_path[1]).stem), r["consensus"]),
                    summary))
    return out


def tagged_decode(in_paths, args):
    """as above, plus the pid that decoded each pair (to see that the work really was spread)"""
    return [(rec, os.getpid()) for rec in oracle_decode_pairs(in_paths, args)]
