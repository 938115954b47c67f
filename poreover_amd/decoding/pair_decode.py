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
    method = getattr(args, 'method', 'envelope')
    if method == 'split':
        # pair_decode.py:336-354: upstream this route only returns together with --diagonal_envelope (the other
        # branch of its final return needs 1-D basecalls the split route never makes: NameError)
        if not getattr(args, 'diagonal_envelope', False):
            raise _lib.EngineError(_lib.E_UNSUPPORTED, "pair-decode --method split without --diagonal_envelope",
                                   "the reference raises NameError on this combination")
    elif method != 'envelope':
        raise _lib.EngineError(_lib.E_UNSUPPORTED, "pair-decode --method %s" % method,
                               "only the envelope and split methods are built ('align' only prints statistics)")
    if getattr(args, 'algorithm', 'beam') != 'beam':
        # pair_decode.py:222: assert(self.kind == "poreover") compares the model name 'ctc' with 'poreover' and
        # always fails; the pair prefix search itself is available as prefix_search.pair_prefix_search_log_cy
        raise _lib.EngineError(_lib.E_UNSUPPORTED, "pair-decode --algorithm %s" % args.algorithm,
                               "the reference's own assert refuses it for every basecaller")
    if getattr(args, 'skip_matches', False) and (getattr(args, 'diagonal_envelope', False) or
                                                 getattr(args, 'single', 'viterbi') != 'viterbi'):
        raise _lib.EngineError(_lib.E_UNSUPPORTED, "pair-decode --skip_matches with --diagonal_envelope / --single beam")


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


def get_anchors(alignment, matches, indels):
    """pair_decode.py:53-89: runs of >= `matches` identical columns ('mat') or >= `indels` gap columns
    ('ins' / 'del') of a 2-row alignment -> ([(start, end), ...], [type, ...]).  A mismatch column ends any
    run and never starts one; the run open at the end of the alignment is not reported."""
    ranges, types = [], []
    start, count, prev = 0, 1, 'START'
    need = {'ins': indels, 'del': indels, 'mat': matches}
    for i, (a1, a2) in enumerate(zip(alignment[0], alignment[1])):
        state = 'mat' if a1 == a2 else ('ins' if a1 == '-' else ('del' if a2 == '-' else 'mis'))
        if state == prev and state != 'mis':
            count += 1
            continue
        if prev in need and count >= need[prev]:
            ranges.append((start, i))
            types.append(prev)
        prev, count, start = state, 1, i
    return ranges, types


def _decode_pairs_skip_matches(in_paths, loaded, args, out):
    """--skip_matches (pair_decode.py:412-467,512-522): long runs of matching columns are copied from read 1's
    basecall, the stretches between them ("boxes") are decoded with the pair beam search inside their slice of
    the envelope, and the pieces are joined in signal order.  Viterbi, alignment, envelope and every box of
    every pair run as batched GPU calls."""
    import numpy as np
    by_kind = {}
    for i, (_, _, m1, _) in enumerate(loaded):
        by_kind.setdefault(m1.kind, []).append(i)
    for kind, idx in by_kind.items():
        model = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}[kind]
        ys1 = [loaded[i][2].log_prob for i in idx]
        ys2 = [loaded[i][3].log_prob for i in idx]
        seq1, map1, st1 = _batch.viterbi_batch(ys1, kind, return_map=True)
        seq2, map2, st2 = _batch.viterbi_batch(ys2, kind, return_map=True)
        live = []
        for j, i in enumerate(idx):
            in_path = in_paths[i]
            summary = {'read1': in_path[0], 'read2': in_path[1], 'length1': len(seq1[j]), 'length2': len(seq2[j])}
            if abs(len(seq1[j]) - len(seq2[j])) > 1000:
                summary['skipped'] = 1
                out[i] = [summary]
            elif st1[j] != 0 or st2[j] != 0:   # the reference's assert on the frame map (pair_decode.py:379,382)
                raise _lib.EngineError(_lib.E_ARG, "pair decode of pair %d" % i, "frame map and basecall lengths differ")
            else:
                live.append((j, i, summary))
        alns = _batch.align_batch([(seq1[j], seq2[j]) for j, _, _ in live], 0 if args.alignment == "full" else 500)
        keep = []
        for (j, i, summary), (a1, a2) in zip(live, alns):
            ident = sum(x == y for x, y in zip(a1, a2)) / len(a1)
            summary['sequence_identity'] = ident
            if ident < 0.5:
                summary['skipped'] = 1
                out[i] = [summary]
            else:
                summary['skipped'] = 0
                keep.append((j, i, summary, (a1, a2)))
        envs = _batch.envelope_batch([k[3] for k in keep], [map1[k[0]] for k in keep], [map2[k[0]] for k in keep],
                                     [len(ys1[k[0]]) for k in keep], [len(ys2[k[0]]) for k in keep], args.padding)
        box_y1, box_y2, box_env, box_owner, anchors_of = [], [], [], [], {}
        for (j, i, summary, aln), env in zip(keep, envs):
            s2s1, s2s2 = map1[j], map2[j]
            # alignment column -> number of bases of each read up to and including it; column 0 looks at
            # "column -1", i.e. the still-zero last entry (pair_decode.py:403-410)
            a2s = np.zeros((2, len(aln[0])), dtype=np.int64)
            for r in range(2):
                acc = 0
                for c, ch in enumerate(aln[r]):
                    acc = acc if ch == '-' else acc + 1
                    a2s[r, c] = acc
            ranges, types = get_anchors(aln, matches=args.skip_threshold, indels=100)
            assert len(ranges) > 0, 'No matches/indels of sufficient length found in alignment. Try decreasing --matches or --indels'
            anchors, boxes = [], []
            for k, (cs, ce) in enumerate(ranges):
                row = 1 if types[k] == 'ins' else 0
                anchors.append((int(s2s1[a2s[0, cs]]), aln[row][cs:ce]))
                if k > 0:
                    pe = ranges[k - 1][1]
                    boxes.append((int(s2s1[a2s[0, pe]]), int(s2s1[a2s[0, cs]]), int(s2s2[a2s[1, pe]]), int(s2s2[a2s[1, cs]])))
                else:
                    boxes.append((0, int(s2s1[a2s[0, cs]]), 0, int(s2s2[a2s[1, cs]])))
            le = ranges[-1][1]
            boxes.append((int(s2s1[a2s[0, le]]), len(ys1[j]), int(s2s2[a2s[1, le]]), len(ys2[j])))
            anchors_of[i] = anchors
            for b in boxes:
                e = np.array(env[b[0]:b[1]], dtype=np.int64)
                lo = int(e[0, 0])            # an empty box raises here, as in the reference
                box_y1.append(ys1[j][b[0]:b[1]])
                box_y2.append(ys2[j][lo:int(e[-1, 1])])
                box_env.append(e - lo)
                box_owner.append((i, b[0]))
        calls = _batch.beam_search_2d_batch(box_y1, box_y2, box_env, args.beam_width, model=model,
                                            method=args.beam_search_method)
        pieces = {}
        for (i, pos), sq in zip(box_owner, calls):
            pieces.setdefault(i, []).append((pos, sq))
        for j, i, summary, aln in keep:
            in_path = in_paths[i]
            path1, path2 = loaded[i][0], loaded[i][1]
            joined = ''.join(x[1] for x in sorted(pieces.get(i, []) + anchors_of[i]))
            out[i] = (fasta_format(in_path[0], seq1[j]) + fasta_format(in_path[1], seq2[j]),
                      fasta_format('consensus;{};{}'.format(path1.stem, path2.stem), joined), summary)
    return out


def _decode_pairs_split(in_paths, loaded, args, out):
    """--method split (pair_decode.py:336-354, parallel_decoder._beam_search_2d :149-165): boxes of --window frames
    of read 1 along the main diagonal, each decoded on its own — pair beam search without an envelope (method
    "row"), or 1-D prefix search when a box has no extent on one read — and the pieces joined in order.  All
    boxes of all pairs go to the GPU in one batched call per kind of box."""
    from . import prefix_search
    by_kind = {}
    for i, (_, _, m1, _) in enumerate(loaded):
        by_kind.setdefault(m1.kind, []).append(i)
    for kind, idx in by_kind.items():
        model = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}[kind]
        b1, b2, owner, pieces = [], [], [], {}
        for i in idx:
            y1, y2 = loaded[i][2].log_prob, loaded[i][3].log_prob
            U, V = len(y1), len(y2)
            boxes = [(u - args.window, u, int(V / U * (u - args.window)), int(V / U * u))
                     for u in range(args.window, U, args.window)]
            boxes.append((boxes[-1][1], U, boxes[-1][3], V))   # IndexError for reads shorter than a window, as upstream
            pieces[i] = [None] * len(boxes)
            for k, (u1, u2, v1, v2) in enumerate(boxes):
                size = (u2 - u1 + 1) * (v2 - v1 + 1)
                if size <= 1:
                    pieces[i][k] = ''
                elif (u2 - u1) < 1:
                    pieces[i][k] = prefix_search.prefix_search_log_cy(y2[v1:v2])[0]
                elif (v2 - v1) < 1:
                    pieces[i][k] = prefix_search.prefix_search_log_cy(y1[u1:u2])[0]
                else:
                    b1.append(y1[u1:u2]); b2.append(y2[v1:v2]); owner.append((i, k))
        calls = _batch.beam_search_2d_batch(b1, b2, None, args.beam_width, model=model, method="row") if b1 else []
        for (i, k), sq in zip(owner, calls):
            pieces[i][k] = sq
        for i in idx:
            path1, path2 = loaded[i][0], loaded[i][1]
            # the header keeps the quirk of pair_decode.py:527 (the format string drops its third argument)
            out[i] = (fasta_format('consensus;{};{}'.format(args.method, path1.stem, path2.stem), ''.join(pieces[i])),
                      {'read1': in_paths[i][0], 'read2': in_paths[i][1]})
    return out


def _pair_cost(in_path, args):
    """Cost proxy of a pair before anything is decoded: bytes of its two trace files (proportional to the frames
    U + V)."""
    tot = 0
    for p in in_path[:2]:
        q = Path(p)
        if q.suffix == ".fast5":
            q = q.with_suffix(".npy")
        try:
            tot += os.path.getsize(os.path.join(args.dir, q))
        except OSError:
            tot += 1
    return max(tot, 1)


def decode_pairs(in_paths, args, devices=None, decode_fn=None):
    """pair_decode_helper for a LIST of pairs, spread over the GPUs of the node where the reference spreads them
    over processes (pair_decode.py:292-297): returns the reference's return tuples in input order.
      * under torchrun (WORLD_SIZE > 1): this rank decodes its shard on device LOCAL_RANK, rank 0 gets every record,
        the other ranks None;
      * otherwise THIS process drives every visible device (devices=[...] names them; an index may repeat): the
        pipelined engine call takes the device list (po_multi_pair_decode: a pipeline and a host thread per device, waves
        dealt as devices become free).  The `split` / `--skip_matches` routes, which make several engine calls per
        pair, and an injected decode_fn still go through one spawned worker process per device (dist.run_sharded).
    decode_fn(list_of_pairs, args) -> records replaces the engine call (tests inject a CPU function)."""
    from .. import dist as podist
    fn = decode_fn or decode_pairs_local
    rank, local_rank, world = podist.env_rank_world()
    costs = [_pair_cost(p, args) for p in in_paths]
    if world > 1:
        if decode_fn is None:
            _lib.set_device(local_rank)
        return podist.decode_distributed(in_paths, costs, fn, args)
    threads = getattr(args, 'threads', 1)
    devs = podist.plan_devices(len(in_paths), devices, threads if threads and threads > 1 else None)
    if len(devs) <= 1:
        if devs and devs[0] != 0 and decode_fn is None:
            _lib.set_device(devs[0])
        return fn(in_paths, args)
    if decode_fn is None and getattr(args, 'method', 'envelope') != 'split' and not getattr(args, 'skip_matches', False) \
            and getattr(args, 'single', 'viterbi') == 'viterbi':
        return decode_pairs_local(in_paths, args, devices=devs)
    return podist.run_sharded(in_paths, costs, fn, devs, args, bind_device=decode_fn is None)


def _write_debug_pickle(pair, rec, args):
    """--debug (pair_decode.py:482-490): debug.p with the alignment, the per-base frame maps and the alignment-column ->
    base-count table of the (last) pair, rebuilt from the engine's stage outputs"""
    import pickle
    import numpy as np
    _, _, m1, m2 = pair
    (s1, s2), _paths, maps, _st = _batch.viterbi_batch([m1.log_prob, m2.log_prob], m1.kind, return_path=True, return_map=True)
    a1, a2 = _batch.align_batch([(s1, s2)], 0 if args.alignment == "full" else 500)[0]
    alignment = np.array([list(a1), list(a2)])
    a2s = np.zeros(shape=alignment.shape, dtype=int)
    for i, col in enumerate(alignment.T):      # (column 0 looks at column -1, still zero: pair_decode.py:403-410)
        for r in range(2):
            a2s[r, i] = a2s[r, i - 1] if col[r] == '-' else a2s[r, i - 1] + 1
    with open("debug.p", "wb") as pfile:
        pickle.dump({'alignment_to_sequence': a2s, 'sequence_to_signal1': [int(x) for x in maps[0]],
                     'sequence_to_signal2': [int(x) for x in maps[1]], 'alignment': alignment}, pfile)


def decode_pairs_local(in_paths, args, devices=None):
    """pair_decode_helper for a list of pairs in THIS process — on its device, or on every device of `devices` through
    the multi-device pipeline: returns a list of the reference's return tuples (1-, 2- or 3-tuples,
    pair_decode.py:375,398,526-529)."""
    _check_supported(args)
    loaded = [_load_pair(p, args) for p in in_paths]
    out = [None] * len(loaded)
    if getattr(args, 'method', 'envelope') == 'split':
        return _decode_pairs_split(in_paths, loaded, args, out)
    if getattr(args, 'skip_matches', False):
        return _decode_pairs_skip_matches(in_paths, loaded, args, out)
    # pairs that can share one engine call: same kind, same input form (float32 logits / uint8 trace / float64
    # log-probabilities) and the same pending column order / reversal — with --reverse_complement on raw logits that
    # is every pair of the file, and the log-softmax, the permutations and the reversal run on the device
    groups = {}
    for i, (_, _, m1, m2) in enumerate(loaded):
        if getattr(args, 'single', 'viterbi') != 'viterbi' or m1.engine_input()[3]:
            key = (m1.kind, 'host')          # --single beam works on the float64 tables
        else:
            e1, e2 = m1.engine_input(), m2.engine_input()
            key = (m1.kind, e1[1], tuple(e1[2]), e2[1], tuple(e2[2]), e2[3]) if e1[1] == e2[1] else (m1.kind, 'host')
        groups.setdefault(key, []).append(i)
    for key, idx in groups.items():
        kind = key[0]
        common = dict(kind=kind, beam_width=args.beam_width, method=args.beam_search_method, padding=args.padding,
                      alignment=args.alignment, diagonal_envelope=args.diagonal_envelope,
                      diagonal_width=args.diagonal_width)
        want_env = bool(getattr(args, 'debug_envelope', False))
        if key[1] == 'host' and getattr(args, 'single', 'viterbi') != 'viterbi':
            res = _batch.pair_decode_batch([loaded[i][2].log_prob for i in idx], [loaded[i][3].log_prob for i in idx],
                                           single=args.single, **common)
        elif key[1] == 'host':
            res = _batch.pair_decode_stream([loaded[i][2].log_prob for i in idx], [loaded[i][3].log_prob for i in idx],
                                            strict=False, return_envelope=want_env, devices=devices, **common)
        else:
            ident = list(range(len(key[2])))
            res = _batch.pair_decode_stream([loaded[i][2].engine_input()[0] for i in idx],
                                            [loaded[i][3].engine_input()[0] for i in idx],
                                            perm1=None if list(key[2]) == ident else list(key[2]),
                                            perm2=None if list(key[4]) == ident else list(key[4]), reverse2=key[5],
                                            strict=False, return_envelope=want_env, devices=devices, **common)
        if getattr(args, 'debug', False) and not args.diagonal_envelope:
            _write_debug_pickle(loaded[idx[-1]], res[-1], args)
        for i, r in zip(idx, res):
            in_path = in_paths[i]
            path1, path2 = loaded[i][0], loaded[i][1]
            if r["status"] not in (0, _lib.SKIP_LENGTH, _lib.SKIP_IDENTITY):
                # a per-pair engine refusal (capacity, an envelope the reference itself is undefined on): the pair is
                # reported and skipped, the other pairs of the batch are unaffected — whatever the output route
                logging.getLogger("poreover_amd").warning("pair %s %s not decoded: %s", in_path[0], in_path[1],
                                                          _lib._CODE_NAMES.get(r["status"], r["status"]))
                out[i] = [{'read1': in_path[0], 'read2': in_path[1], 'length1': r["length1"], 'length2': r["length2"],
                           'skipped': 1, 'error': _lib._CODE_NAMES.get(r["status"], str(r["status"]))}]
                continue
            if getattr(args, 'debug_envelope', False) and r["status"] == 0:
                # pair_decode.py:503-507: band statistics of the envelope instead of a consensus
                import numpy as np
                env = np.asarray(r["envelope"])
                size = env[:, 1] - env[:, 0]
                U, V = loaded[i][2].t_max, loaded[i][3].t_max
                print(path1.stem, path2.stem, r["length1"], r["length2"], U, V, np.mean(size), np.std(size), np.median(size),
                      np.min(size), np.max(size))
                out[i] = [{"skipped": 1}]
                continue
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
        if results is None:   # a rank other than 0 of a torchrun launch: rank 0 writes the files
            return
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
