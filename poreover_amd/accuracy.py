"""Accuracy report of decoded sequences against a known truth: the per-record columns of
the reference's `poreover benchmark` (benchmark.py:199-280, parse_cs: match / mismatch / insertion / deletion /
alignment_length / identity = match / alignment_length) without its minimap2 dependency — the truth of a synthetic
read is known, so each record is aligned to it globally (unit costs) instead of mapped to a reference genome.
Host-side numpy; nothing here is on the decode path."""
import numpy as np

__all__ = ["alignment_summary", "identity_table", "table_to_csv"]


def alignment_summary(query, ref):
    """Global unit-cost alignment of query to ref -> the counts benchmark.py reports per record."""
    n, m = len(query), len(ref)
    if n == 0 or m == 0:
        return {"match": 0, "mismatch": 0, "insertion": n, "deletion": m, "alignment_length": n + m,
                "identity": 0.0, "edit_distance": n + m}
    q = np.frombuffer(query.encode(), dtype=np.uint8)
    r = np.frombuffer(ref.encode(), dtype=np.uint8)
    idx = np.arange(m + 1)
    D = np.empty((n + 1, m + 1), dtype=np.int32)
    D[0] = idx
    for i in range(1, n + 1):
        t = np.empty(m + 1, dtype=np.int64)
        t[0] = i
        t[1:] = np.minimum(D[i - 1, 1:] + 1, D[i - 1, :-1] + (r != q[i - 1]))
        D[i] = np.minimum.accumulate(t - idx) + idx      # the left neighbour: running minimum
    i, j = n, m
    c = {"match": 0, "mismatch": 0, "insertion": 0, "deletion": 0}
    while i > 0 or j > 0:
        if i > 0 and j > 0 and D[i, j] == D[i - 1, j - 1] + (q[i - 1] != r[j - 1]):
            c["match" if q[i - 1] == r[j - 1] else "mismatch"] += 1
            i -= 1; j -= 1
        elif i > 0 and D[i, j] == D[i - 1, j] + 1:
            c["insertion"] += 1; i -= 1          # a base of the query the truth does not have
        else:
            c["deletion"] += 1; j -= 1
    c["alignment_length"] = c["match"] + c["mismatch"] + c["insertion"] + c["deletion"]
    c["identity"] = c["match"] / c["alignment_length"]
    c["edit_distance"] = int(D[n, m])
    return c


def identity_table(records):
    """records: iterable of (name, {"read1": seq, "read2": seq, "consensus": seq or None}, truth) -> list of rows
    (one per sequence, as benchmark.py writes one CSV row per FASTA record) and a summary of mean identities."""
    rows = []
    for name, seqs, truth in records:
        for kind in ("read1", "read2", "consensus"):
            s = seqs.get(kind)
            if s is None:
                continue
            row = {"name": name, "kind": kind, "length": len(s)}
            row.update(alignment_summary(s, truth))
            rows.append(row)
    summary = {}
    for kind in ("read1", "read2", "consensus"):
        sel = [r for r in rows if r["kind"] == kind]
        if sel:
            tot = {k: sum(r[k] for r in sel) for k in ("match", "mismatch", "insertion", "deletion", "alignment_length")}
            summary[kind] = {"records": len(sel), "identity": tot["match"] / max(tot["alignment_length"], 1),
                             "mean_record_identity": float(np.mean([r["identity"] for r in sel])), **tot}
    return rows, summary


def table_to_csv(rows):
    cols = ["name", "kind", "length", "match", "mismatch", "insertion", "deletion", "alignment_length", "identity"]
    out = [",".join(cols)]
    for r in rows:
        out.append(",".join(("%.6f" % r[c]) if c == "identity" else str(r[c]) for c in cols))
    return "\n".join(out) + "\n"
