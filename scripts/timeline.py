#!/usr/bin/env python3
"""Timeline of the LAST job in a rocprofv3 --kernel-trace --memory-copy-trace --output-format csv directory (jobs are separated
by device-idle gaps of more than GAP ms): python3 scripts/timeline.py DIR [GAP=6] [MINDUR=0.3]"""
import csv, glob, sys
root = sys.argv[1]
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
mind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M", r["Direction"].replace("MEMORY_COPY_", "")[:30], r.get("Bytes", r.get("Size", "0"))))
ev.sort()
segs, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] - end > gap * 1e6:
        segs.append(cur); cur = []
    cur.append(e); end = e[1] if end is None else max(end, e[1])
segs.append(cur)
segs = [s for s in segs if len(s) > 20]
print("%d jobs; the last one:" % len(segs))
ev = segs[-1]
t0 = ev[0][0]
def union(xs):
    xs = sorted(xs); tot = 0; cs, ce = xs[0]
    for s, e in xs[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
ks = [(e[0], e[1]) for e in ev if e[2] == "K"]
h2d = [(e[0], e[1]) for e in ev if e[2] == "M" and "HOST_TO_DEVICE" in e[3]]
print("span %.2f ms; kernels busy (union) %.2f ms; H2D busy (union) %.2f ms, last H2D ends at %.2f ms" % (
    (max(e[1] for e in ev) - t0) / 1e6, union(ks) / 1e6, union(h2d) / 1e6, (max(e[1] for e in h2d) - t0) / 1e6))
for e in ev:
    d = (e[1] - e[0]) / 1e6
    if d > mind and e[2] == "K":
        print("%8.2f ms + %7.2f ms  %s %s" % ((e[0] - t0) / 1e6, d, e[3], e[4]))

# the uploads in 5 ms buckets: bytes on the wire and busy time
print("H2D per 5 ms bucket: start ms, MB, busy ms")
bk = {}
for e in ev:
    if e[2] == "M" and "HOST_TO_DEVICE" in e[3]:
        b = int((e[0] - t0) / 5e6)
        x = bk.setdefault(b, [0.0, 0.0])
        try: x[0] += float(e[4]) / 1e6
        except ValueError: pass
        x[1] += (e[1] - e[0]) / 1e6
for b in sorted(bk): print("%6d  %8.1f MB  %6.2f ms" % (b * 5, bk[b][0], bk[b][1]))
