#!/usr/bin/env python3
"""Basic-block census of one kernel in a hipcc -S listing: per block the VALU / SALU / LDS / VMEM / scratch /
lane-move counts and the back edges (loops).  Usage: blocks.py file.s kernel_mangled_name [--min N]"""
import re, sys
def main():
    path, kname = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kname + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], {"label": "entry", "line": start, "ins": []}
    for i in range(start + 1, end + 1):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = {"label": m.group(1), "line": i, "ins": []}
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."): continue
        cur["ins"].append(s)
    blocks.append(cur)
    idx = {b["label"]: k for k, b in enumerate(blocks)}
    tot = {}
    for k, b in enumerate(blocks):
        c = dict(valu=0, salu=0, lds=0, vmem=0, scr=0, lanemv=0, smem=0, wait=0)
        back = []
        for s in b["ins"]:
            op = s.split()[0]
            if op.startswith("scratch_"): c["scr"] += 1
            elif op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): c["lanemv"] += 1; c["valu"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): c["vmem"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_dcache"): c["smem"] += 1
            elif op.startswith("s_waitcnt"): c["wait"] += 1; c["salu"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                t = s.split()[-1]
                if t in idx and idx[t] <= k: back.append(t)
        b["c"] = c; b["back"] = back
        for kk, v in c.items(): tot[kk] = tot.get(kk, 0) + v
    print("total", tot, "blocks", len(blocks))
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    for k, b in enumerate(blocks):
        c = b["c"]
        if len(b["ins"]) >= mn or b["back"]:
            print(f"{k:4d} {b['label']:12s} L{b['line']:6d} n={len(b['ins']):4d} valu={c['valu']:4d} salu={c['salu']:4d} lds={c['lds']:3d} vmem={c['vmem']:3d} scr={c['scr']:2d} lane={c['lanemv']:3d} wait={c['wait']:2d}" + (f"  BACK->{','.join(b['back'])}" if b["back"] else ""))
main()
