cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/gpurun_out/e2e_trace -- python3 $root/scripts/e2e_trace.py 10000 > $root/gpurun_out/r04_g54_e2e_trace.log 2>&1
cd $root
python3 - <<'PY' > gpurun_out/r04_g54_timeline.txt 2>&1
import csv, glob
kt = glob.glob("gpurun_out/e2e_trace/**/*kernel_trace.csv", recursive=True)
mc = glob.glob("gpurun_out/e2e_trace/**/*memory_copy_trace.csv", recursive=True)
ev = []
for f in kt:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", r.get("Stream_Id", ""))))
for f in mc:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M", r.get("Direction", r.get("Name", "")), ""))
ev.sort()
# the last ~0.2 s of activity = the last repetition
t_end = ev[-1][1]
sel = [e for e in ev if e[0] > t_end - 160_000_000]
t0 = sel[0][0]
for s, e, kind, name, q in sel:
    if (e - s) > 300_000:
        print("%8.2f ms  +%7.2f ms  %s %s %s" % ((s - t0) / 1e6, (e - s) / 1e6, kind, name, q))
PY
exit 0
