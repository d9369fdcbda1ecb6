#!/bin/bash
# round 6, run ab: ordered kernel sequence of ONE replayed test-time pose-optimisation iteration (start offsets, durations, gaps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/r6ab_trace -o k -- python3 $R/tools/eval_bench.py --no-render --graph --test-iters 100 > $O/r6ab.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/r6ab_trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# iterations delimited by k_march_fwd
cuts = [i for i, r in enumerate(rows) if "k_march_fwd" in r[2]]
a, b = cuts[-20], cuts[-19]
# start from the first kernel after the previous iteration's last big kernel: print from a-12 to b
t0 = rows[a - 14][0]
prev = None
for s, e, n in rows[a - 14:b + 2]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%9.1f  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:110]))
    prev = e
its = [(rows[cuts[i + 1]][0] - rows[cuts[i]][0]) / 1e3 for i in range(len(cuts) - 60, len(cuts) - 1)]
print("iteration period us: median %.1f" % sorted(its)[len(its) // 2])
PY
rm -rf $O/r6ab_trace
