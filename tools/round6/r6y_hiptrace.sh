#!/bin/bash
# round 6, run y: HIP API trace of the host-bound eager step: which runtime call does the host wait in?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
F="--scene fitted --no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
timeout 600 rocprofv3 --hip-runtime-trace --output-format csv -d $O/r6y_hip -o h -- python3 $R/bench.py $F --steps 100 --warmup 10 > $O/r6y.log 2>&1
ls $O/r6y_hip
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$O/r6y_hip/*hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
rows.sort()
print(len(rows), "calls")
last = rows[len(rows) * 2 // 3:]          # the timed steps
tot = collections.defaultdict(lambda: [0, 0, 0])
for s, e, f, t in last:
    d = tot[f]; d[0] += 1; d[1] += e - s; d[2] = max(d[2], e - s)
for f, (n, t, m) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:20]:
    print("%-40s n=%6d total %9.1f us  avg %7.1f us  max %8.1f us" % (f, n, t / 1e3, t / 1e3 / n, m / 1e3))
# the long calls, with what came before them
print("calls longer than 100 us in the last third:")
k = 0
for i, (s, e, f, t) in enumerate(last):
    if e - s > 100e3 and k < 40:
        k += 1
        prev = last[i - 1]
        print("  %-28s %8.1f us  thread %s   (previous call: %s, ended %.1f us earlier)" % (f, (e - s) / 1e3, t, prev[2], (s - prev[1]) / 1e3))
PY
rm -rf $O/r6y_hip
