#!/bin/bash
# ordered kernel timeline of the last traced iteration (start offset, duration, queue) -- eager, default bench workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="$R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-live-pmc $*"
rm -rf $O/tl
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o k -- python3 $CMD > $O/tl.log 2>&1
python3 - $O/tl > $O/r5_timeline.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70], r.get("Queue_Id")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_pose_fwd" in r[2]]
for a, b in zip(idx[-4:-1], idx[-3:]):
    it = rows[a:b]
    t0 = it[0][0]
    span = (rows[b][0] - t0) / 1e3
    main = [r for r in it if r[3] == it[0][3]]
    busy_main = sum(e - s for s, e, *_ in main) / 1e3
    print("iteration: launches %d, start-to-next-start %.1f us, main-queue busy %.1f us, idle %.1f us" % (len(it), span, busy_main, span - busy_main))
it = rows[idx[-2]:idx[-1]]
t0 = it[0][0]
prev_end = {}
for s, e, n, q in it:
    gap = (s - prev_end.get(q, s)) / 1e3
    prev_end[q] = e
    print("  %8.1f %8.1f  gap %6.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, n))
PY
head -60 $O/r5_timeline.txt
