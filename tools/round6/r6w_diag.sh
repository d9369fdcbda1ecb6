#!/bin/bash
# round 6, run w: where the HOST time of a host-bound eager step goes (cProfile), and the kernel trace of test-time optimisation
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
F="--scene fitted --no-extras --no-roofline --no-cpu-baseline --no-torch-baseline --no-probe"
timeout 600 python bench.py $F --steps 100 --warmup 20 2>/dev/null | tail -1 > $O/r6w_fitted_line.json
timeout 600 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py'] + '$F --steps 300 --warmup 20'.split()
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats('tottime').print_stats(45)
    st.sort_stats('cumtime').print_stats(70)
    open('$O/r6w_fitted_cprofile.txt', 'w').write(s.getvalue())
" > $O/r6w_cprofile.log 2>&1
python - <<'PY'
import json
print(open("gpurun_out/r6w_fitted_line.json").read()[:300])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6w_testoptim_trace -o k -- python3 $GRAFT_REPO_ROOT/tools/eval_bench.py --no-render --graph --test-iters 100 > $O/r6w_testoptim.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py $O/r6w_testoptim_trace/k_kernel_stats.csv 30 200 > $O/r6w_testoptim_trace_summary.txt
rm -rf $O/r6w_testoptim_trace/*kernel_trace.csv
head -30 $O/r6w_testoptim_trace_summary.txt
head -60 $O/r6w_fitted_cprofile.txt
