#!/bin/bash
# round-3 evidence for the single-launch test-time kernel: parity tests, then rocprofv3 kernel traces of
# tools/eval_bench.py (hipGraph replay) with the staged kernels and with the fused kernel, dense and blob scene
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(python -m pytest tests/test_gpu_fused.py tests/test_gpu_eval.py tests/test_gpu_graph.py tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_lifecycle.py tests/test_gpu_edge.py -q 2>&1 | tail -6) > gpurun_out/r3e_tests.log
cd /tmp && export TMPDIR=/tmp
for f in "" "--fused"; do for sc in random blobs; do
  tag=r3e_testoptim_${sc}${f:+_fused}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o k -- python3 $R/tools/eval_bench.py --size 400 --samples 1000 --test-iters 100 --no-render --graph --scene $sc $f > $R/gpurun_out/$tag.log 2>&1
  python3 $R/tools/prof_summary.py $R/gpurun_out/$tag/k_kernel_stats.csv 14 200 > $R/gpurun_out/${tag}_summary.txt
  grep -o "\"ms_per_iter\": [0-9.]*" $R/gpurun_out/$tag.log >> $R/gpurun_out/${tag}_summary.txt
  rm -f $R/gpurun_out/$tag/*kernel_trace.csv
done; done
cd $R; cat gpurun_out/r3e_tests.log
for t in gpurun_out/r3e_testoptim_*_summary.txt; do echo "== $t"; head -9 $t; tail -1 $t; done
