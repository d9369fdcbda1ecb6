#!/bin/bash
# round 6, run c: G2 derived inside the dW2 GEMM (no G2 rows in the records), chip-geometry query, engine-trace tests on the GPU
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > gpurun_out/r6c_parity.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp or llff_final_grid or 299cube or configs3" 2>&1 | tail -5 > gpurun_out/r6c_fullsize.txt
timeout 900 python -m pytest tests/test_engine_trace.py tests/test_gpu_dist.py::test_rccl_process_group_of_one_rank tests/test_gpu_matrix_mode.py tests/test_gpu_guards.py -x -q -s 2>&1 | tail -40 > gpurun_out/r6c_engine.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
for i in 1 2 3; do
$B > gpurun_out/r6c_bench_lean_$i.json 2> gpurun_out/r6c_bench_lean_$i.err
JT_LEAN_TAPE=0 $B > gpurun_out/r6c_bench_full_$i.json 2>/dev/null
done
JT_SCATTER_WGS=232 $B > gpurun_out/r6c_bench_lean_wg232.json 2>/dev/null
JT_SCATTER_WGS=240 $B > gpurun_out/r6c_bench_lean_wg240.json 2>/dev/null
JT_SCATTER_WGS=208 $B > gpurun_out/r6c_bench_lean_wg208.json 2>/dev/null
$B --config bat_llff_VM_MLP > gpurun_out/r6c_bench_llff_lean.json 2>/dev/null
JT_LEAN_TAPE=0 $B --config bat_llff_VM_MLP > gpurun_out/r6c_bench_llff_full.json 2>/dev/null
$B --stage 0 > gpurun_out/r6c_bench_stage0_lean.json 2>/dev/null
JT_LEAN_TAPE=0 $B --stage 0 > gpurun_out/r6c_bench_stage0_full.json 2>/dev/null
$B --config bat_llff_VM_MLP --stage 0 > gpurun_out/r6c_bench_llff0_lean.json 2>/dev/null
JT_LEAN_TAPE=0 $B --config bat_llff_VM_MLP --stage 0 > gpurun_out/r6c_bench_llff0_full.json 2>/dev/null
JT_SCATTER_WAVES=16 $B --config bat_llff_VM_MLP --stage 0 > gpurun_out/r6c_bench_llff0_lean_w16.json 2>/dev/null
cat gpurun_out/r6c_parity.txt gpurun_out/r6c_fullsize.txt; tail -25 gpurun_out/r6c_engine.txt
