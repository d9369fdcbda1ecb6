#!/bin/bash
# the split backward's chain on the bf16 matrix cores (matrix-mode bit 2)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix_mode.py tests/test_gpu_edge.py tests/test_gpu_eval.py -x -q 2>&1 | tail -n 6 > gpurun_out/r5_b16chain_tests.log
JT_BF16X3=3 python tools/eval_bench.py --no-render --graph --test-iters 100 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c330-460 > gpurun_out/r5_b16chain_old.txt
python tools/eval_bench.py --no-render --graph --test-iters 100 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c330-460 > gpurun_out/r5_b16chain_new.txt
JT_BF16X3=3 bash tools/kstat.sh r5bc_llff_old --config bat_llff_VM_MLP > gpurun_out/r5_b16chain_llff_old.txt 2>&1
bash tools/kstat.sh r5bc_llff --config bat_llff_VM_MLP > gpurun_out/r5_b16chain_llff.txt 2>&1
JT_BWD_SPLIT=1 JT_TILE_CFG=1 bash tools/kstat.sh r5bc_tile > gpurun_out/r5_b16chain_tile.txt 2>&1
tail -n 4 gpurun_out/r5_b16chain_tests.log; cat gpurun_out/r5_b16chain_old.txt gpurun_out/r5_b16chain_new.txt
for f in llff_old llff tile; do echo "== $f"; grep -E "total kernel|k_shade_bwd" gpurun_out/r5_b16chain_$f.txt | cut -c1-130; done
