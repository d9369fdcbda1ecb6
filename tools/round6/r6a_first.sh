set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mfma or fulltape" 2>&1 | tail -15 > gpurun_out/r6a_parity.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stage4_sharp_400cube or llff_final_grid" 2>&1 | tail -15 > gpurun_out/r6a_fullsize.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-extras --no-live-pmc"
for i in 1 2; do
$B > gpurun_out/r6a_bench_lean_$i.json 2> gpurun_out/r6a_bench_lean_$i.err
JT_LEAN_TAPE=0 $B > gpurun_out/r6a_bench_full_$i.json 2>/dev/null
JT_SCATTER_WGS=256 $B > gpurun_out/r6a_bench_lean_wg256_$i.json 2>/dev/null
JT_SCATTER_WGS=224 $B > gpurun_out/r6a_bench_lean_wg224_$i.json 2>/dev/null
done
JT_NO_AUX=1 $B > gpurun_out/r6a_bench_lean_noaux.json 2>/dev/null
JT_NO_AUX=1 JT_LEAN_TAPE=0 $B > gpurun_out/r6a_bench_full_noaux.json 2>/dev/null
$B --config bat_llff_VM_MLP > gpurun_out/r6a_bench_llff_lean.json 2>/dev/null
JT_LEAN_TAPE=0 $B --config bat_llff_VM_MLP > gpurun_out/r6a_bench_llff_full.json 2>/dev/null
