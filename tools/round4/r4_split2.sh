#!/bin/bash
# round 4: what bounds k_shade_scatter -- occupancy variants, no-atomics variant, WRITE_SIZE (atomic segments) per run length
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export JT_NO_AUX=1
B="$R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras"
run() { # tag, env...
  tag=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4s2_$tag -o k -- python3 $B > $O/r4s2_$tag.log 2>&1
  python3 $R/tools/prof_summary.py $O/r4s2_$tag/k_kernel_stats.csv 6 27 > $O/r4s2_${tag}_summary.txt
  echo "== $tag"; grep -E "k_shade_scatter|k_shade_bwd|k_march_bwd_walk" $O/r4s2_${tag}_summary.txt | cut -c1-130
  rm -rf $O/r4s2_$tag/*kernel_trace.csv
}
V=$R/joint_tensorf_amd/lib/variants
run occ2_16 JT_BWD_SPLIT=16 JT_LIB_PATH=$V/occ2.so
run occ4_16 JT_BWD_SPLIT=16 JT_LIB_PATH=$V/occ4.so
run occ4_8 JT_BWD_SPLIT=8 JT_LIB_PATH=$V/occ4.so
run noatom_16 JT_BWD_SPLIT=16 JT_LIB_PATH=$V/noatom.so
run noatom_32 JT_BWD_SPLIT=32 JT_LIB_PATH=$V/noatom.so
run noatom_0 JT_BWD_SPLIT=0 JT_LIB_PATH=$V/noatom.so
run bpc1_16 JT_BWD_SPLIT=16 JT_SCATTER_BLOCKS_PER_CU=1
run bpc2_16 JT_BWD_SPLIT=16 JT_SCATTER_BLOCKS_PER_CU=2
for s in 0 8 16 32; do
  JT_BWD_SPLIT=$s rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r4s2_pmcW_$s -o p -- python3 $B > $O/r4s2_pmcW_$s.log 2>&1
  echo "== WRITE_SIZE split=$s"; python3 $R/tools/pmc_summary.py $O/r4s2_pmcW_$s/p_counter_collection.csv k_shade_scatter k_shade_bwd k_march_bwd_walk | cut -c1-160
  rm -rf $O/r4s2_pmcW_$s/*kernel_trace.csv
done
