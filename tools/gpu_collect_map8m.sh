#!/bin/bash
# usage (GPU box, repo root): tools/gpu_collect_map8m.sh <round-tag>
# memory-side counters of the two map-sized kernels at 8 M surfels on the current build: three rocprofv3 --pmc passes
# (FETCH_SIZE | WRITE_SIZE | L2 hits and misses: they do not fit one pass; counters only, no extra trace domains) over
# tools/map_kernels_8m.py -> gpurun_out/<tag>_pmc_map_kernels_8m.{json,md}; copy both to profiles/.
r=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=${r}_m8_$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex 'k_fuse_surfels|k_warp' --output-format csv -d $R/gpurun_out/pmc_$tag -o $tag -- \
      python $R/tools/map_kernels_8m.py > $R/gpurun_out/pmc_$tag.log 2>&1 < /dev/null
  tail -1 $R/gpurun_out/pmc_$tag.log | cut -c1-200
done
python $R/tools/pmc_map8m.py $R/gpurun_out $r
