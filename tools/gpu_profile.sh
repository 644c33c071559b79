#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag> [bench args...]
# kernel trace of one bench.py run -> gpurun_out/prof_<tag>/ + gpurun_out/prof_<tag>.md
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout ${DSM_PROF_TIMEOUT:-120} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- \
    python $R/bench.py --no-cpu-baseline --no-roofline --no-dropin "$@" > $R/gpurun_out/prof_$tag.log 2>&1 < /dev/null
tail -1 $R/gpurun_out/prof_$tag.log
python $R/tools/kernel_stats.py $R/gpurun_out/prof_$tag $R/gpurun_out/prof_$tag.md < /dev/null
