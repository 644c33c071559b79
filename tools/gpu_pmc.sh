#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag> "<counters>" [bench args...]
# one rocprofv3 --pmc pass (counters only, no extra trace domains) over a short bench.py run
tag=$1; shift
ctrs=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout ${DSM_PROF_TIMEOUT:-120} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o $tag -- \
    python $R/bench.py --no-cpu-baseline --no-roofline --no-dropin "$@" > $R/gpurun_out/pmc_$tag.log 2>&1 < /dev/null
tail -1 $R/gpurun_out/pmc_$tag.log | cut -c1-300
python $R/tools/pmc_stats.py $R/gpurun_out/pmc_$tag $R/gpurun_out/pmc_$tag.md < /dev/null
