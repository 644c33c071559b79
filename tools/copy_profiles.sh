#!/bin/bash
# usage (this container, repo root): tools/copy_profiles.sh <round-tag>
# what tools/gpu_collect.sh <round-tag> (+ tools/bench_variance.sh) left in gpurun_out/ -> profiles/, under the names bench.py,
# the tests and DESIGN.md use
r=${1:-r05}
cp gpurun_out/prof_${r}_b32x1.md profiles/${r}_kernel_trace_batch32x1.md
cp gpurun_out/prof_${r}_b32x4.md profiles/${r}_kernel_trace_batch32x4_default.md
cp gpurun_out/prof_${r}_b8x1.md profiles/${r}_kernel_trace_batch8x1.md
cp gpurun_out/prof_${r}_s1.md profiles/${r}_kernel_trace_streams1.md
cp gpurun_out/pmc_${r}b_fetch.md profiles/${r}_pmc_FETCH_SIZE_batch32.md
cp gpurun_out/pmc_${r}b_write.md profiles/${r}_pmc_WRITE_SIZE_batch32.md
cp gpurun_out/pmc_${r}b32_sq.md profiles/${r}_pmc_sq_batch32.md
cp gpurun_out/pmc_${r}b_sq.md profiles/${r}_pmc_sq_batch8.md
cp gpurun_out/${r}_pmc_traffic.json gpurun_out/${r}_pmc_traffic_batched.json profiles/
[ -f gpurun_out/${r}_bench_variance.md ] && cp gpurun_out/${r}_bench_variance.md profiles/
ls -la profiles/${r}_*
