#!/bin/bash
# usage (GPU box, repo root): tools/gpu_collect.sh <round-tag>
# everything profiles/ holds for a round, from the current build: PMC traffic of the batched and of the single
# launches (FETCH_SIZE and WRITE_SIZE in separate passes), SQ instruction counters, kernel traces of the bench in its
# default (batched) configuration, with one batch at a time, and with one subsequence.  Copy what is wanted from
# gpurun_out/ into profiles/.
r=${1:-r03}
B8="--mode batched --streams 8 --batches 1 --steps 3 --warmup 1 --frames-per-step 16"
S1="--mode streams --streams 1 --steps 3 --warmup 1 --frames-per-step 16"
SQ="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
tools/gpu_pmc.sh ${r}b_fetch "FETCH_SIZE" $B8 > /dev/null
tools/gpu_pmc.sh ${r}b_write "WRITE_SIZE" $B8 > /dev/null
python tools/pmc_traffic.py gpurun_out/pmc_${r}b_fetch gpurun_out/pmc_${r}b_write gpurun_out/${r}_pmc_traffic_batched.json \
    "bench.py --no-cpu-baseline --no-roofline --no-dropin $B8" 8 | grep -E "update_seeds|seed_fit|seed_points|seed_stats|pixel_normals"
tools/gpu_pmc.sh ${r}_fetch "FETCH_SIZE" $S1 > /dev/null
tools/gpu_pmc.sh ${r}_write "WRITE_SIZE" $S1 > /dev/null
python tools/pmc_traffic.py gpurun_out/pmc_${r}_fetch gpurun_out/pmc_${r}_write gpurun_out/${r}_pmc_traffic.json \
    "bench.py --no-cpu-baseline --no-roofline --no-dropin $S1" | grep -E "update_seeds|seed_fit|seed_points|seed_stats|pixel_normals"
tools/gpu_pmc.sh ${r}_sq "$SQ" $S1 > /dev/null
tools/gpu_pmc.sh ${r}b_sq "$SQ" $B8 > /dev/null
tools/gpu_profile.sh ${r}_b8x1 $B8 | tail -1 | cut -c1-200
tools/gpu_profile.sh ${r}_b8x4 --steps 6 --warmup 2 | tail -1 | cut -c1-200
tools/gpu_profile.sh ${r}_s1 $S1 | tail -1 | cut -c1-200
