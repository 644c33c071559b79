#!/bin/bash
# usage (GPU box, repo root): tools/gpu_collect.sh <round-tag>
# everything profiles/ holds for a round, from the current build: PMC traffic of the launches the timed region makes
# (batched over 32 subsequences) and of the single launches (FETCH_SIZE and WRITE_SIZE in separate passes), SQ instruction
# counters of launches batched over 8 and over 32, kernel traces of the bench in its default configuration (four batches of 32 in
# flight), with one batch of 32 / of 8 at a time, and with one subsequence.  Copy what is wanted from gpurun_out/ into
# profiles/.
r=${1:-r04}
B8="--mode batched --streams 8 --batches 1 --steps 3 --warmup 1 --frames-per-step 16 --no-verify"
B32="--mode batched --streams 32 --batches 1 --steps 2 --warmup 1 --frames-per-step 16 --no-verify"
S1="--mode streams --streams 1 --steps 3 --warmup 1 --frames-per-step 16 --no-verify"
SQ="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
tools/gpu_pmc.sh ${r}b_fetch "FETCH_SIZE" $B32 > /dev/null
tools/gpu_pmc.sh ${r}b_write "WRITE_SIZE" $B32 > /dev/null
python tools/pmc_traffic.py gpurun_out/pmc_${r}b_fetch gpurun_out/pmc_${r}b_write gpurun_out/${r}_pmc_traffic_batched.json \
    "bench.py --no-cpu-baseline --no-roofline --no-dropin $B32" 32 | grep -E "update_seeds|seed_fit|seed_points|seed_stats|pixel_normals|apply"
tools/gpu_pmc.sh ${r}_fetch "FETCH_SIZE" $S1 > /dev/null
tools/gpu_pmc.sh ${r}_write "WRITE_SIZE" $S1 > /dev/null
python tools/pmc_traffic.py gpurun_out/pmc_${r}_fetch gpurun_out/pmc_${r}_write gpurun_out/${r}_pmc_traffic.json \
    "bench.py --no-cpu-baseline --no-roofline --no-dropin $S1" | grep -E "update_seeds|seed_fit|seed_points|seed_stats|pixel_normals"
tools/gpu_pmc.sh ${r}b_sq "$SQ" $B8 > /dev/null
tools/gpu_pmc.sh ${r}b32_sq "$SQ" $B32 > /dev/null
tools/gpu_profile.sh ${r}_b8x1 $B8 | tail -1 | cut -c1-200
tools/gpu_profile.sh ${r}_b32x1 $B32 | tail -1 | cut -c1-200
tools/gpu_profile.sh ${r}_b32x4 --steps 4 --warmup 2 --no-verify | tail -1 | cut -c1-200
tools/gpu_profile.sh ${r}_s1 $S1 | tail -1 | cut -c1-200
