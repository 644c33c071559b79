#!/bin/bash
# GPU trip 4 (round 5): the round's profiles from the shipped build (tools/gpu_collect.sh r05), a kernel trace of the
# kitti_like leg's launches, ten headline runs back to back
mkdir -p gpurun_out
tools/gpu_collect.sh r05 2>&1 | tail -30
tools/bench_variance.sh 10 gpurun_out/r05_bench_variance.md | tail -12
