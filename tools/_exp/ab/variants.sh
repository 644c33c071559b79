#!/bin/bash
# usage (GPU box, repo root): tools/_exp/ab/variants.sh <kernel-name-pattern> lib1.so lib2.so ...
# per library: the average duration of the matching kernels in a short batch-of-32 trace (one batch alone on the GPU)
pat=$1; shift
L=densesurfelmapping_amd/libdsm_hip.so
cp $L /tmp/var_keep.so
for lib in base "$@"; do
  [ "$lib" = base ] || cp $lib $L
  tag=var_$(basename $lib .so)
  tools/gpu_profile.sh $tag --mode batched --streams 32 --batches 1 --steps 2 --warmup 1 --frames-per-step 16 --no-verify > /dev/null 2>&1
  echo "== $lib"; grep -E "$pat" gpurun_out/prof_$tag.md | cut -c1-110
  cp /tmp/var_keep.so $L
done
