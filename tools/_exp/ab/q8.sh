run() { python bench.py --no-cpu-baseline --no-dropin --no-roofline --no-verify "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"],1))'; }
echo "default 32x4: $(run --steps 20 --warmup 5)"
echo "128x4: $(run --steps 8 --warmup 3 --streams 128 --batches 4)"
cp densesurfelmapping_amd/libdsm_hip.so /tmp/keep.so; cp tools/_exp/ab/libdsm_hip_q8.so densesurfelmapping_amd/libdsm_hip.so
export GPU_MAX_HW_QUEUES=8
echo "q8 64x8: $(run --steps 10 --warmup 3 --streams 64 --batches 8 --host-threads 8)"
echo "q8 128x8: $(run --steps 8 --warmup 3 --streams 128 --batches 8 --host-threads 8)"
echo "q8 32x4: $(run --steps 20 --warmup 5)"
unset GPU_MAX_HW_QUEUES
echo "q4(8 streams) 64x8: $(run --steps 10 --warmup 3 --streams 64 --batches 8 --host-threads 8)"
cp /tmp/keep.so densesurfelmapping_amd/libdsm_hip.so
echo "default again 32x4: $(run --steps 20 --warmup 5)"
