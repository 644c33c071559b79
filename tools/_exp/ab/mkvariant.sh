#!/bin/bash
# usage (this container, repo root): tools/_exp/ab/mkvariant.sh <name> <edit.py>
# A variant of the library without touching the tree: csrc/ is copied to /tmp/var/<name>, <edit.py> (a python script that
# takes the directory as its argument and rewrites what it wants there) is applied, and the result is built as
# tools/_exp/ab/libdsm_hip_<name>.so (git-ignored; it travels with the next gpurun call).  Prints the registers, LDS and
# static instruction mix of the kernel named by $KERNEL (mangled-name fragment; default k_update_seedsILb1E).
# Time the variants against the in-tree build on ONE box: tools/_exp/ab/variants.sh "<kernel-name pattern>" lib1.so lib2.so ...
set -e
name=$1; ed=$2; kernel=${KERNEL:-k_update_seedsILb1E}
root=$(cd "$(dirname "$0")/../../.." && pwd)
d=/tmp/var/$name; rm -rf $d; mkdir -p $d
cp $root/densesurfelmapping_amd/csrc/* $d/
sed -i "s#\"../../include/#\"$root/include/#" $d/*.h $d/*.cpp $d/*.hip
python3 $ed $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function -Wno-unused-value \
    $d/dsm_kernels.hip $d/dsm_api.hip $d/dsm_surfel_map.cpp -o $root/tools/_exp/ab/libdsm_hip_$name.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o $d/kern.s $d/dsm_kernels.hip 2>/dev/null
python3 - $d/kern.s $kernel <<'PY'
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)', s):
    name = m.group(1)
    if sys.argv[2] not in name:
        continue
    blk = s[m.start():m.start() + 6000]
    g = lambda key: (re.search(key + r'\s+(\d+)', blk) or [None, None])[1]
    i = s.index('\n' + name + ':'); j = s.index('s_endpgm', i)
    lines = [l.strip() for l in s[i:j].split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter(l.split()[0].split('_')[0] for l in lines if not l.endswith(':'))
    print(name, 'vgpr', g('next_free_vgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'), dict(c))
PY
