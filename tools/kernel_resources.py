#!/usr/bin/env python3
"""Registers, scratch, LDS and the occupancy the compiler reports for every kernel of the product library (no GPU needed):

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.md

hipcc -Rpass-analysis=kernel-resource-usage over csrc/dsm_kernels.hip with the library's own flags (build.py)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def main():
    src = os.path.join(ROOT, "densesurfelmapping_amd", "csrc", "dsm_kernels.hip")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
                            "-Wno-unused-value", "-c", src, "-o", os.path.join(tmp, "k.o"), "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-2000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = subprocess.run(["c++filt"], input="\n".join(b.split("\n")[0].strip() for b in blocks), capture_output=True, text=True).stdout.split("\n")
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD | LDS B/workgroup |\n|---|---|---|---|---|---|---|")
    for b, name in zip(blocks, names):
        vals = [(re.search(re.escape(k) + r": (\S+)", b) or [None, "?"])[1] for k in KEYS]
        print("| `" + name.split("(")[0].replace("void ", "").replace("dsm::", "") + "` | " + " | ".join(vals) + " |")


if __name__ == "__main__":
    main()
