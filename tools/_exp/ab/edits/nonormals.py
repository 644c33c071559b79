"""CEILING experiment, results are WRONG on purpose: what would deleting the 12 B/pixel normal plane buy if the normals were
free?  k_pixel_normals is not launched and k_seed_stats skips its second walk (every fitted seed starts the plane fit from
the normal (0, 0, 1): the fit runs its five steps all the same)."""
import sys, os
d = sys.argv[1]
p = os.path.join(d, "dsm_kernels.hip")
s = open(p).read()
old = "        hipLaunchStage(k_pixel_normals<true>, k_pixel_normals<true>, g_pix4, dim3(256));\n"
assert old in s
s = s.replace(old, "")
open(p, "w").write(s)
p = os.path.join(d, "dsm_k_planes.h")
s = open(p).read()
old = "    if (__ballot(fit) != 0) {\n        // ---- second walk"
assert old in s
s = s.replace(old, "    nz = 1.0f;\n    if (false) {\n        // ---- second walk")
open(p, "w").write(s)
