"""CEILING experiment, results are WRONG on purpose: k_update_seeds at the occupancy a one-byte position list would allow.
The LDS rows are cut to 60 (16 KB per wave: the registers' eight waves per CU instead of five); a list's length is counted
separately and the first Huber pass still runs its true length (elements beyond row 60 read row 60), so the instruction
count is that of the shipped kernel plus one add per pixel; no seed takes the overflow tier."""
import sys, os
d = sys.argv[1]
p = os.path.join(d, "dsm_k_superpixel.h")
s = open(p).read()
def rep(old, new, cnt=1):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, cnt)
rep("    constexpr int CAP = kLaneCap - 3; // the longest list kept here", "    constexpr int CAP = 60; // CEILING EXPERIMENT (was kLaneCap - 3)")
rep("    unsigned tail = lane4;\n", "    unsigned tail = lane4;\n    int nd_true = 0;\n")
rep("                tail += dv ? 256u : 0u;\n", "                tail += dv ? 256u : 0u;\n                nd_true += dv ? 1 : 0;\n")
rep('asm volatile("" : "+v"(acc_ci), "+v"(sum), "+v"(tail),', 'asm volatile("" : "+v"(acc_ci), "+v"(sum), "+v"(tail), "+v"(nd_true),')
rep("    const int nd = (int)((tail - lane4) >> 8);\n", "    const int nd = nd_true;\n")
rep("    const bool over = stats && nd >= CAP && !settled;", "    const bool over = false;")
rep("s_list[(i + t) * 64 + lane]);", "s_list[((i + t) < CAP ? (i + t) : CAP) * 64 + lane]);")
open(p, "w").write(s)
