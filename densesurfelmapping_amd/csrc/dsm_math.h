// dsm_math.h -- the per-element arithmetic of the surfel-fusion hot path, written once and used
// by every HIP kernel in dsm_kernels.hip (and, compiled for the host, by tests/hostemu.cpp,
// which checks it against the oracle without a GPU).
//
// Every expression is typed exactly as the reference's C++ evaluates it (usual arithmetic
// conversions, double literals, no FMA): the translation unit must be built with
// -ffp-contract=off and hipcc's default correctly-rounded fp32 divide/sqrt.
// "FF.cpp" = surfel_fusion/src/fusion_functions.cpp of the reference.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define DSM_HD __host__ __device__ __forceinline__
#else
#define DSM_HD static inline
#endif

namespace dsm {

constexpr int kCell = 8;          // SP_SIZE        fusion_functions.h:10
constexpr int kSweeps = 3;        // ITERATION_NUM  fusion_functions.h:8
constexpr int kWorkers = 10;      // THREAD_NUM     fusion_functions.h:9
constexpr double kAngleCos = 0.1; // MAX_ANGLE_COS  fusion_functions.h:11

struct Intrinsics {
    float fx, fy, cx, cy;
};

// A float against a double constant, compared in fp32.  The reference compares in double -- its thresholds are
// double literals (0.1, 0.4 ...: none of them a float), so `x > 0.1` is `(double)x > 0.1` -- and on this hardware
// the conversion alone costs four fp32 operations.  For a float x and c > 0 the outcome is that of a compare with
// the float neighbouring c on the side of the relation:
//     (double)x >  c  <=>  x >  flt_below(c)      (double)x <  c  <=>  x <  flt_above(c)
//     (double)x <= c  <=>  x <= flt_below(c)      (double)x >= c  <=>  x >= flt_above(c)
// flt_below(c) = the largest float <= c, flt_above(c) = the smallest float >= c: every float is a double, and
// between those two there is no float, so x > c means x >= flt_above(c) > flt_below(c), and x > flt_below(c) means
// x >= the next float up, which is > c (or c itself is a float and both are c).  NaN compares false either way, and
// -c mirrors.  (tests/test_cpu.py walks the floats around every threshold used.)
DSM_HD constexpr float flt_below(double c) {
    const float f = (float)c; // round to nearest: off by at most one float
    return (double)f <= c ? f : __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f) - 1u);
}
DSM_HD constexpr float flt_above(double c) {
    const float f = (float)c;
    return (double)f >= c ? f : __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f) + 1u);
}

// x / 100.0, correctly rounded, in three operations instead of a full fp64 divide expansion:
// q0 = RN(x*r), e = x - 100*q0 (exact in one FMA), q = RN(q0 + e*r) with r = RN(1/100) is the
// correctly rounded quotient (Markstein's theorem; 100 = 1.5625*2^6 has no all-ones significand).
// Checked against x/100.0 on 3e8 floats spanning all exponents (tests/test_cpu.py runs a sample).
DSM_HD double div_by_100(double x) {
    const double r = 1.0 / 100.0;
    const double q0 = x * r;
    const double e = __builtin_fma(-100.0, q0, x);
    return __builtin_fma(e, r, q0);
}

// ---------------------------------------------------------------- SLIC cost, FF.cpp:364-387
// Seed side: x, y, mean intensity, and the double 1.0/mean_depth (valid iff has_depth).
// Returns whether the depth term applied.
DSM_HD bool pixel_cost(float sx, float sy, float si, bool seed_has_depth, double seed_inv_depth, float pix_i,
                       float pix_invd, int x, int y, float &no_d, float &with_d) {
    float ddx = sx - (float)x, ddy = sy - (float)y;
    float dist = ddx * ddx + ddy * ddy;
    float cost = 0.0f;
    cost += dist / (float)((kCell / 2) * (kCell / 2));
    float di = si - pix_i;
    cost = (float)((double)cost + div_by_100((double)(di * di))); // FF.cpp:376: (double)(di*di) / 100.0
    no_d = cost;
    with_d = cost;
    if (seed_has_depth && pix_invd > 0) {
        float dd = (float)(seed_inv_depth - (double)pix_invd);      // FF.cpp:380
        with_d = (float)((double)cost + (double)(dd * dd) * 400.0); // FF.cpp:381
        return true;
    }
    return false;
}

// inverse depth of a pixel, FF.cpp:404-405
// (float)(1.0 / (double)d): rounding a quotient of two 24-bit values first to 53 >= 2*24+2 bits and then
// to 24 bits equals rounding it once (Figueroa), so the correctly rounded fp32 divide gives the same bits.
DSM_HD float pixel_inv_depth(float d) {
    float invd = 0.0f;
    if (d > flt_below(0.01)) invd = 1.0f / d; // (double)d > 0.01
    return invd;
}

// Pixels right of / below the last cell centre by 4 or more have no candidate cell at all (only when
// (size mod 8) > 4, e.g. KITTI's 1242x375): the reference labels them -1 and then reads and writes
// superpixel_seeds[-1] (FF.cpp:400,442-451,242), memory in front of the vector.  Policy here (and in the
// C restatement used by the tests): label -1, member of no superpixel, never stable; fusion treats seed -1 as an all-zero
// record (the free-space test still applies, then the surfel is skipped).
DSM_HD bool has_candidate_cell(int x, int y, int gw, int gh) { return x < gw * kCell + kCell / 2 && y < gh * kCell + kCell / 2; }

// Pick the seed of pixel (x,y): candidates are the <=2x2 in-grid cells whose centre is closer
// than one cell in both axes, visited x-offset outer / y-offset inner, strict '<' (FF.cpp:413-451).
// load(gx, gy, sx, sy, si, has_depth, inv_depth) fetches the cost-side state of grid cell (gx,gy).
// Returns -1 when every candidate cost is >= the 1e6 sentinel (or there is no candidate).
template <typename LoadSeed>
DSM_HD int pick_seed(int x, int y, float pix_i, float pix_d, int gw, int gh, LoadSeed load) {
    const float invd = pixel_inv_depth(pix_d);
    const int bx = x / kCell, by = y / kCell;
    float best_d = 1e6f, best_n = 1e6f;
    int arg_d = -1, arg_n = -1;
    bool all_depth = true;
    // Of the reference's 3x3 offsets at most two per axis pass its distance filter |8g+4 - x| < 8: the pixel's own
    // cell, and the lower neighbour when x mod 8 < 4 or the upper one when x mod 8 > 4 (at x mod 8 == 4 both
    // neighbour centres are exactly 8 away: neither).  Visiting exactly those, lower cell first, is the reference's
    // order (x-offset outer, y-offset inner, ascending) without the offsets it skips -- and without lanes of one wave
    // disagreeing about which of three iterations are live.
    const int xr = x % kCell, yr = y % kCell;
    const int gx0 = bx - (xr < kCell / 2 ? 1 : 0), gy0 = by - (yr < kCell / 2 ? 1 : 0);
    for (int jx = 0; jx < 2; jx++) {
        const int gx = gx0 + jx;
        if (!(gx >= 0 && gx < gw) || (jx == 1 && xr == kCell / 2)) continue;
        for (int jy = 0; jy < 2; jy++) {
            const int gy = gy0 + jy;
            if (!(gy >= 0 && gy < gh) || (jy == 1 && yr == kCell / 2)) continue;
            const int s = gy * gw + gx;
            float sx, sy, si;
            bool has_d;
            double inv_d;
            load(gx, gy, sx, sy, si, has_d, inv_d);
            float cn, cd;
            const bool with = pixel_cost(sx, sy, si, has_d, inv_d, pix_i, invd, x, y, cn, cd);
            all_depth = all_depth && with;
            if (cd < best_d) { best_d = cd; arg_d = s; }
            if (cn < best_n) { best_n = cn; arg_n = s; }
        }
    }
    return all_depth ? arg_d : arg_n; // FF.cpp:442-451
}

// The same pick, filtered: what pick_seed must return is an argmin, not the costs, and the costs above are expensive
// because of their typing (per candidate seven float <-> double conversions and seven double operations, all slower than
// fp32 on this hardware).  pick_seed_fast evaluates every candidate's cost in plain fp32 (fused multiply-adds allowed)
// together with a bound on how far that value can be from the reference's, and answers only when the smallest interval
// lies strictly below all the others and below the 1e6 sentinel: then the reference's strict '<' scan picks the same
// candidate whatever the order.  Otherwise it returns kPickUnsure and the caller runs pick_seed.
//
// Round 6: the filter is laid out for PACKED fp32 -- gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 for a whole
// wave in the slot of their scalar forms, so the four candidates of a pixel are two register pairs -- with every factor
// that does not depend on the pixel folded into the candidate once per column of four pixels, and the pixel's inverse
// depth from v_rcp_f32 (1 ulp; the reference's correctly rounded quotient costs ten more instructions) with its error
// inside the bound.  Per candidate (xf, yq = y / 4, pi, p20 = 20 * rcp(depth) are the pixel's):
//     ax16   = RN(RN(sx - xf)^2) / 16                     once per column
//     dist16 = fma(sy/4 - yq, sy/4 - yq, ax16)            = RN((sy - y)^2 + RN((sx - x)^2)) / 16
//     t      = RN(RN(si - pi) * 0.1f),  c_no = fma(t, t, dist16)
//     dd20   = fma(s20, m, m ? -p20 : 0)                 s20 = RN32(20 / mean_depth) (0 without one), m = 1 (0: no depth term)
//     c      = fma(dd20, dd20, c_no)
//
// Error bound (u = 2^-24; the double roundings of the reference are 2^-29 of that and ignored).  All three cost terms are
// >= 0, so sums do not cancel.  Spatial term: the reference rounds (sy - y)^2 before the add, here it is exact inside the
// FMA: <= 3u relative.  Intensity term: the reference's RN32(di^2) / 100 against t^2 = 0.01 di^2 (1 + <=u)^4: <= 5u.  The
// two final roundings: <= 2u.  So |c_no~ - c_no| <= 8u c_no~.  Depth term, exact value 400 dd^2 with dd = RN32(D - p),
// D = 1.0 / mean_depth (double), p = RN32(1 / depth): s20 differs from 20 D by u |s20|; the pixel side from 20 p by
// 20 |rcp - p| + u |p20| <= 4u |p20| + 20 * 2^-126 (v_rcp_f32 is within one ulp of the correctly rounded quotient --
// tests/test_gpu_parity.py walks every float -- and flushes a denormal quotient to 0); the subtraction and the reference's own
// rounding of dd one u |dd20| each; |p20| <= |s20| + |dd20|:
//     |dd20~ - 20 dd| <= e20 = u (5 |s20| + 6.1 |dd20~|) + 2.4e-37,   |dd20~^2 - 400 dd^2| <= e20 (2 |dd20~| + e20),
// the reference's RN32(dd^2) and the last add 2u more.  In all
//     |c~ - cost| <= 11u c~ + e20 (2 |dd20~| + e20),
// used below with 20u (4u of it for the candidate tag in the key's low bits) and e2 = 2 e20 in both factors, evaluated with
// e2 = u (12 |s20| + 16 |dd20|) + 2e-36: that also covers the roundings of the bound's own evaluation.
// tests/hostemu.cpp checks the bound against the reference costs on every candidate of its test frames, with the pixel's
// inverse depth moved off the correctly rounded one by a hash-chosen ulp either way.
struct FastCost {
    float c, err;
};
constexpr float kFastU = 5.9604645e-8f; // u = 2^-24

// ---- two floats per operation.  Device: a register pair and one packed instruction; host (tests/hostemu.cpp): the same
// IEEE operations component by component -- identical bits.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f32x2 __attribute__((ext_vector_type(2)));
DSM_HD f32x2 f2_make(float a, float b) { f32x2 r; r.x = a; r.y = b; return r; }
DSM_HD f32x2 f2_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
DSM_HD f32x2 f2_mul(f32x2 a, f32x2 b) { return a * b; }
DSM_HD f32x2 f2_sub(f32x2 a, f32x2 b) { return a - b; }
// 1 / d by the hardware's reciprocal: within one ulp of the correctly rounded quotient, denormal quotients flushed to 0
DSM_HD float rcp_1ulp(float d) { return __builtin_amdgcn_rcpf(d); }
#else
struct f32x2 {
    float x, y;
};
DSM_HD f32x2 f2_make(float a, float b) { return f32x2{a, b}; }
DSM_HD f32x2 f2_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
DSM_HD f32x2 f2_mul(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
DSM_HD f32x2 f2_sub(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
// the host's stand-in for v_rcp_f32: the correctly rounded quotient moved by one ulp up, down or not at all (a hash of the
// operand decides), denormal quotients flushed -- every value the hardware may return lies within this model's reach
DSM_HD float rcp_1ulp(float d) {
    const float q = 1.0f / d;
    uint32_t b = __builtin_bit_cast(uint32_t, q);
    if ((b & 0x7f800000u) == 0u) return 0.0f;
    uint32_t hsh = __builtin_bit_cast(uint32_t, d) * 2654435761u;
    hsh ^= hsh >> 15;
    const uint32_t k = hsh % 3u;
    if (q == q && (b & 0x7f800000u) != 0x7f800000u) b += k == 1u ? 1u : (k == 2u ? 0xffffffffu : 0u);
    return __builtin_bit_cast(float, b);
}
#endif
DSM_HD f32x2 f2_splat(float a) { return f2_make(a, a); }

// the pixel side of the depth term: has = the reference's `inv_d > 0` (FF.cpp:404-405,378: depth > 0.01 and a quotient that
// does not round to 0, i.e. a finite depth), p20 = 20 / depth for the filter
DSM_HD bool pixel_has_inv_depth(float d) { return d > flt_below(0.01) && d < __builtin_inff(); }
DSM_HD float pixel_p20(float d) { return 20.0f * rcp_1ulp(d); }
// the seed side: 20 / mean_depth rounded once to float; 0 for a seed without a mean depth (its candidates never take the depth
// term: PickCol::depth_ok) so that no infinity enters the arithmetic of a pixel that does not use it
DSM_HD float seed_s20(float mean_depth, double inv_depth) { return mean_depth > 0 ? (float)(20.0 * inv_depth) : 0.0f; }

// the depth term's share of the bound: e2 (2 |dd20| + e2), e2 = 16u |dd20| + df, df = 12u max |s20| + 2e-36.  Monotone in both
// arguments (every operand is >= 0 and rounding is monotone): evaluated at the largest |dd20| and the largest |s20| of several
// candidates it bounds every one of them.
DSM_HD float fast_cost_df(float s20_abs_max) { return __builtin_fmaf(12.0f * kFastU, s20_abs_max, 2e-36f); }
DSM_HD float fast_cost_slack(float ad20, float df) {
    const float e2 = __builtin_fmaf(16.0f * kFastU, ad20, df);
    return e2 * __builtin_fmaf(2.0f, ad20, e2);
}
// bound on |cost~ - cost| (20u: 4u of it for the candidate tag of pick_seed_fast); monotone in c and in the slack
DSM_HD float fast_cost_err(float c, float slack) { return __builtin_fmaf(20.0f * kFastU, c, slack); }

// one candidate with its own bound, in the filter's own operations: what the proof above is about, and what
// tests/hostemu.cpp checks against the reference's costs on every candidate.  p20 = pixel_p20(depth).
DSM_HD FastCost pixel_cost_fast(float sx, float sy, float si, float s20, bool with_depth, float pix_i, float p20, int x, int y) {
    const float ddx = sx - (float)x;
    const float ax16 = (ddx * ddx) * 0.0625f;
    const float ddyq = sy * 0.25f - (float)y * 0.25f;
    const float dist16 = __builtin_fmaf(ddyq, ddyq, ax16);
    const float t = (si - pix_i) * 0.1f;
    const float c_no = __builtin_fmaf(t, t, dist16);
    const float dd20 = __builtin_fmaf(s20, with_depth ? 1.0f : 0.0f, with_depth ? -p20 : 0.0f);
    FastCost r;
    r.c = __builtin_fmaf(dd20, dd20, c_no);
    r.err = fast_cost_err(r.c, fast_cost_slack(fabsf(dd20), fast_cost_df(fabsf(s20))));
    return r;
}

// The candidates of a pixel are those of its 4 x 4 quadrant of a cell (see pick_seed), and a thread works down a column of
// four pixels: PickCol holds, for the column x and all rows y' with y' / 4 == y / 4, the four candidates (k = (x offset
// k >> 1, y offset k & 1), the reference's scan order) as two pairs -- (0, 1) and (2, 3) -- with everything folded in that
// does not depend on the row.  load(k, gx, gy, sx4, sy4, si, seed_depth, s20) fetches the cost-side state of grid cell
// (gx, gy) -- which may lie OUTSIDE the grid by one cell: the loader then returns the state of the nearest cell inside (the
// kernel's LDS tile keeps such copies in its halo; the candidate is out of play and only has to be finite-or-harmless) --
// as sx / 4, sy / 4 (exact scalings), the mean intensity, the mean depth and s20 = seed_s20(...).
// A candidate out of play (outside the grid, or past the reference's distance filter |8g + 4 - x| < 8) carries a penalty of
// 1e30 in its spatial term: it sorts behind every real cost, and below only a pick under the reference's 1e6 sentinel counts.
struct PickCol {
    f32x2 ax16[2];      // (sx - x)^2 / 16 (+ penalty)
    f32x2 ax16_edge[2]; // the same for a row with y mod 8 == 4, where the upper neighbour row is out of play too (FF.cpp:420-422)
    f32x2 sy4[2], si[2], s20[2];
    bool depth_ok[2]; // [row offset]: both candidates of that row either out of play or with a mean depth
    float df;         // fast_cost_df over the four candidates
    int base;         // seed index of candidate 0
};
constexpr float kPickPenalty = 1e30f;
template <typename LoadSeedF> DSM_HD PickCol pick_col(int x, int y, int gw, int gh, LoadSeedF load) {
    PickCol q;
    const int xr = x & (kCell - 1), yr = y & (kCell - 1); // (x, y >= 0)
    const int gx0 = (x >> 3) - (xr < kCell / 2 ? 1 : 0), gy0 = (y >> 3) - (yr < kCell / 2 ? 1 : 0);
    static_assert(kCell == 8, "the shifts above");
    q.base = gy0 * gw + gx0;
    const bool col_ok[2] = {gx0 >= 0, gx0 + 1 < gw && xr != kCell / 2}; // (gx0 < gw and gx0 + 1 >= 0 always)
    const bool row_in[2] = {gy0 >= 0, gy0 + 1 < gh};
    float sx4[4], sy4[4], si[4], sd[4], s20[4];
    for (int k = 0; k < 4; k++) load(k, gx0 + (k >> 1), gy0 + (k & 1), sx4[k], sy4[k], si[k], sd[k], s20[k]);
    const f32x2 xq = f2_splat((float)x * 0.25f);
    for (int j = 0; j < 2; j++) {
        const f32x2 ddx4 = f2_sub(f2_make(sx4[2 * j], sx4[2 * j + 1]), xq);
        const f32x2 a = f2_mul(ddx4, ddx4);
        const bool in0 = col_ok[j] && row_in[0], in1 = col_ok[j] && row_in[1];
        q.ax16[j] = f2_make(in0 ? a.x : kPickPenalty, in1 ? a.y : kPickPenalty);
        q.ax16_edge[j] = f2_make(in0 ? a.x : kPickPenalty, kPickPenalty);
        q.sy4[j] = f2_make(sy4[2 * j], sy4[2 * j + 1]);
        q.si[j] = f2_make(si[2 * j], si[2 * j + 1]);
        q.s20[j] = f2_make(s20[2 * j], s20[2 * j + 1]);
    }
    q.df = fast_cost_df(fmaxf(fmaxf(fabsf(s20[0]), fabsf(s20[1])), fmaxf(fabsf(s20[2]), fabsf(s20[3]))));
    for (int j = 0; j < 2; j++)
        q.depth_ok[j] = !row_in[j] | ((!col_ok[0] | (sd[j] > 0)) & (!col_ok[1] | (sd[2 + j] > 0))); // (no short cuts: lane masks)
    return q;
}
struct FastPickTrace { // what a host-side check wants to see of a pick (tests/hostemu.cpp); the kernels pass none
    float err;
    bool all_depth;
};
struct FastPick {
    int seed;  // the reference's pick ...
    bool sure; // ... if the bounds separate it from the runner-up; otherwise the caller runs pick_seed
};
DSM_HD FastPick pick_seed_fast(const PickCol &q, int y, float pix_i, float pix_d, int gw, FastPickTrace *trace = nullptr) {
    const bool edge = (y & (kCell - 1)) == kCell / 2;
    // every live candidate has a mean depth (and the pixel a depth): per row offset that is known for the whole quadrant
    // (depth_ok); the upper row is out of play altogether for a pixel on the filter's edge
    const bool all_depth = pixel_has_inv_depth(pix_d) && q.depth_ok[0] && (q.depth_ok[1] || edge);
    // FF.cpp:442-451: with every candidate's depth term applied the pick is the argmin with it, else the argmin without
    // (m = 0 switches the term off: s20 * 0 - 0 = 0, every s20 being finite or the cost infinite and the pick open; the
    // pixel's side by a select, 20 / depth being anything at all for a pixel without a depth).
    const f32x2 mm = f2_splat(all_depth ? 1.0f : 0.0f), p20m = f2_splat(all_depth ? -pixel_p20(pix_d) : 0.0f);
    const f32x2 yq = f2_splat((float)y * 0.25f), pi = f2_splat(pix_i), tenth = f2_splat(0.1f);
    f32x2 c[2], dd20[2];
    for (int j = 0; j < 2; j++) {
        const f32x2 ddyq = f2_sub(q.sy4[j], yq);
        const f32x2 dist16 = f2_fma(ddyq, ddyq, edge ? q.ax16_edge[j] : q.ax16[j]);
        const f32x2 t = f2_mul(f2_sub(q.si[j], pi), tenth);
        const f32x2 c_no = f2_fma(t, t, dist16);
        dd20[j] = f2_fma(q.s20[j], mm, p20m);
        c[j] = f2_fma(dd20[j], dd20[j], c_no);
    }
    // Costs are >= 0, so their bit patterns order like the values; the candidate's position in the reference's scan goes
    // into the two lowest bits (a change of < 4 ulp, inside the error bound): the smallest tagged cost names the winner.
    // A candidate out of play sits at 1e30 or above; a NaN cost has a bit pattern above every number and never wins either,
    // as it never wins the reference's '<'.  The pick is the reference's for sure if ONE bound separates the winner from the
    // runner-up and from the sentinel the scan starts from: the per-candidate bound (pixel_cost_fast) is monotone in the
    // cost, in |dd20| and in |s20|, so taken at the runner-up's cost (it is the larger of the two; cut at 2e6: above that
    // the second test decides alone) and at the largest |dd20| and |s20| of the four it holds for both -- and c - err(c)
    // grows with c, so the third and fourth lie beyond the runner-up's lower end.
    const float ck[4] = {c[0].x, c[0].y, c[1].x, c[1].y};
    const float ad_max = fmaxf(fmaxf(fabsf(dd20[0].x), fabsf(dd20[0].y)), fmaxf(fabsf(dd20[1].x), fabsf(dd20[1].y)));
    uint32_t key[4];
    for (int k = 0; k < 4; k++) key[k] = (__builtin_bit_cast(uint32_t, ck[k]) & ~3u) | (uint32_t)k;
    const uint32_t lo01 = key[0] < key[1] ? key[0] : key[1], hi01 = key[0] < key[1] ? key[1] : key[0];
    const uint32_t lo23 = key[2] < key[3] ? key[2] : key[3], hi23 = key[2] < key[3] ? key[3] : key[2];
    const uint32_t first = lo01 < lo23 ? lo01 : lo23;
    const uint32_t mid_a = lo01 < lo23 ? lo23 : lo01, mid_b = hi01 < hi23 ? hi01 : hi23;
    const uint32_t second = mid_a < mid_b ? mid_a : mid_b;
    const float c1 = __builtin_bit_cast(float, first), c2 = fminf(__builtin_bit_cast(float, second), 2e6f); // (fminf drops a NaN)
    const float err = fast_cost_err(c2, fast_cost_slack(ad_max, q.df));
    if (trace) { trace->err = err; trace->all_depth = all_depth; }
    FastPick r;
    r.seed = q.base + (int)(first & 1u) * gw + (int)((first >> 1) & 1u);
    r.sure = c1 + err < c2 - err && c1 + err < 1e6f;
    return r;
}
template <typename LoadSeedF>
DSM_HD FastPick pick_seed_fast(int x, int y, float pix_i, float pix_d, int gw, int gh, LoadSeedF load, FastPickTrace *trace = nullptr) {
    return pick_seed_fast(pick_col(x, y, gw, gh, load), y, pix_i, pix_d, gw, trace);
}

// ------------------------------------------------- robust mean depth of a seed, FF.cpp:530-556
// list[0..n) = member depths > 0.1 in window row-major order, sum = their sequential fp32 sum.
// Newton step of the robust mean, FF.cpp:553: delta = (float)((double)(-a) / ((double)b + 10.0)), where b = 2 x (core
// elements) is a small integer.  -a and b + 10 are both floats, and rounding the quotient of two 24-bit values first
// to 53 >= 2*24+2 bits and then to 24 is rounding it once (Figueroa): the correctly rounded fp32 divide gives the
// same bits as the double divide and the cast, at a third of the instructions.
DSM_HD float huber_newton_step(float a, float b) { return (-a) / (b + 10.0f); }

// A seed with a +inf member depth (the reference's feed has them: depth = bf / disparity with disparity 0,
// kitti_publisher/scripts/publisher.py:40) starts from md = +inf, and so does one whose sum overflowed.  Every residual
// inf - d is then +inf (d finite) or NaN (d = +inf): no element is in the Huber core, a = a finite multiple of hr, b = 0,
// delta = -a / 10 is finite and md + delta = +inf again -- whether the loop leaves after one pass or five, the result is
// the +inf it started from.  (The list holds depths > 0.1 only: the sum is never NaN, and never -inf.)  The passes are
// skipped; tests/hostemu.cpp runs this against the reference's loop.
DSM_HD bool mean_depth_is_settled(float md) { return md == __builtin_inff(); }

DSM_HD float huber_mean_depth(const float *list, int n, float sum, double huber) {
    const float hr_above = flt_above(huber);
    float md = sum / (float)n;
    if (mean_depth_is_settled(md)) return md;
    for (int it = 0; it < 5; it++) {
        float a = 0, b = 0;
        for (int k = 0; k < n; k++) {
            float r = md - list[k];
            if (fabsf(r) < hr_above) { // (double)r < huber && (double)r > -huber
                a += 2 * r;
                b += 2;
            } else {
                a = (float)((double)a + (r > 0 ? huber : -1 * huber));
            }
        }
        float delta = huber_newton_step(a, b);
        md = md + delta;
        if (fabsf(delta) < flt_above(0.01)) break; // (double)delta < 0.01 && (double)delta > -0.01
    }
    return md;
}

// ------------------------------------------------------------- back-projection, FF.cpp:91-97
DSM_HD void back_project(const Intrinsics &k, float u, float v, float d, float &x, float &y, float &z) {
    x = (u - k.cx) / k.fx * d;
    y = (v - k.cy) / k.fy * d;
    z = d;
}

// (u - cx) / fx of back_project for an integer pixel column (row: cy, fy): a property of the column, tabulated once per
// handle (`ray_x[x]`, `ray_y[y]`) -- the correctly rounded fp32 divide is ten instructions, and the normal of one pixel needs
// eight of them.  back_project(k, x, y, d) == (ray_x[x] * d, ray_y[y] * d, d), operation for operation.
DSM_HD float ray_coeff(int u, float c, float f) { return ((float)u - c) / f; }

// per-pixel normal from forward differences, FF.cpp:664-712, from the ray coefficients of columns x, x+1 and rows y, y+1.
// Returns false (normal stays 0) when rejected.
DSM_HD bool pixel_normal_rays(float rx0, float rx1, float ry0, float ry1, float d, float d_right, float d_down, float &nx, float &ny,
                              float &nz) {
    constexpr float kMin = flt_above(0.1); // (double)d < 0.1
    if (d < kMin || d_right < kMin || d_down < kMin) return false;
    const float px = rx0 * d, py = ry0 * d, pz = d;
    float rx = rx1 * d_right, ry = ry0 * d_right, rz = d_right;
    float dx = rx0 * d_down, dy = ry1 * d_down, dz = d_down;
    rx = rx - px; ry = ry - py; rz = rz - pz;
    dx = dx - px; dy = dy - py; dz = dz - pz;
    float ax = ry * dz - rz * dy, ay = rz * dx - rx * dz, az = rx * dy - ry * dx;
    float len = sqrtf(ax * ax + ay * ay + az * az);
    ax /= len; ay /= len; az /= len;
    float va = (ax * px + ay * py + az * pz) / sqrtf(px * px + py * py + pz * pz);
    if (fabsf(va) < flt_above(kAngleCos)) return false; // (double)va > -kAngleCos && (double)va < kAngleCos
    nx = ax; ny = ay; nz = az;
    return true;
}

// per-pixel normal from forward differences, FF.cpp:664-712.  Caller guarantees
// 1 <= x <= w-2 and 1 <= y <= h-2; returns false (normal stays 0) when rejected.
DSM_HD bool pixel_normal(const Intrinsics &k, int x, int y, float d, float d_right, float d_down, float &nx,
                         float &ny, float &nz) {
    constexpr float kMin = flt_above(0.1); // (double)d < 0.1
    if (d < kMin || d_right < kMin || d_down < kMin) return false;
    float px, py, pz, rx, ry, rz, dx, dy, dz;
    back_project(k, (float)x, (float)y, d, px, py, pz);
    back_project(k, (float)(x + 1), (float)y, d_right, rx, ry, rz);
    back_project(k, (float)x, (float)(y + 1), d_down, dx, dy, dz);
    rx = rx - px; ry = ry - py; rz = rz - pz;
    dx = dx - px; dy = dy - py; dz = dz - pz;
    float ax = ry * dz - rz * dy, ay = rz * dx - rx * dz, az = rx * dy - ry * dx;
    float len = sqrtf(ax * ax + ay * ay + az * az);
    ax /= len; ay /= len; az /= len;
    float va = (ax * px + ay * py + az * pz) / sqrtf(px * px + py * py + pz * pz);
    if (fabsf(va) < flt_above(kAngleCos)) return false; // (double)va > -kAngleCos && (double)va < kAngleCos
    nx = ax; ny = ay; nz = az;
    return true;
}

// -------------------------------------- general 4x4 inverse, column-major, adjugate / determinant
// (stands in for Eigen's Matrix4::inverse at FF.cpp:59 and FF.cpp:176; same closed form and
// operation order as the oracle's Eigen shim).
template <typename T> DSM_HD void inverse4(const T *a, T *o) {
    T s0 = a[0] * a[5] - a[1] * a[4], s1 = a[0] * a[9] - a[1] * a[8], s2 = a[0] * a[13] - a[1] * a[12];
    T s3 = a[4] * a[9] - a[5] * a[8], s4 = a[4] * a[13] - a[5] * a[12], s5 = a[8] * a[13] - a[9] * a[12];
    T c5 = a[10] * a[15] - a[11] * a[14], c4 = a[6] * a[15] - a[7] * a[14], c3 = a[6] * a[11] - a[7] * a[10];
    T c2 = a[2] * a[15] - a[3] * a[14], c1 = a[2] * a[11] - a[3] * a[10], c0 = a[2] * a[7] - a[3] * a[6];
    T det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    T id = (T)1 / det;
    o[0] = (a[5] * c5 - a[9] * c4 + a[13] * c3) * id;
    o[4] = (-a[4] * c5 + a[8] * c4 - a[12] * c3) * id;
    o[8] = (a[7] * s5 - a[11] * s4 + a[15] * s3) * id;
    o[12] = (-a[6] * s5 + a[10] * s4 - a[14] * s3) * id;
    o[1] = (-a[1] * c5 + a[9] * c2 - a[13] * c1) * id;
    o[5] = (a[0] * c5 - a[8] * c2 + a[12] * c1) * id;
    o[9] = (-a[3] * s5 + a[11] * s2 - a[15] * s1) * id;
    o[13] = (a[2] * s5 - a[10] * s2 + a[14] * s1) * id;
    o[2] = (a[1] * c4 - a[5] * c2 + a[13] * c0) * id;
    o[6] = (-a[0] * c4 + a[4] * c2 - a[12] * c0) * id;
    o[10] = (a[3] * s4 - a[7] * s2 + a[15] * s0) * id;
    o[14] = (-a[2] * s4 + a[6] * s2 - a[14] * s0) * id;
    o[3] = (-a[1] * c3 + a[5] * c1 - a[9] * c0) * id;
    o[7] = (a[0] * c3 - a[4] * c1 + a[8] * c0) * id;
    o[11] = (-a[3] * s3 + a[7] * s1 - a[11] * s0) * id;
    o[15] = (a[2] * s3 - a[6] * s1 + a[10] * s0) * id;
}

// The same inverse as data: 12 2x2 determinants D[t] = a[da[t][0]]*a[da[t][1]] - a[da[t][2]]*a[da[t][3]]
// (D[0..5] = s0..s5, D[6..11] = c0..c5), the determinant as a left-to-right chain over kDetTerms, and
// 16 adjugate entries  o[k] = ((g*(a[x1]*D[d1]) - g*(a[x2]*D[d2])) + g*(a[x3]*D[d3])) * (1/det), g = +-1
// (negating a product is exact, so this is inverse4 bit for bit).  The HIP kernel evaluates the table
// with one lane per determinant / entry; inverse4_tabled is the serial form used to test the table.
struct Inv4Tables {
    signed char det2[12][4];
    signed char det_terms[6][3];  // s index, c index (into D), sign
    signed char out[16][7];       // x1 d1 x2 d2 x3 d3 leading sign
};
constexpr Inv4Tables kInv4 = {
    {{0, 5, 1, 4}, {0, 9, 1, 8}, {0, 13, 1, 12}, {4, 9, 5, 8}, {4, 13, 5, 12}, {8, 13, 9, 12},
     {2, 7, 3, 6}, {2, 11, 3, 10}, {2, 15, 3, 14}, {6, 11, 7, 10}, {6, 15, 7, 14}, {10, 15, 11, 14}},
    {{0, 11, 1}, {1, 10, -1}, {2, 9, 1}, {3, 8, 1}, {4, 7, -1}, {5, 6, 1}},
    {{5, 11, 9, 10, 13, 9, 1},   // o[0]  =  a5*c5 - a9*c4 + a13*c3
     {1, 11, 9, 8, 13, 7, -1},   // o[1]  = -a1*c5 + a9*c2 - a13*c1
     {1, 10, 5, 8, 13, 6, 1},    // o[2]  =  a1*c4 - a5*c2 + a13*c0
     {1, 9, 5, 7, 9, 6, -1},     // o[3]  = -a1*c3 + a5*c1 - a9*c0
     {4, 11, 8, 10, 12, 9, -1},  // o[4]  = -a4*c5 + a8*c4 - a12*c3
     {0, 11, 8, 8, 12, 7, 1},    // o[5]  =  a0*c5 - a8*c2 + a12*c1
     {0, 10, 4, 8, 12, 6, -1},   // o[6]  = -a0*c4 + a4*c2 - a12*c0
     {0, 9, 4, 7, 8, 6, 1},      // o[7]  =  a0*c3 - a4*c1 + a8*c0
     {7, 5, 11, 4, 15, 3, 1},    // o[8]  =  a7*s5 - a11*s4 + a15*s3
     {3, 5, 11, 2, 15, 1, -1},   // o[9]  = -a3*s5 + a11*s2 - a15*s1
     {3, 4, 7, 2, 15, 0, 1},     // o[10] =  a3*s4 - a7*s2 + a15*s0
     {3, 3, 7, 1, 11, 0, -1},    // o[11] = -a3*s3 + a7*s1 - a11*s0
     {6, 5, 10, 4, 14, 3, -1},   // o[12] = -a6*s5 + a10*s4 - a14*s3
     {2, 5, 10, 2, 14, 1, 1},    // o[13] =  a2*s5 - a10*s2 + a14*s1
     {2, 4, 6, 2, 14, 0, -1},    // o[14] = -a2*s4 + a6*s2 - a14*s0
     {2, 3, 6, 1, 10, 0, 1}},    // o[15] =  a2*s3 - a6*s1 + a10*s0
};
template <typename T> DSM_HD T inv4_det2(const T *a, int t) {
    return a[kInv4.det2[t][0]] * a[kInv4.det2[t][1]] - a[kInv4.det2[t][2]] * a[kInv4.det2[t][3]];
}
template <typename T> DSM_HD T inv4_det(const T *D) {
    T det = D[kInv4.det_terms[0][0]] * D[kInv4.det_terms[0][1]];
    for (int i = 1; i < 6; i++) {
        const T prod = D[kInv4.det_terms[i][0]] * D[kInv4.det_terms[i][1]];
        det = kInv4.det_terms[i][2] > 0 ? det + prod : det - prod;
    }
    return det;
}
template <typename T> DSM_HD T inv4_entry(const T *a, const T *D, T inv_det, int k) {
    const signed char *e = kInv4.out[k];
    const T g = (T)e[6];
    const T t1 = g * (a[e[0]] * D[e[1]]), t2 = g * (a[e[2]] * D[e[3]]), t3 = g * (a[e[4]] * D[e[5]]);
    return ((t1 - t2) + t3) * inv_det;
}
template <typename T> DSM_HD void inverse4_tabled(const T *a, T *o) {
    T D[12];
    for (int t = 0; t < 12; t++) D[t] = inv4_det2(a, t);
    const T id = (T)1 / inv4_det(D);
    for (int k = 0; k < 16; k++) o[k] = inv4_entry(a, D, id, k);
}

// One Gauss-Newton accumulator of get_huber_norm (FF.cpp:129-170).  The 16 Hessian entries and
// the 4 Jacobian entries are independent sequential double sums; with the homogeneous point
// p = (p0,p1,p2,1) they are  H(a,b) += (double)(2*p_a*p_b)  and  J(a) += (double)(2*r*p_a)  in
// the Huber core and  J(a) += +-hr*(double)p_a  in the tails (multiplying by 1.0f is exact, so
// the reference's special-cased last row/column give the same bits).
struct GnTerm {
    int a, b;  // b < 0: Jacobian entry a
};
DSM_HD double gn_term_add(double acc, const GnTerm &t, const float p[4], float r, double hr) {
    const float pa = p[t.a];
    if ((double)r < hr && (double)r > -1 * hr) {
        if (t.b < 0) return acc + (double)(2 * r * pa);
        return acc + (double)(2 * pa * p[t.b]);
    } else if ((double)r >= hr) {
        if (t.b < 0) return acc + hr * (double)pa;
    } else if ((double)r <= -1 * hr) {
        if (t.b < 0) return acc + -1 * hr * (double)pa;
    }
    return acc;
}

// The same sums in the form the HIP kernel streams them: residual class first, then one branch-free
// term per (element, accumulator).  X,Y = (p_a, p_b) for H(a,b) and (r, p_a) for J(a).
// (hr_above = flt_above(hr): the reference's double compares of the float residual, in fp32)
DSM_HD int huber_class32(float r, float hr_above) {
    if (fabsf(r) < hr_above) return 0; // core:       (double)r < hr && (double)r > -hr
    if (r >= hr_above) return 1;       // upper tail: (double)r >= hr
    if (r <= -hr_above) return 2;      // lower tail: (double)r <= -hr
    return 3;                          // NaN: contributes nothing
}
DSM_HD double gn_term(bool is_jacobian, float X, float Y, int cls, double hr) {
    const double core = (double)(2 * X * Y);
    const double tail = (cls == 1 ? hr : -1 * hr) * (double)Y;
    return cls == 0 ? core : ((is_jacobian && cls != 3) ? tail : 0.0);
}

// solve and apply one Gauss-Newton step, FF.cpp:172-180.  H column-major 4x4 (without damping).
DSM_HD void gn_step(double *H, const double *J, float &nx, float &ny, float &nz, float &nb) {
    H[0] += 5; H[5] += 5; H[10] += 5; H[15] += 5;
    double Hi[16];
    inverse4<double>(H, Hi);
    double u[4];
    for (int i = 0; i < 4; i++) u[i] = ((Hi[i] * J[0] + Hi[4 + i] * J[1]) + Hi[8 + i] * J[2]) + Hi[12 + i] * J[3];
    nx = (float)((double)nx - u[0]);
    ny = (float)((double)ny - u[1]);
    nz = (float)((double)nz - u[2]);
    nb = (float)((double)nb - u[3]);
}

// Gauss-Newton steps whose Jacobian sums are NOT taken in the reference's order -- where that provably changes nothing
// (round 6; k_seed_fit, steps 2..5).  Every term of J(a) = 2 sum_i (double)(r_i * p_i,a) is formed exactly as the reference
// forms it: an fp32 product widened to double.  The terms are fp32 values, every one a multiple of the granule
// g = 2^(floor(log2 min|t|) - 23) of the smallest non-zero one, and while sum|t| < 2^53 g every partial sum of every order is a
// multiple of g below 2^53 g -- a double -- so NO addition rounds and all orders give the same bits.  With 2^floor(log2 x) > x / 2
// that is  sum|t| < 2^29 min|t|,  tested as  sum|t| * 2^-28 < min|t|  (a factor 2 for the fp32 roundings of sum|t|, which
// the kernel accumulates in fp32: n 2^-24 relative, n <= 232); a sum that overflowed to +inf or holds a NaN fails it, an
// all-zero list passes (min = +inf: the sum is 0 in any order).  tests/hostemu.cpp: 99.8 % of the steps that qualify (Huber
// classes unchanged, all in the core) pass for all four components, and their free-order sums ARE the ordered sums, bit for bit.
DSM_HD bool gn_sum_is_exact(float t_abs_sum, float t_abs_min_nonzero) { return t_abs_sum * 3.7252902984619140625e-9f < t_abs_min_nonzero; }
// the smallest non-zero magnitude by an unsigned minimum: (bits(t) << 1) - 1 drops the sign, keeps the order of the non-zero
// magnitudes and sends +-0 to the top (0xffffffff) -- one shift-and-add per term; gn_min_key_value undoes it (the top -> +inf)
DSM_HD uint32_t gn_min_key(float t) { return (__builtin_bit_cast(uint32_t, t) << 1) - 1u; }
DSM_HD float gn_min_key_value(uint32_t key) { return key == 0xffffffffu ? __builtin_inff() : __builtin_bit_cast(float, (key + 1u) >> 1); }

// tail of get_huber_norm, FF.cpp:182-187
DSM_HD void plane_finish(float &nx, float &ny, float &nz, float &nb, float mx, float my, float mz) {
    nb = nb - (nx * mx + ny * my + nz * mz);
    float len = sqrtf(nx * nx + ny * ny + nz * nz);
    nx /= len; ny /= len; nz /= len; nb /= len;
}

// Seed geometry after the plane fit, FF.cpp:884-912.
struct SeedGeom {
    float nx, ny, nz, px, py, pz, view_cos, mean_depth;
};
DSM_HD SeedGeom seed_geometry(const Intrinsics &k, float seed_x, float seed_y, float md, float nx, float ny, float nz,
                              float nb) {
    float bx, by, bz;
    back_project(k, seed_x, seed_y, md, bx, by, bz);
    double ax = bx, ay = by, az = bz;
    float kk = (float)(-1 * (ax * (double)nx + ay * (double)ny + az * (double)nz) - (double)nb); // FF.cpp:890
    ax += (double)(kk * nx); ay += (double)(kk * ny); az += (double)(kk * nz);
    md = (float)az;
    float vc = (float)(-1.0 * ((double)nx * ax + (double)ny * ay + (double)nz * az) / sqrt(ax * ax + ay * ay + az * az));
    if (vc < 0) { vc = -vc; nx = -nx; ny = -ny; nz = -nz; }
    SeedGeom g;
    g.nx = nx; g.ny = ny; g.nz = nz;
    g.px = (float)ax; g.py = (float)ay; g.pz = (float)az;
    g.view_cos = vc; g.mean_depth = md;
    return g;
}

// ------------------------------------------------------------------ rigid transforms, FF.cpp:220,228
DSM_HD void xform_point(const float *m, const float *p, float *o) {
    for (int i = 0; i < 3; i++) o[i] = ((m[i] * p[0] + m[4 + i] * p[1]) + m[8 + i] * p[2]) + m[12 + i] * 1.0f;
}
DSM_HD void xform_dir(const float *m, const float *v, float *o) {
    for (int i = 0; i < 3; i++) o[i] = (m[i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2];
}
DSM_HD float depth_weight(float d) { // FF.cpp:99-102
    double w = 1.0 / (double)d / (double)d;
    return (float)(1.0 < w ? 1.0 : w);
}
DSM_HD float camera_focal(const Intrinsics &k) { // FF.cpp:250,350
    return (float)((double)(fabsf(k.fx) + fabsf(k.fy)) / 2.0);
}
// int(x + 0.5) as x86-64 cvttsd2si does it (NaN / out of range -> INT_MIN), FF.cpp:235-236
DSM_HD int round_to_pixel(float u) {
    double ud = (double)u + 0.5;
    return (ud >= -2147483648.0 && ud < 2147483648.0) ? (int)ud : (-2147483647 - 1);
}

// The plain-data views the fuse functions work on (layouts of elements.h:5-31).
struct Surfel {
    float px, py, pz, nx, ny, nz, size, color, weight;
    int32_t update_times, last_update;
};
struct SeedView {  // the fields of Superpixel_seed that fusion reads
    float size, nx, ny, nz, px, py, pz, view_cos, mean_depth, mean_intensity;
};

struct FuseConst {
    Intrinsics k;
    float far_d, near_d;
    double baseline, disp_err, min_tol;
    int w, h;
    // derived by fuse_const_prepare()
    float cam_f;                 // camera_focal(k)
    bool tol32;                  // the depth tolerance may be evaluated in fp32 (below)
    float tol_den, tol_scale;    // baseline * cam_f and disp_err as floats, when tol32
    float min_tol_above, min_tol_f;
};

enum FuseOutcome { kFuseSkip = 0, kFuseDeleted = 1, kFuseFused = 2, kFuseNeedPixel = 3 };

// Depth tolerance of the association, FF.cpp:250-253:  tol = z*z / (BASELINE * camera_f) * DISPARITY_ERROR  evaluated
// in double (the macros are double literals) and assigned to a float, then clamped from below by MIN_TOLERATE_DIFF.
// When BASELINE * (double)camera_f is itself a float value and DISPARITY_ERROR a power of two -- the reference's
// two constant sets: 0.5 and 4.0, 0.08 and 1.0 with the 525-pixel focal length -- this is an fp32 expression: scaling
// by a power of two commutes with rounding, and rounding the quotient of two floats first to 53 >= 2*24+2 bits and
// then to 24 bits is rounding it once (Figueroa), so  tol = (z*z / den_f) * scale_f  has the same bits (operands kept
// away from the subnormal range by the near plane).  Any other constant set takes the double expression.
DSM_HD void fuse_const_prepare(FuseConst &c) {
    c.cam_f = camera_focal(c.k);
    const double den = c.baseline * (double)c.cam_f;
    const float den_f = (float)den;
    const uint64_t scale_bits = __builtin_bit_cast(uint64_t, c.disp_err);
    const bool scale_pow2 = c.disp_err >= 1.0 / 1048576.0 && c.disp_err <= 1048576.0 && (scale_bits & 0x000fffffffffffffull) == 0;
    c.tol32 = (double)den_f == den && den >= 1e-6 && den <= 1e6 && scale_pow2 && c.near_d >= 1e-3f;
    c.tol_den = den_f;
    c.tol_scale = (float)c.disp_err;
    c.min_tol_above = flt_above(c.min_tol);
    c.min_tol_f = (float)c.min_tol;
}
DSM_HD float fuse_depth_tolerance(const FuseConst &c, float z) {
    float tol;
    if (c.tol32) tol = (z * z) / c.tol_den * c.tol_scale;
    else tol = (float)((double)(z * z) / (c.baseline * (double)c.cam_f) * c.disp_err);
    // tol = (float)((double)tol < min_tol ? min_tol : (double)tol)
    return tol < c.min_tol_above ? c.min_tol_f : tol;
}

// Stage 1 of fuse_surfels_kernel (FF.cpp:205-238): pruning and projection.  Returns kFuseNeedPixel
// with (ui,vi) and camera-frame position/normal when the surfel lands inside the image.
DSM_HD FuseOutcome fuse_project(const FuseConst &c, int ref_idx, const float *inv, Surfel &e, int &ui, int &vi,
                                float pc[3], float nc[3]) {
    if (ref_idx - e.last_update > 5 && e.update_times < 5) {
        e.update_times = 0;
        return kFuseDeleted;
    }
    if (e.update_times == 0) return kFuseSkip;
    float pw[3] = {e.px, e.py, e.pz}, nw[3] = {e.nx, e.ny, e.nz};
    xform_point(inv, pw, pc);
    if (pc[2] < c.near_d || pc[2] > c.far_d) return kFuseSkip;
    xform_dir(inv, nw, nc);
    float u = pc[0] * c.k.fx / pc[2] + c.k.cx, v = pc[1] * c.k.fy / pc[2] + c.k.cy; // FF.cpp:85-89
    ui = round_to_pixel(u);
    vi = round_to_pixel(v);
    if (ui < 1 || ui > c.w - 2 || vi < 1 || vi > c.h - 2) return kFuseSkip;
    return kFuseNeedPixel;
}

// Stage 2 (FF.cpp:239-311) given the depth at the projected pixel and the seed owning it.
// (c prepared by fuse_const_prepare; w1 = depth_weight(sd.mean_depth), which k_seed_fit leaves per seed: two double
// divides that every surfel fusing into the seed would repeat)
DSM_HD FuseOutcome fuse_update(const FuseConst &c, int ref_idx, const float *pose, Surfel &e, const float pc[3],
                               const float nc[3], float pix_depth, const SeedView &sd, float w1) {
    if ((double)pc[2] < (double)pix_depth - 1.0) {
        e.update_times = 0;
        return kFuseDeleted;
    }
    if (sd.nx == 0 && sd.ny == 0 && sd.nz == 0) return kFuseSkip;
    if (sd.view_cos < flt_above(kAngleCos)) return kFuseSkip; // (double)view_cos < MAX_ANGLE_COS
    const float cam_f = c.cam_f;
    const float tol = fuse_depth_tolerance(c, pc[2]);
    if (pc[2] < sd.mean_depth - tol) return kFuseSkip;
    if (pc[2] > sd.mean_depth + tol) return kFuseSkip;
    float ncos = nc[0] * sd.nx + nc[1] * sd.ny + nc[2] * sd.nz;
    if (ncos < flt_above(kAngleCos)) { // (double)ncos < MAX_ANGLE_COS
        e.update_times = 0;
        return kFuseDeleted;
    }
    float w0 = e.weight, ws = w0 + w1;
    float sc[3] = {sd.px, sd.py, sd.pz}, sw[3];
    xform_point(pose, sc, sw);
    float fpx = (e.px * w0 + w1 * sw[0]) / ws, fpy = (e.py * w0 + w1 * sw[1]) / ws, fpz = (e.pz * w0 + w1 * sw[2]) / ws;
    float fn[3] = {nc[0] * w0 + w1 * sd.nx, nc[1] * w0 + w1 * sd.ny, nc[2] * w0 + w1 * sd.nz};
    // FF.cpp:287-291: double len = sqrt(float); fn /= len in double, stored to float.  Numerator and denominator are
    // float values, so each quotient is the correctly rounded fp32 divide (Figueroa, as above).
    const float len = sqrtf(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
    fn[0] = fn[0] / len;
    fn[1] = fn[1] / len;
    fn[2] = fn[2] / len;
    float fw[3];
    xform_dir(pose, fn, fw);
    e.px = fpx; e.py = fpy; e.pz = fpz;
    e.nx = fw[0]; e.ny = fw[1]; e.nz = fw[2];
    e.weight = ws;
    e.color = sd.mean_intensity;
    float nsz = sd.size * fabsf(sd.mean_depth / (cam_f * sd.view_cos));
    if (nsz < e.size) e.size = nsz;
    e.last_update = ref_idx;
    e.update_times += 1;
    return kFuseFused;
}

// initialize_surfels, FF.cpp:315-361: does this seed create a surfel, and which one.
DSM_HD bool seed_spawns(const SeedView &sd, bool fused) {
    if (sd.mean_depth == 0) return false;
    if (fused) return false;
    if (sd.view_cos < flt_above(kAngleCos)) return false; // (double)view_cos < MAX_ANGLE_COS
    if (sd.nx == 0 && sd.ny == 0 && sd.nz == 0) return false;
    return true;
}
DSM_HD Surfel spawn_surfel(const Intrinsics &k, int ref_idx, const float *pose, const SeedView &sd) {
    float pc[3] = {sd.px, sd.py, sd.pz}, nc[3] = {sd.nx, sd.ny, sd.nz}, pw[3], nw[3];
    xform_point(pose, pc, pw);
    xform_dir(pose, nc, nw);
    float cam_f = camera_focal(k);
    Surfel e;
    e.px = pw[0]; e.py = pw[1]; e.pz = pw[2];
    e.nx = nw[0]; e.ny = nw[1]; e.nz = nw[2];
    e.size = sd.size * fabsf(sd.mean_depth / (cam_f * sd.view_cos));
    e.color = sd.mean_intensity;
    e.weight = depth_weight(sd.mean_depth);
    e.update_times = 1;
    e.last_update = ref_idx;
    return e;
}

// worker k's [begin,end) over n items, FF.cpp:198-202 / 392-396 / 471-475
DSM_HD int chunk_of(int n, int i) {
    int step = n / kWorkers;
    if (step == 0) return kWorkers - 1;
    int k = i / step;
    return k > kWorkers - 1 ? kWorkers - 1 : k;
}

} // namespace dsm
