// TEST INFRASTRUCTURE (CPU, no GPU needed): a serial walk through the *GPU formulation* of the hot
// path -- the same dsm_math.h the HIP kernels use, the same tmin / worklist fixed point, staged seed
// commit, 20-accumulator Gauss-Newton and parallel-exact compaction as dsm_kernels.hip, with every
// "thread" / "wave" executed one after the other.  tests/test_hostemu.py compares it with the oracle.
// It validates the reformulations and the shared arithmetic before any GPU time is spent; it is
// never loaded by the product package.
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off -shared -fPIC tests/hostemu.cpp -o tests/_build/libhostemu.so
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>

#include "../densesurfelmapping_amd/csrc/dsm_math.h"
#include "../include/dsm.h"

using namespace dsm;

namespace {
constexpr int kIntMax = 0x7fffffff;
struct Core { float x, y, i, d; };

struct Emu {
    int w, h, gw, gh, S;
    Intrinsics K;
    float far_d, near_d;
    double huber, baseline, disp_err, min_tol;
    std::vector<int32_t> label, cand, tmin, stable_stage, worklist;
    std::vector<Core> core, stage;
    std::vector<double> inv_depth;
    std::vector<dsm_seed> seeds;
    int first_empty[kSweeps][kWorkers];
    const uint8_t *img; size_t img_step;
    const float *dep; size_t dep_step;
    int order_salt = 0; // permutes "thread" execution order to exercise order independence
    std::set<long long> unsure_waves; long long sweep_id = 0;
    long long gn_first_noncore = 0; // seeds with a residual outside the Huber core at step 1
    long long gn_seeds = 0, gn_seeds_mask_changed = 0, gn_steps = 0, gn_steps_mask_changed = 0; // Gauss-Newton steps 2..5 whose Huber classes differ from the step before
    // exact-sum study: Gauss-Newton sums whose fp32-product terms span <= 21 binades are exact in
    // double in ANY order; [0] fitted seeds whose step-1 sums (9 H + 4 J) all qualify, [1] of steps 2..5: J sums all
    // qualify, [2] steps 2..5 counted, [3] qualifying sums whose tree-order value differs from the ordered one (must be 0),
    // [4] all-core steps
    long long exact_stats[5] = {0, 0, 0, 0, 0};
    // k_seed_fit's free-order steps (dsm_math.h, gn_sum_is_exact): [0] steps 2..5 that qualify (all-core, classes unchanged: the
    // cached inverse is reused), [1] of them with all four Jacobian sums provably exact in any order, [2] of those, sums that
    // differ from the ordered sums all the same (must be 0)
    long long cert_stats[3] = {0, 0, 0};
    long long exact_operand_pass = 0; // steps 2..5 that pass the cheaper OPERAND-level test: range(r) + range(p_a) + 1 <= 21 for every a
    long long fast_total = 0, fast_unsure = 0, fast_mismatches = 0, fast_checked = 0, fast_bound_violations = 0; // pick_seed_fast
    long long unsure_by_sweep[kSweeps] = {0, 0, 0};

    float I(int x, int y) const { return (float)img[(size_t)y * img_step + x]; }
    float D(int x, int y) const { return *(const float *)((const char *)dep + (size_t)y * dep_step + (size_t)x * 4); }
    int key(int x, int y) const { return y * w + x; }
};

void init_seeds(Emu &e) {
    for (int s = 0; s < e.S; s++) {
        int gx = s % e.gw, gy = s / e.gw;
        int ix = gx * kCell + kCell / 2, iy = gy * kCell + kCell / 2;
        float md = e.D(ix, iy);
        if (md < flt_above(0.01)) { // (double)md < 0.01
            int x0 = ix - kCell, y0 = iy - kCell, x1 = x0 + 2 * kCell, y1 = y0 + 2 * kCell;
            if (x0 < 0) x0 = 0;
            if (y0 < 0) y0 = 0;
            if (x1 > e.w - 1) x1 = e.w - 1;
            if (y1 > e.h - 1) y1 = e.h - 1;
            bool found = false;
            for (int y = y0; y < y1 && !found; y++)
                for (int x = x0; x < x1; x++)
                    if (e.D(x, y) > flt_below(0.01)) { md = e.D(x, y); found = true; break; }
        }
        e.core[s] = {(float)ix, (float)iy, e.I(ix, iy), md};
        e.inv_depth[s] = 1.0 / (double)md;
        e.tmin[s] = -1;
    }
    for (int k = 0; k < kSweeps; k++)
        for (int j = 0; j < kWorkers; j++) e.first_empty[k][j] = kIntMax;
}

void assign(Emu &e, bool first) {
    e.sweep_id++;
    e.worklist.clear();
    const int n = e.w * e.h;
    for (int q = 0; q < n; q++) {
        // visit pixels in a scrambled order: the result must not depend on it
        const int p = e.order_salt ? (int)(((int64_t)q * 7919 + e.order_salt) % n) : q;
        const int x = p % e.w, y = p / e.w;
        if (!has_candidate_cell(x, y, e.gw, e.gh)) { // ragged border: label -1 for the whole frame (k_assign)
            if (first) e.label[p] = -1;
            continue;
        }
        const int exact = pick_seed(x, y, e.I(x, y), e.D(x, y), e.gw, e.gh,
                                    [&](int gx, int gy, float &sx, float &sy, float &si, bool &hd, double &inv) {
                                        const int s = gy * e.gw + gx;
                                        sx = e.core[s].x; sy = e.core[s].y; si = e.core[s].i;
                                        hd = e.core[s].d > 0; inv = e.inv_depth[s];
                                    });
        // the kernel's filtered pick (dsm_math.h, pick_seed_fast): wherever it answers, it must be the reference's pick
        FastPickTrace trace = {0.0f, false};
        const float p20 = pixel_has_inv_depth(e.D(x, y)) ? pixel_p20(e.D(x, y)) : 0.0f; // (the host's model of v_rcp_f32: a hash-chosen ulp off)
        const FastPick fast = pick_seed_fast(x, y, e.I(x, y), e.D(x, y), e.gw, e.gh,
                                        [&](int, int gx, int gy, float &sx4, float &sy4, float &si, float &sd, float &s20) {
                                            const bool inside = gx >= 0 && gx < e.gw && gy >= 0 && gy < e.gh;
                                            gx = gx < 0 ? 0 : (gx > e.gw - 1 ? e.gw - 1 : gx); // (outside the grid: the nearest cell inside)
                                            gy = gy < 0 ? 0 : (gy > e.gh - 1 ? e.gh - 1 : gy);
                                            const int s = gy * e.gw + gx;
                                            const float sx = e.core[s].x, sy = e.core[s].y;
                                            sx4 = sx * 0.25f; sy4 = sy * 0.25f; si = e.core[s].i;
                                            sd = e.core[s].d; s20 = seed_s20(sd, e.inv_depth[s]);
                                            if (!inside) return;
                                            const bool hd = sd > 0;
                                            // ... and its error bound must hold against the reference's typed costs
                                            const float invd = pixel_inv_depth(e.D(x, y));
                                            float cn, cd;
                                            const bool with = pixel_cost(sx, sy, si, hd, e.inv_depth[s], e.I(x, y), invd, x, y, cn, cd);
                                            for (int depth_term = 0; depth_term < 2; depth_term++) {
                                                if (depth_term && !with) continue;
                                                const FastCost f = pixel_cost_fast(sx, sy, si, s20, depth_term != 0, e.I(x, y), p20, x, y);
                                                const double ref = depth_term ? (double)cd : (double)cn;
                                                e.fast_checked++;
                                                if (!(fabs((double)f.c - ref) <= (double)f.err)) e.fast_bound_violations++;
                                            }
                                        }, &trace);
        if (fast.sure) {
            // the ONE bound the pick was decided with must cover the distance between the filter's cost and the reference's typed
            // cost for every candidate in play whose cost lies at or below the runner-up's (cut at 2e6) -- the winner and the
            // runner-up among them
            const int gx0 = x / kCell - (x % kCell < kCell / 2 ? 1 : 0), gy0 = y / kCell - (y % kCell < kCell / 2 ? 1 : 0);
            const float invd = pixel_inv_depth(e.D(x, y));
            float cf[4], cr[4], cs[4];
            int n_in = 0;
            for (int k = 0; k < 4; k++) {
                const int gx = gx0 + (k >> 1), gy = gy0 + (k & 1);
                const bool in = gx >= 0 && gx < e.gw && gy >= 0 && gy < e.gh && abs(gx * kCell + kCell / 2 - x) < kCell && abs(gy * kCell + kCell / 2 - y) < kCell;
                if (!in) continue;
                const int s = gy * e.gw + gx;
                cf[n_in] = pixel_cost_fast(e.core[s].x, e.core[s].y, e.core[s].i, seed_s20(e.core[s].d, e.inv_depth[s]), trace.all_depth, e.I(x, y), p20, x, y).c;
                float cn, cd;
                pixel_cost(e.core[s].x, e.core[s].y, e.core[s].i, e.core[s].d > 0, e.inv_depth[s], e.I(x, y), invd, x, y, cn, cd);
                cr[n_in] = trace.all_depth ? cd : cn;
                cs[n_in] = cf[n_in];
                n_in++;
            }
            std::sort(cs, cs + n_in);
            const float c2 = n_in > 1 ? std::min(cs[1], 2e6f) : 2e6f;
            for (int k = 0; k < n_in; k++) {
                if (!(cf[k] <= c2)) continue;
                e.fast_checked++;
                if (!(fabs((double)cf[k] - (double)cr[k]) <= (double)trace.err)) e.fast_bound_violations++;
            }
        }
        e.fast_total++;
        if (!fast.sure) { e.fast_unsure++; e.unsure_by_sweep[(e.sweep_id - 1) % kSweeps]++; e.unsure_waves.insert(((long long)e.sweep_id << 40) | (long long)(y * ((e.w + 63) / 64) + x / 64)); }
        else if (fast.seed != exact) e.fast_mismatches++;
        const int pick = fast.sure ? fast.seed : exact;
        if (first) { e.label[p] = pick; continue; }
        e.cand[p] = pick;
        const int l = e.label[p];
        if (e.tmin[l] == -1) {
            if (e.tmin[pick] > p) e.tmin[pick] = p;
        } else if (pick != l && e.tmin[pick] != -1) {
            e.worklist.push_back(p);
        }
    }
}

void resolve(Emu &e) {
    for (;;) {
        bool changed = false;
        for (size_t i = 0; i < e.worklist.size(); i++) {
            const int p = e.worklist[e.order_salt ? e.worklist.size() - 1 - i : i];
            const int l = e.label[p], pk = e.cand[p];
            if (e.tmin[l] < p && e.tmin[pk] > p) { e.tmin[pk] = p; changed = true; }
        }
        if (!changed) break;
    }
}

void apply(Emu &e) {
    for (int p = 0; p < e.w * e.h; p++)
        if (e.label[p] >= 0 && e.tmin[e.label[p]] < p) e.label[p] = e.cand[p];
}

void update_seeds(Emu &e, int sweep) {
    float dl[256];
    for (int s = 0; s < e.S; s++) {
        if (e.tmin[s] == kIntMax) continue;
        const int wx0 = (s % e.gw) * kCell + kCell / 2 - kCell, wy0 = (s / e.gw) * kCell + kCell / 2 - kCell;
        int cnt = 0, sx = 0, sy = 0, si = 0, nd = 0;
        for (int idx = 0; idx < 256; idx++) {
            const int x = wx0 + (idx & 15), y = wy0 + (idx >> 4);
            const bool in = x >= 0 && x < e.w - 1 && y >= 0 && y < e.h - 1;
            if (!in || e.label[e.key(x, y)] != s) continue;
            cnt++; sx += x; sy += y; si += (int)e.img[(size_t)y * e.img_step + x];
            const float d = e.D(x, y);
            if (d > flt_below(0.1)) dl[nd++] = d; // (double)d > 0.1
        }
        if (cnt == 0) {
            int &fe = e.first_empty[sweep][chunk_of(e.S, s)];
            if (s < fe) fe = s;
            continue;
        }
        const float fn = (float)cnt;
        const float mi = (float)si / fn, mx = (float)sx / fn, my = (float)sy / fn;
        const Core old = e.core[s];
        const float moved = fabsf(old.i - mi) + fabsf(old.x - mx) + fabsf(old.y - my);
        float md = 0.0f;
        if (nd > 0) {
            float sum = 0.0f;
            for (int i = 0; i < nd; i++) sum += dl[i];
            md = huber_mean_depth(dl, nd, sum, e.huber);
        }
        e.stage[s] = {mx, my, mi, md};
        e.stable_stage[s] = moved < flt_above(0.2) ? 1 : 0; // (double)moved < 0.2
    }
    for (int s = 0; s < e.S; s++) { // commit
        if (e.tmin[s] == kIntMax) continue;
        int t = -1;
        if (s < e.first_empty[sweep][chunk_of(e.S, s)]) {
            e.core[s] = e.stage[s];
            e.inv_depth[s] = 1.0 / (double)e.stage[s].d;
            if (e.stable_stage[s]) t = kIntMax;
        }
        e.tmin[s] = t;
    }
}

void seed_planes(Emu &e) {
    float ld[256], ln[768], lp[768];
    int lx[256], ly[256];
    for (int s = 0; s < e.S; s++) {
        const Core core = e.core[s];
        const int wx0 = (s % e.gw) * kCell + kCell / 2 - kCell, wy0 = (s / e.gw) * kCell + kCell / 2 - kCell;
        int n = 0;
        float far2 = 0.0f;
        for (int idx = 0; idx < 256; idx++) {
            const int x = wx0 + (idx & 15), y = wy0 + (idx >> 4);
            if (!(x >= 0 && x < e.w && y >= 0 && y < e.h)) continue;
            if (e.label[e.key(x, y)] != s) continue;
            const float ex = (float)x - core.x, ey = (float)y - core.y;
            const float d2 = ex * ex + ey * ey;
            if (d2 > far2) far2 = d2;
            const float d = e.D(x, y);
            if (d > flt_below(0.05)) { ld[n] = d; lx[n] = x; ly[n] = y; n++; }
        }
        dsm_seed out;
        memset(&out, 0, sizeof out);
        out.x = core.x; out.y = core.y; out.mean_depth = core.d; out.mean_intensity = core.i;
        out.stable = e.tmin[s] == kIntMax ? 1 : 0;
        if (n >= 16) {
            const float md = core.d;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const float r = md - ld[i];
                if (!(fabsf(r) < flt_above(e.huber))) continue;
                float nx = 0, ny = 0, nz = 0;
                const int x = lx[i], y = ly[i];
                // (the kernels' form: tabulated ray coefficients per column / row, dsm_math.h ray_coeff)
                const float rx0 = ray_coeff(x, e.K.cx, e.K.fx), rx1 = ray_coeff(x + 1, e.K.cx, e.K.fx);
                const float ry0 = ray_coeff(y, e.K.cy, e.K.fy), ry1 = ray_coeff(y + 1, e.K.cy, e.K.fy);
                if (x >= 1 && x <= e.w - 2 && y >= 1 && y <= e.h - 2)
                    pixel_normal_rays(rx0, rx1, ry0, ry1, ld[i], e.D(x + 1, y), e.D(x, y + 1), nx, ny, nz);
                ln[m * 3] = nx; ln[m * 3 + 1] = ny; ln[m * 3 + 2] = nz;
                lp[m * 3] = rx0 * ld[i]; lp[m * 3 + 1] = ry0 * ld[i]; lp[m * 3 + 2] = ld[i];
                m++;
            }
            if (!((float)m / (float)n < flt_above(0.8))) {
                float nx = 0, ny = 0, nz = 0, nb = 0, mx = 0, my = 0, mz = 0;
                for (int i = 0; i < m; i++) {
                    nx += ln[i * 3]; ny += ln[i * 3 + 1]; nz += ln[i * 3 + 2];
                    mx += lp[i * 3]; my += lp[i * 3 + 1]; mz += lp[i * 3 + 2];
                }
                const float len = sqrtf(nx * nx + ny * ny + nz * nz);
                nx = nx / len; ny = ny / len; nz = nz / len;
                mx /= (float)m; my /= (float)m; mz /= (float)m;
                for (int i = 0; i < m; i++) { lp[i * 3] -= mx; lp[i * 3 + 1] -= my; lp[i * 3 + 2] -= mz; }
                char prev_core[256] = {0};
                bool changed_any = false;
                e.gn_seeds++;
                for (int it = 0; it < 5; it++) {
                    double acc[20];
                    float res[256];
                    int cls[256];
                    bool same = it > 0;
                    for (int i = 0; i < m; i++) {
                        res[i] = lp[i * 3] * nx + lp[i * 3 + 1] * ny + lp[i * 3 + 2] * nz + nb;
                        const int cl = huber_class32(res[i], flt_above(e.huber));
                        same = same && ((cl == 0) == (prev_core[i] != 0));
                        cls[i] = cl;
                        prev_core[i] = cl == 0;
                    }
                    if (it == 0) { bool any = false; for (int i = 0; i < m; i++) any = any || cls[i] != 0; if (any) e.gn_first_noncore++; }
                    if (it > 0) { e.gn_steps++; if (!same) e.gn_steps_mask_changed++; if (!same) changed_any = true; }
                    bool all_core = true;
                    for (int i = 0; i < m; i++) all_core = all_core && cls[i] == 0;
                    if (all_core) e.exact_stats[4]++;
                    bool sums_ok = all_core; // every sum of this step spans <= 21 binades
                    for (int lane = 0; lane < 20; lane++) {
                        const bool is_j = lane >= 16;
                        const int ta = lane < 16 ? (lane & 3) : ((lane - 16) & 3), tb = (lane >> 2) & 3;
                        double a = 0.0;
                        int e_min = 1 << 30, e_max = -(1 << 30);
                        float terms[256];
                        for (int i = 0; i < m; i++) {
                            const float p4[4] = {lp[i * 3], lp[i * 3 + 1], lp[i * 3 + 2], 1.0f};
                            const float X = is_j ? res[i] : p4[ta], Y = is_j ? p4[ta] : p4[tb];
                            a += gn_term(is_j, X, Y, cls[i], e.huber);
                            const float t = (2 * X) * Y;
                            terms[i] = t;
                            if (t != 0.0f && t == t && !std::isinf(t)) { const int ex = std::ilogb(t); e_min = ex < e_min ? ex : e_min; e_max = ex > e_max ? ex : e_max; }
                            else if (t != 0.0f) { e_max = 1 << 29; e_min = 0; } // NaN / inf: never exact
                        }
                        acc[lane] = a;
                        // (H(3,3) is an integer count; the upper triangle repeats the lower; J lanes matter on every step, H lanes on the first)
                        const bool counted = is_j || (it == 0 && ta <= tb && !(ta == 3 && tb == 3));
                        if (all_core && counted) {
                            const bool ok = e_max < e_min || e_max - e_min <= 21;
                            sums_ok = sums_ok && ok;
                            if (ok) { // any order gives the ordered sum's bits: check a 16-way strided tree
                                double part[16] = {0};
                                for (int i = 0; i < m; i++) part[i & 15] += (double)terms[i];
                                for (int w = 8; w >= 1; w >>= 1) for (int q = 0; q < w; q++) part[q] += part[q + w];
                                if (memcmp(&part[0], &a, sizeof a) != 0) e.exact_stats[3]++;
                            }
                        }
                    }
                    if (it > 0 && all_core) {
                        auto range_of = [&](auto get) {
                            int lo = 1 << 30, hi = -(1 << 30);
                            for (int i = 0; i < m; i++) { const float v = get(i); if (v != 0.0f) { const int ex = std::ilogb(v); lo = ex < lo ? ex : lo; hi = ex > hi ? ex : hi; } }
                            return hi < lo ? 0 : hi - lo;
                        };
                        const int rr = range_of([&](int i) { return res[i]; });
                        bool ok = rr <= 21;
                        for (int a2 = 0; a2 < 3; a2++) ok = ok && rr + range_of([&](int i) { return lp[i * 3 + a2]; }) + 1 <= 21;
                        if (ok) e.exact_operand_pass++;
                    }
                    if (it == 0 && sums_ok) e.exact_stats[0]++;
                    if (it > 0) { e.exact_stats[2]++; if (sums_ok) e.exact_stats[1]++; }
                    if (it > 0 && all_core && same) {
                        // the kernel's form of this step (k_seed_fit, round 6): J summed as four interleaved partial sums per
                        // component (blocks of eight elements dealt out in turn), used only where the test of dsm_math.h says
                        // that no addition of any order rounds -- then it must BE the ordered sum
                        e.cert_stats[0]++;
                        bool exact = true;
                        double Jt[4];
                        for (int a = 0; a < 4; a++) {
                            double part[4] = {0, 0, 0, 0};
                            float pabs[4] = {0, 0, 0, 0};
                            uint32_t pmin[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                            for (int i = 0; i < m; i++) {
                                const float pa = a < 3 ? lp[i * 3 + a] : 1.0f;
                                const float t = res[i] * pa;
                                part[(i >> 3) & 3] += (double)t;
                                pabs[(i >> 3) & 3] += fabsf(t);
                                const uint32_t key = gn_min_key(t);
                                pmin[(i >> 3) & 3] = key < pmin[(i >> 3) & 3] ? key : pmin[(i >> 3) & 3];
                            }
                            Jt[a] = 2.0 * ((part[0] + part[2]) + (part[1] + part[3]));
                            const uint32_t kmin = std::min(std::min(pmin[0], pmin[2]), std::min(pmin[1], pmin[3]));
                            exact = exact && gn_sum_is_exact((pabs[0] + pabs[2]) + (pabs[1] + pabs[3]), gn_min_key_value(kmin));
                        }
                        if (exact) {
                            e.cert_stats[1]++;
                            if (memcmp(Jt, acc + 16, sizeof Jt) != 0) e.cert_stats[2]++;
                        }
                    }
                    gn_step(acc, acc + 16, nx, ny, nz, nb);
                }
                if (changed_any) e.gn_seeds_mask_changed++;
                plane_finish(nx, ny, nz, nb, mx, my, mz);
                const SeedGeom g = seed_geometry(e.K, core.x, core.y, md, nx, ny, nz, nb);
                out.norm_x = g.nx; out.norm_y = g.ny; out.norm_z = g.nz;
                out.posi_x = g.px; out.posi_y = g.py; out.posi_z = g.pz;
                out.mean_depth = g.mean_depth; out.view_cos = g.view_cos;
                out.size = sqrtf(far2);
            }
        }
        e.seeds[s] = out;
    }
}

SeedView view_of(const dsm_seed &sp) {
    SeedView sd;
    sd.size = sp.size; sd.nx = sp.norm_x; sd.ny = sp.norm_y; sd.nz = sp.norm_z;
    sd.px = sp.posi_x; sd.py = sp.posi_y; sd.pz = sp.posi_z;
    sd.view_cos = sp.view_cos; sd.mean_depth = sp.mean_depth; sd.mean_intensity = sp.mean_intensity;
    return sd;
}

void fuse(Emu &e, int ref_idx, const float *pose, const float *inv, dsm_surfel *local, int M) {
    FuseConst fc;
    fc.k = e.K; fc.far_d = e.far_d; fc.near_d = e.near_d;
    fc.baseline = e.baseline; fc.disp_err = e.disp_err; fc.min_tol = e.min_tol; fc.w = e.w; fc.h = e.h;
    fuse_const_prepare(fc);
    for (int i = 0; i < M; i++) {
        Surfel s;
        memcpy(&s, &local[i], sizeof s);
        int ui, vi;
        float pc[3], nc[3];
        FuseOutcome oc = fuse_project(fc, ref_idx, inv, s, ui, vi, pc, nc);
        if (oc == kFuseNeedPixel) {
            const int sidx = e.label[e.key(ui, vi)];
            const SeedView none = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // label -1: the all-zero seed (k_fuse_surfels)
            const SeedView sv = sidx >= 0 ? view_of(e.seeds[sidx]) : none;
            oc = fuse_update(fc, ref_idx, pose, s, pc, nc, e.D(ui, vi), sv, depth_weight(sv.mean_depth));
            if (oc == kFuseFused) e.seeds[sidx].fused = 1;
        }
        if (oc == kFuseDeleted) local[i].update_times = 0;
        else if (oc == kFuseFused) memcpy(&local[i], &s, sizeof s);
    }
}

int spawn(Emu &e, int ref_idx, const float *pose, dsm_surfel *fresh) {
    int k = 0;
    for (int s = 0; s < e.S; s++) {
        const SeedView sd = view_of(e.seeds[s]);
        if (!seed_spawns(sd, e.seeds[s].fused != 0)) continue;
        const Surfel n = spawn_surfel(e.K, ref_idx, pose, sd);
        memcpy(&fresh[k++], &n, sizeof n);
    }
    return k;
}

// parallel-exact form of SM.cpp:1087-1109, as k_hole_scan + k_compact do it
int compact(dsm_surfel *local, int M, const dsm_surfel *fresh, int K) {
    std::vector<int> holes, rank(M + 1, 0);
    std::vector<char> is_hole(M, 0);
    for (int i = 0; i < M; i++) {
        rank[i] = (int)holes.size();
        if (local[i].update_times == 0) { holes.push_back(i); is_hole[i] = 1; }
    }
    const int k = (int)holes.size();
    if (K >= k) {
        for (int j = 0; j < K; j++) local[j < k ? holes[k - 1 - j] : M + (j - k)] = fresh[j];
        return M + K - k;
    }
    const int r = k - K, cut = M - r;
    std::vector<dsm_surfel> snapshot(local, local + M); // reads must not see this pass's writes
    for (int j = 0; j < K; j++) local[holes[k - 1 - j]] = fresh[j];
    for (int i = 0; i < r; i++) {
        const int tgt = holes[r - 1 - i];
        if (tgt >= cut) continue;
        int src = M - 1 - i;
        while (is_hole[src] && rank[src] < r) src = M - 1 - (r - 1 - rank[src]);
        local[tgt] = is_hole[src] ? fresh[k - 1 - rank[src]] : snapshot[src];
    }
    return cut;
}
} // namespace

extern "C" {

void *emu_create(int w, int h, float fx, float fy, float cx, float cy, float far_d, float near_d, int rgbd) {
    Emu *e = new Emu();
    e->w = w; e->h = h; e->gw = w / kCell; e->gh = h / kCell; e->S = e->gw * e->gh;
    e->K = {fx, fy, cx, cy};
    e->far_d = far_d; e->near_d = near_d;
    if (rgbd) { e->huber = 0.05; e->baseline = 0.08; e->disp_err = 1.0; e->min_tol = 0.05; }
    else { e->huber = 0.4; e->baseline = 0.5; e->disp_err = 4.0; e->min_tol = 0.1; }
    e->label.assign((size_t)w * h, 0); e->cand.assign((size_t)w * h, 0);
    e->tmin.assign(e->S, -1); e->stable_stage.assign(e->S, 0);
    e->core.resize(e->S); e->stage.resize(e->S); e->inv_depth.resize(e->S); e->seeds.resize(e->S);
    return e;
}
void emu_destroy(void *p) { delete (Emu *)p; }
void emu_set_order_salt(void *p, int salt) { ((Emu *)p)->order_salt = salt; }
void emu_cert_stats(void *p, long long *out) { for (int i = 0; i < 3; i++) out[i] = ((Emu *)p)->cert_stats[i]; }
void emu_exact_sum_stats(void *p, long long *out) { for (int i = 0; i < 5; i++) out[i] = ((Emu *)p)->exact_stats[i]; out[5] = ((Emu *)p)->exact_operand_pass; }
// pick_seed_fast over every pixel assigned so far: out[24] (12..14: undecided picks by sweep; 7: seeds with a non-core residual at step 1; 8..11: fitted seeds, seeds with a class change after step 1, steps 2..5, steps with a change);
// out[0..7] = [pixels, unsure, answered differently from pick_seed, costs checked,
// bound violated, 64-pixel row segments (waves) with an unsure pixel, sweeps]
void emu_fast_pick_stats(void *p, long long *out) {
    Emu &e = *(Emu *)p;
    out[5] = (long long)e.unsure_waves.size(); out[6] = e.sweep_id;
    out[7] = e.gn_first_noncore;
    out[8] = e.gn_seeds; out[9] = e.gn_seeds_mask_changed; out[10] = e.gn_steps; out[11] = e.gn_steps_mask_changed;
    out[0] = e.fast_total; out[1] = e.fast_unsure; out[2] = e.fast_mismatches; out[3] = e.fast_checked; out[4] = e.fast_bound_violations;
    for (int i = 0; i < kSweeps; i++) out[12 + i] = e.unsure_by_sweep[i]; // (out[24])
}

int emu_fuse_map(void *p, int ref_idx, const uint8_t *img, size_t img_step, const float *depth, size_t depth_step,
                 const float *pose16, dsm_surfel *local, int *n_local, int cap, int *n_new) {
    Emu &e = *(Emu *)p;
    e.img = img; e.img_step = img_step; e.dep = depth; e.dep_step = depth_step;
    init_seeds(e);
    for (int sweep = 0; sweep < kSweeps; sweep++) {
        assign(e, sweep == 0);
        if (sweep) { resolve(e); apply(e); }
        update_seeds(e, sweep);
    }
    seed_planes(e);
    float inv[16];
    inverse4<float>(pose16, inv);
    fuse(e, ref_idx, pose16, inv, local, *n_local);
    std::vector<dsm_surfel> fresh(e.S);
    const int K = spawn(e, ref_idx, pose16, fresh.data());
    if (*n_local + K > cap) return -1;
    *n_local = compact(local, *n_local, fresh.data(), K);
    *n_new = K;
    return 0;
}
void emu_get_labels(void *p, int32_t *out) { Emu &e = *(Emu *)p; memcpy(out, e.label.data(), sizeof(int32_t) * e.label.size()); }
void emu_get_seeds(void *p, dsm_seed *out) { Emu &e = *(Emu *)p; memcpy(out, e.seeds.data(), sizeof(dsm_seed) * e.seeds.size()); }
int emu_compact(dsm_surfel *local, int n, const dsm_surfel *fresh, int k) { return compact(local, n, fresh, k); }
int emu_div100_mismatches(const float *x, int n) {
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const double a = (double)x[i] / 100.0, b = div_by_100((double)x[i]);
        bad += memcmp(&a, &b, 8) != 0;
    }
    return bad;
}
// fp32 forms of the reference's double-typed compares (dsm_math.h, flt_below / flt_above): every float within
// `reach` floats of +-c, plus the specials, through all four relations against c and -c
int emu_threshold_mismatches(double c, int reach) {
    int bad = 0;
    const float lo = flt_below(c), hi = flt_above(c);
    bad += !((double)lo <= c && (double)hi >= c);
    bad += !(lo == hi || nextafterf(lo, INFINITY) == hi); // neighbours, or c is a float
    float specials[] = {0.0f, -0.0f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 3.4e38f, -3.4e38f};
    for (int sign = -1; sign <= 1; sign += 2) {
        float x = (float)c * (float)sign;
        for (int i = 0; i < reach; i++) x = nextafterf(x, -INFINITY);
        for (int i = 0; i <= 2 * reach + (int)(sizeof(specials) / sizeof(float)); i++) {
            const float v = i <= 2 * reach ? x : specials[i - 2 * reach - 1];
            const double d = (double)v;
            bad += (d > c) != (v > lo);
            bad += (d < c) != (v < hi);
            bad += (d <= c) != (v <= lo);
            bad += (d >= c) != (v >= hi);
            bad += (d > -c) != (v > -hi);
            bad += (d < -c) != (v < -lo);
            bad += (d < c && d > -c) != (fabsf(v) < hi);
            x = nextafterf(x, INFINITY);
        }
    }
    return bad;
}
// the Newton step of the robust mean in fp32 vs the reference's double expression (FF.cpp:553)
int emu_newton_step_mismatches(const float *a, const int *n_core, int n) {
    int bad = 0;
    for (int i = 0; i < n; i++) {
        float b = 0;
        for (int k = 0; k < n_core[i]; k++) b += 2; // as the reference accumulates it
        const float ref = (float)((double)(-a[i]) / ((double)b + 10.0)), got = huber_newton_step(a[i], b);
        bad += memcmp(&ref, &got, 4) != 0 && !(ref != ref && got != got);
    }
    return bad;
}
// the association's depth tolerance (dsm_math.h, fuse_depth_tolerance) vs the reference's double expression
// (FF.cpp:250-253), and the normal's renormalisation as fp32 divides vs FF.cpp:287-291
int emu_fuse_fp32_mismatches(const float *z, const float *a, const float *b, int n, int rgbd, float focal, int *used_fp32) {
    FuseConst c;
    c.k.fx = focal; c.k.fy = focal; c.k.cx = 0; c.k.cy = 0;
    c.far_d = 30.0f; c.near_d = 0.5f; c.w = c.h = 0;
    if (rgbd) { c.baseline = 0.08; c.disp_err = 1.0; c.min_tol = 0.05; }
    else { c.baseline = 0.5; c.disp_err = 4.0; c.min_tol = 0.1; }
    fuse_const_prepare(c);
    *used_fp32 = c.tol32 ? 1 : 0;
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const float cam_f = camera_focal(c.k);
        float tol = (float)((double)(z[i] * z[i]) / (c.baseline * (double)cam_f) * c.disp_err);
        tol = (float)((double)tol < c.min_tol ? c.min_tol : (double)tol);
        const float got = fuse_depth_tolerance(c, z[i]);
        bad += memcmp(&tol, &got, 4) != 0 && !(tol != tol && got != got);
        const float q_ref = (float)((double)a[i] / (double)b[i]), q = a[i] / b[i];
        bad += memcmp(&q_ref, &q, 4) != 0 && !(q_ref != q_ref && q != q);
    }
    return bad;
}
// the robust mean depth with the settled-mean shortcut (dsm_math.h: a +inf member needs no pass) against the reference's loop
// as written (FF.cpp:530-556, no shortcut), on lists that hold +inf members, overflowing sums and ordinary depths
int emu_settled_mean_mismatches(const float *lists, const int *len, int n_lists, int stride, double huber) {
    int bad = 0;
    for (int i = 0; i < n_lists; i++) {
        const float *d = lists + (size_t)i * stride;
        const int n = len[i];
        float sum = 0;
        for (int k = 0; k < n; k++) sum += d[k];
        float mean_depth = sum / n;
        for (int it = 0; it < 5; it++) { // the reference, statement by statement
            float sum_a = 0, sum_b = 0;
            for (int k = 0; k < n; k++) {
                float residual = mean_depth - d[k];
                if (residual < huber && residual > -huber) { sum_a += 2 * residual; sum_b += 2; }
                else sum_a += residual > 0 ? huber : -1 * huber;
            }
            float delta_depth = -sum_a / (sum_b + 10.0);
            mean_depth = mean_depth + delta_depth;
            if (delta_depth < 0.01 && delta_depth > -0.01) break;
        }
        const float got = huber_mean_depth(d, n, sum, huber);
        bad += memcmp(&mean_depth, &got, 4) != 0 && !(mean_depth != mean_depth && got != got);
    }
    return bad;
}
// table-driven inverse (the form the HIP kernel evaluates lane-parallel) vs the closed form
void emu_inverse4d(const double *a, double *closed, double *tabled) { inverse4<double>(a, closed); inverse4_tabled<double>(a, tabled); }

} // extern "C"
