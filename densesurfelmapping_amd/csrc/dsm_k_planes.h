// dsm_k_planes.h -- from superpixels to planes: k_seed_points (wave form), k_pixel_normals + k_seed_stats (lane form),
// k_seed_fit, k_seed_finish.  FF.cpp:104-188, 644-712, 792-914.  Included by dsm_kernels.hip.
#pragma once
#include "dsm_k_common.h"

namespace dsm {

// ------------------------------------------------------------------------------ seed planes
// calculate_spaces / calculate_pixels_norms / calculate_sp_depth_norms + get_huber_norm (FF.cpp:644-712, 792-914,
// 104-188) in three kernels: k_pixel_normals (thread per pixel), k_seed_stats (lane per seed) and k_seed_fit.  The
// reference's 36 B/pixel space_map never exists in memory (a back-projection is two multiplies by tabulated ray
// coefficients), its norm_map only for the pixels that are read.  Every order-sensitive sum runs in the reference's
// order.  (Until round 3 a wave-per-seed kernel, k_seed_points, did the work of the first two and handed the centred
// inlier points to the fit through a [S][3][232] buffer: 611 VALU instructions per seed, 97 MB of hand-off traffic
// per batched launch.)
//
// k_seed_fit, FOUR seeds per wave: the 5 Huber-weighted Gauss-Newton steps.  A step's 10 + 4 double accumulators
// (the Hessian is symmetric: H(a,b) and H(b,a) add the same products) are independent ordered sums,
//     H(a,b) += (double)((2*p_a)*p_b),  J(a) += (double)((2*r)*p_a)   (p_3 = 1; core residuals)
//     J(a)   += +-hr*(double)p_a                                      (Huber tails)
// i.e. (double)((2*X)*Y) with per-lane operand columns X, Y out of {p0, p1, p2, 1, r}: 14 lanes of a 16-lane group
// each carry one, so four seeds fill the wave where one seed used 20 of 64 lanes (the Gauss-Newton steps were 60 %
// of the one-kernel form's time).  The 4x4 solve is one lane per 2x2 determinant / adjugate entry, again per group.
// ---- seed statistics, one WAVE per seed: the launch form for one handle or a few (frame groups), where the kernel's
// latency counts -- a wave gathers its window with 4 pixels per lane and ends in ~11 us; the lane-per-seed pair below
// walks 256 pixels per lane (35-45 us) and pays only when thousands of seeds share a launch.  Same header out.
// Gather the member pixels with valid depth (window row-major order), keep the depth inliers, recompute their
// back-projections and forward-difference normals from the depth plane, sum normals and points in the reference's order:
// operands are produced lane-parallel, parked in LDS as structure-of-arrays columns and block-fetched; the six fp32
// sums are six lanes.
constexpr int kCols = 6; // LDS columns per wave of k_seed_points, reused across phases:
//   gather / inlier phase:  depth list | packed xy | -       | n0       | n1   | n2
//   sums phase:             p0         | p1        | p2      | n0       | n1   | n2
// (p0/p1 overwrite the depth/xy lists in place: a chunk's 64 entries are read before its compacted
// entries, which land at or below the same indices, are written)
// column stride: 260 floats shifts successive columns by 4 banks, so that lanes streaming different
// columns at the same element offset (ds_read_b128) do not collide
constexpr int kColStride = kWin * kWin + 4;

template <bool BATCH> __global__ __launch_bounds__(256) void k_seed_points(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    __shared__ __attribute__((aligned(16))) float s_col[4][kCols][kColStride];
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int s = __builtin_amdgcn_readfirstlane(seed_of_block(blk.x, wv, c->gw, c->gh)); // scalar, see k_update_seeds
    if (s < 0) return;
    if (BATCH && lane == 0) fit_order(c)[s] = s;
    const FrameParams &fp = frame_params(c);
    const float *dep = frame_depth(c, fp);
    const int w = c->w, h = c->h, pitch = c->pitch;
    stamp(c, 3, s, 0, lane);
    const double hr = c->huber;
    const float hr_above = flt_above(hr); // the Huber class tests in fp32 (dsm_math.h)
    const float4 core = c->core[s];
    int gx, gy;
    seed_cell(c, s, gx, gy);
    const int wx0 = gx * kCell + kCell / 2 - kCell, wy0 = gy * kCell + kCell / 2 - kCell;
    float *P0 = s_col[wv][0], *P1 = s_col[wv][1], *P2 = s_col[wv][2];
    float *N0 = s_col[wv][3], *N1 = s_col[wv][4], *N2 = s_col[wv][5];
    float *ld = P0;
    int *lxy = reinterpret_cast<int *>(P1);

    // ---- members with depth > 0.05, and the superpixel radius (FF.cpp:813-838)
    int n = 0;
    float far2 = 0.0f;
    int lab[4];
    float pd[4];
    const int x0 = wx0 + (lane & (kWin - 1)), y0 = wy0 + (lane >> 4);
    const int key0 = __mul24(y0, pitch) + x0, row4 = 4 * pitch; // pixel keys as byte offsets: see ld_off
#pragma unroll
    for (int k = 0; k < 4; k++) { // 8 independent loads, one round trip
        const int y = y0 + 4 * k;
        const bool in = x0 >= 0 && x0 < w && y >= 0 && y < h;
        const unsigned o4 = in ? (unsigned)(key0 + k * row4) << 2 : 0u;
        const int l = (int)ld_off(c->label, o4 >> 1); // (16 bits: kNoLabel equals no seed)
        lab[k] = in ? l : -1;
        pd[k] = ld_off(dep, o4);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idx = k * 64 + lane;
        const int x = wx0 + (idx & (kWin - 1)), y = wy0 + (idx >> 4);
        const bool mem = lab[k] == s;
        float d = 0.0f;
        if (mem) {
            d = pd[k];
            const float ex = (float)x - core.x, ey = (float)y - core.y;
            const float d2 = ex * ex + ey * ey;
            if (d2 > far2) far2 = d2;
        }
        const bool ok = mem && d > flt_below(0.05); // (double)d > 0.05
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const int pos = n + rank_below(m);
            ld[pos] = d;
            lxy[pos] = x | (y << 16);
        }
        n += __popcll(m);
    }
    far2 = wave_max(far2);
    wave_lds_sync();
    stamp(c, 3, s, 1, lane);

    int m_fit = 0; // inliers handed to the fit; 0: the seed keeps its defaults
    wave_priority(n); // long lists first: they are the kernel's critical path
    if (n >= 16) { // FF.cpp:841
        // ---- depth inliers: their pixel normals and back-projected points, in order (FF.cpp:846-861)
        const float md = core.w;
        int m_in = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bool ok = false, interior = false;
            float d = 0.0f, d_right = 0.0f, d_down = 0.0f, rx0 = 0.0f, rx1 = 0.0f, ry0 = 0.0f, ry1 = 0.0f;
            int x = 0, y = 0;
            if (i < n) {
                d = ld[i];
                const int xy = lxy[i];
                x = xy & 0xffff; y = xy >> 16;
                interior = x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2; // FF.cpp:670-677
                if (interior) { // neighbours for the forward differences, fetched before they are known to be needed
                    const unsigned o4 = (unsigned)(__mul24(y, pitch) + x) << 2;
                    d_right = ld_off(dep, o4 + 4u);
                    d_down = ld_off(dep, o4 + ((unsigned)pitch << 2));
                }
                rx0 = ld_off(c->ray_x, (unsigned)x << 2); rx1 = ld_off(c->ray_x, ((unsigned)x << 2) + 4u);
                ry0 = ld_off(c->ray_y, (unsigned)y << 2); ry1 = ld_off(c->ray_y, ((unsigned)y << 2) + 4u);
                const float r = md - d;
                ok = fabsf(r) < hr_above; // (double)r < hr && (double)r > -hr
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int pos = m_in + rank_below(m);
                float nx = 0.0f, ny = 0.0f, nz = 0.0f;
                if (interior) pixel_normal_rays(rx0, rx1, ry0, ry1, d, d_right, d_down, nx, ny, nz);
                N0[pos] = nx; N1[pos] = ny; N2[pos] = nz;
                P0[pos] = rx0 * d; P1[pos] = ry0 * d; P2[pos] = d; // back_project, FF.cpp:91-97
            }
            m_in += __popcll(m);
        }
        // pad every column the ordered sums stream to a multiple of 16 with +0.0f (see ordered_sum)
        wave_lds_sync();
        pad_column(P0, m_in, lane); pad_column(P1, m_in, lane); pad_column(P2, m_in, lane);
        pad_column(N0, m_in, lane); pad_column(N1, m_in, lane); pad_column(N2, m_in, lane);
        wave_lds_sync();
        stamp(c, 3, s, 2, lane);
        if (m_in > kGnCap) {
            // more inliers than a superpixel can have (15 x 15 = 225 members): the label image did not come from
            // k_assign (dsm_debug_set_label_buffer).  The hand-off to the fit holds kGnCap points: report, no fit.
            if (lane == 0) atomicOr(c->status, kStatusBadLabels);
        } else if (!((float)m_in / (float)n < flt_above(0.8))) { // FF.cpp:862, (double)ratio < 0.8
            // sequential fp32 sums, FF.cpp:852-857 and 111-116
            // six ordered sums at once: lane q < 6 streams column q (n0 n1 n2 p0 p1 p2)
            const float part = ordered_sum(s_col[wv][lane < 3 ? 3 + lane : lane < 6 ? lane - 3 : 0], m_in);
            float nx = __shfl(part, 0), ny = __shfl(part, 1), nz = __shfl(part, 2);
            float mx = __shfl(part, 3), my = __shfl(part, 4), mz = __shfl(part, 5);
            const float len = sqrtf(nx * nx + ny * ny + nz * nz);
            nx = nx / len; ny = ny / len; nz = nz / len;
            mx /= (float)m_in; my /= (float)m_in; mz /= (float)m_in;
            if (lane == 0) {
                GnHeader hd;
                hd.m_in = m_in;
                hd.nx = nx; hd.ny = ny; hd.nz = nz;
                hd.mx = mx; hd.my = my; hd.mz = mz;
                hd.far2 = far2;
                c->gn_hdr[s] = hd;
            }
            m_fit = m_in;
        }
    }
    if (m_fit == 0 && lane == 0) c->gn_hdr[s].m_in = 0;
    stamp(c, 3, s, 5, lane);
    if (kWaveStamps && c->stamps && lane == 0) c->stamps[((int64_t)3 * c->n_seed + s) * 8 + 7] = n;
}

// ---- seed statistics without a wave per seed
// k_pixel_normals, one thread per pixel: the forward-difference normal (FF.cpp:664-712) of every pixel that is a depth
// inlier of its own superpixel (FF.cpp:846-850: member, depth > 0.05, |mean depth - depth| < HUBER_RANGE), written into
// a 12 B/pixel plane; the other pixels' entries are stale and never read.  (calculate_pixels_norms computes all of them;
// only these are ever read, FF.cpp:852-857.)
// A thread takes a COLUMN of four pixels (a workgroup 64 x 16): a quarter of the waves -- a launch over 32 frames was 238 000 waves of
// ~150 instructions each, a third of them address arithmetic and context loads, and beside three other batches' kernels it took
// four times as long as alone (wave dispatch) -- the depth below a pixel is the next pixel's own, and everything a pixel may need
// is requested before the first result is looked at.
constexpr int kNormalRows = 4;
template <bool BATCH> __global__ __launch_bounds__(256) void k_pixel_normals(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const FrameParams &fp = frame_params(c);
    const float *dep = frame_depth(c, fp);
    const int w = c->w, h = c->h, pitch = c->pitch;
    const int x = blk.x * 64 + (threadIdx.x & 63);
    const int y0 = blk.y * (4 * kNormalRows) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kNormalRows; // (wave-uniform)
    if (x >= w || y0 >= h) return;
    const unsigned p0 = (unsigned)(__mul24(y0, pitch) + x), row = (unsigned)pitch;
    // labels and depths of the column (and the row below it), the right-hand neighbours: all in flight at once.  Rows past the
    // image read the last row again (never used: a pixel there is skipped below)
    int l[kNormalRows];
    float d[kNormalRows + 1], dr[kNormalRows];
    const bool has_right = x <= w - 2;
#pragma unroll
    for (int r = 0; r <= kNormalRows; r++) {
        const unsigned p = p0 + row * (unsigned)min(r, h - 1 - y0);
        d[r] = ld_off(dep, p << 2);
        if (r < kNormalRows) {
            l[r] = label_at(c->label, p);
            dr[r] = has_right ? ld_off(dep, (p << 2) + 4u) : 0.0f;
        }
    }
    const float rx0 = ld_off(c->ray_x, (unsigned)x << 2), rx1 = ld_off(c->ray_x, ((unsigned)x << 2) + 4u);
    float ry[kNormalRows + 1];
#pragma unroll
    for (int r = 0; r <= kNormalRows; r++) ry[r] = ld_off(c->ray_y, (unsigned)min(y0 + r, h) << 2); // (ray_y has h + 1 entries)
    // only the depth inliers of their own superpixel are ever read (k_seed_stats asks for exactly those): nothing is
    // stored for any other pixel; an inlier on the image border has no normal (FF.cpp:670-677) and stores zeros
    float md[kNormalRows];
    bool member[kNormalRows];
#pragma unroll
    for (int r = 0; r < kNormalRows; r++) {
        member[r] = y0 + r < h && l[r] >= 0 && d[r] > flt_below(0.05); // (double)d > 0.05
        md[r] = member[r] ? ld_off(reinterpret_cast<const float *>(c->core), ((unsigned)l[r] << 4) + 12u) : 0.0f;
    }
    const float hub = flt_above(c->huber);
#pragma unroll
    for (int r = 0; r < kNormalRows; r++) {
        if (!(member[r] && fabsf(md[r] - d[r]) < hub)) continue;
        const int y = y0 + r;
        float nx = 0.0f, ny = 0.0f, nz = 0.0f;
        if (x >= 1 && has_right && y >= 1 && y <= h - 2) pixel_normal_rays(rx0, rx1, ry[r], ry[r + 1], d[r], dr[r], d[r + 1], nx, ny, nz);
        float *o = reinterpret_cast<float *>(reinterpret_cast<char *>(c->normals) + (p0 + row * (unsigned)r) * 12u);
        o[0] = nx; o[1] = ny; o[2] = nz;
    }
}

// k_seed_stats, ONE LANE PER SEED (64 consecutive seeds per wave): calculate_sp_depth_norms up to the plane fit's
// starting point (FF.cpp:813-871) and the head of get_huber_norm (FF.cpp:111-120).  A lane walks its seed's 16x16 window
// twice in row-major order: once over labels and depths (member count with depth, radius, depth inliers and which
// pixels they are, the ordered sums of their back-projected points), once over the normal plane for exactly those pixels
// (ordered sum of the inliers' normals; a pixel that is no inlier adds +0, which leaves a running sum that starts at +0
// unchanged, bit for bit).  The wave-per-seed form spent 611 VALU instructions per seed on this, most of them per-seed
// bookkeeping and six-lane sums; a lane spends ~26 per window pixel for 64 seeds at once.
struct StatRow { // one window row of one lane: labels and depths
    LabelQuad lab[4];
    float4 dp[4];
};
template <bool BATCH> __global__ __launch_bounds__(64) void k_seed_stats(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    // which pixels of every window row are depth inliers of the lane's seed (bit j = window column j): found by the first
    // walk, and all the second walk needs to know -- it fetches normals only for the quads that hold one and no labels at all
    __shared__ unsigned short s_inl[kWin + 2][64];
    const int lane = lane_id();
    const int S = c->n_seed;
    const int s = (((S + 63) >> 6) - 1 - blk.x) * 64 + lane; // bottom rows first, see seed_of_block
    const bool live = s < S;
    const int sc = live ? s : S - 1;
    const FrameParams &fp = frame_params(c);
    const float *dep = frame_depth(c, fp);
    const int w = c->w, h = c->h, pitch = c->pitch;
    int gx, gy;
    seed_cell(c, sc, gx, gy);
    const int wx0 = gx * kCell + kCell / 2 - kCell, wy0 = gy * kCell + kCell / 2 - kCell;
    const float4 core = c->core[sc];
    const float md = core.w;
    const float hr_above = flt_above(c->huber); // the Huber class tests in fp32 (dsm_math.h)
    const unsigned s_match = live ? (unsigned)s : (unsigned)kNoSeed;
    s_inl[kWin][lane] = s_inl[kWin + 1][lane] = 0; // (the second walk's loop runs two rows past the window)
    int qx[4];                                  // window quads as pixel offsets within a row, redirected into the row (see k_update_seeds)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = wx0 + 4 * q;
        qx[q] = x < 0 ? 0 : (x > pitch - 4 ? pitch - 4 : x);
    }
    bool col_in[kWin];
    float exx[kWin], rx[kWin];
#pragma unroll
    for (int j = 0; j < kWin; j++) {
        const int x = wx0 + j;
        col_in[j] = (unsigned)x < (unsigned)w;
        const float ex = (float)x - core.x;
        exx[j] = ex * ex; // FF.cpp:820-823: the radius term of this column
        rx[j] = ld_off(c->ray_x, (unsigned)(x < 0 ? 0 : (x > w ? w : x)) << 2);
    }
    auto row_offset = [&](int r) {
        int y = wy0 + r;
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        return (unsigned)__mul24(y, pitch);
    };

    // ---- first walk: labels and depths
    int n = 0, m_in = 0;
    float far2 = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
    auto load_a = [&](int r) {
        StatRow R;
        const unsigned row = row_offset(r);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned o4 = (row + (unsigned)qx[q]) << 2;
            R.lab[q] = label_quad(c->label, o4 >> 2);
            R.dp[q] = ld_vec<float4>(dep, o4);
        }
        return R;
    };
    auto walk_a = [&](const StatRow &A, int r) {
        const int y = wy0 + r;
        const bool row_in = (unsigned)y < (unsigned)h;
        const unsigned s_row = row_in ? s_match : (unsigned)kNoSeed;
        const int yc = y < 0 ? 0 : (y > h ? h : y);
        const float ry = ld_off(c->ray_y, (unsigned)yc << 2);
        const float ey = (float)y - core.y, eyy = ey * ey;
        unsigned bits = 0u;
#pragma unroll
        for (int j = 0; j < kWin; j++) {
            const bool mem = comp(A.lab[j >> 2], j & 3) == s_row && col_in[j];
            const float d2 = exx[j] + eyy;
            far2 = fmaxf(far2, mem ? d2 : 0.0f);               // FF.cpp:820-824, over all members
            const float d = comp(A.dp[j >> 2], j & 3);
            const bool ok = mem && d > flt_below(0.05);        // (double)d > 0.05
            n += ok ? 1 : 0;
            const bool inl = ok && fabsf(md - d) < hr_above;    // (double)r < hr && (double)r > -hr
            m_in += inl ? 1 : 0;
            bits |= inl ? 1u << j : 0u;
            sx += inl ? rx[j] * d : 0.0f;                       // back_project (FF.cpp:91-97), summed in window order (FF.cpp:111-116)
            sy += inl ? ry * d : 0.0f;
            sz += inl ? d : 0.0f;
            if ((j & 3) == 3) {
                asm volatile("" : "+v"(n), "+v"(m_in), "+v"(far2), "+v"(sx), "+v"(sy), "+v"(sz), "+v"(bits)); // see k_update_seeds
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        s_inl[r][lane] = (unsigned short)bits;
    };
    {
        StatRow B0 = load_a(0), B1 = load_a(1), B2 = load_a(2), B3;
#pragma unroll 1
        for (int r = 0; r < kWin; r += 4) {
            B3 = load_a(r + 3);
            __builtin_amdgcn_sched_barrier(0);
            walk_a(B0, r);
            if (r + 4 < kWin) B0 = load_a(r + 4);
            __builtin_amdgcn_sched_barrier(0);
            walk_a(B1, r + 1);
            if (r + 4 < kWin) B1 = load_a(r + 5);
            __builtin_amdgcn_sched_barrier(0);
            walk_a(B2, r + 2);
            if (r + 4 < kWin) B2 = load_a(r + 6);
            __builtin_amdgcn_sched_barrier(0);
            walk_a(B3, r + 3);
        }
    }
    // does this seed get a plane at all?  FF.cpp:841 (>= 16 members with depth), FF.cpp:862 (>= 80 % of them inliers)
    bool fit = live && n >= 16 && !((float)m_in / (float)n < flt_above(0.8)); // (double)ratio < 0.8
    if (fit && m_in > kGnCap) { // more inliers than a superpixel can have: the label image did not come from k_assign
        atomicOr(c->status, kStatusBadLabels);
        fit = false;
    }
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    wave_lds_sync();
    if (__ballot(fit) != 0) {
        // ---- second walk: the normals of the depth inliers, in window order (k_pixel_normals left zero where an inlier
        // has no normal).  A lane fetches the twelve floats of a quad only if the quad holds one of its inliers: on
        // average a window's 64 quads hold inliers in 20, so two thirds of the plane's lines are never asked for -- this
        // walk used to pull every window's 4.6 KB of labels and normals through an L2 that four frames share.
        struct NormRowM {
            unsigned m;
            float4 nv[12]; // 16 pixels x 3 floats
        };
        auto load_b = [&](int r) {
            NormRowM R;
            R.m = fit ? (unsigned)s_inl[r][lane] : 0u;
            const unsigned row = row_offset(r);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned o = row + (unsigned)qx[q];
                if ((R.m >> (4 * q)) & 0xfu) {
#pragma unroll
                    for (int t = 0; t < 3; t++) R.nv[3 * q + t] = ld_vec<float4>(c->normals, o * 12u + 16u * t);
                }
            }
            return R;
        };
        auto walk_b = [&](const NormRowM &A) {
#pragma unroll
            for (int j = 0; j < kWin; j++) {
                const bool mem = (A.m >> j) & 1u;
                const int e = 3 * (j & 3); // the pixel's three floats within its quad's twelve
                nx += mem ? comp(A.nv[3 * (j >> 2) + (e >> 2)], e & 3) : 0.0f;
                ny += mem ? comp(A.nv[3 * (j >> 2) + ((e + 1) >> 2)], (e + 1) & 3) : 0.0f;
                nz += mem ? comp(A.nv[3 * (j >> 2) + ((e + 2) >> 2)], (e + 2) & 3) : 0.0f;
                if ((j & 3) == 3) {
                    asm volatile("" : "+v"(nx), "+v"(ny), "+v"(nz));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        NormRowM B0 = load_b(0), B1 = load_b(1), B2;
#pragma unroll 1
        for (int r = 0; r < kWin; r += 3) { // 18 rows: the two past the window hold no inlier
            B2 = load_b(r + 2);
            __builtin_amdgcn_sched_barrier(0);
            walk_b(B0);
            B0 = load_b(r + 3 < kWin + 2 ? r + 3 : kWin + 1);
            __builtin_amdgcn_sched_barrier(0);
            walk_b(B1);
            B1 = load_b(r + 4 < kWin + 2 ? r + 4 : kWin + 1);
            __builtin_amdgcn_sched_barrier(0);
            walk_b(B2);
        }
    }
    // ---- the order in which k_seed_fit takes this wave's 64 seeds, four per wave: by the length of their lists.  A
    // group of four pads its lists to the longest one's multiple of 8 and every ordered sum runs that far, so four
    // neighbours in the grid cost their longest list each (a quarter of all seeds gets no plane at all and sits between
    // the others): grouped by length, the element loops of a frame shrink by 15 % and 6 % of the groups have nothing to
    // do.  The fit's results do not depend on which seeds share a wave.  rank = number of smaller keys; the lane breaks
    // ties, and the lanes past the last seed come last.
    {
        const unsigned key = ((live ? (fit ? (unsigned)(m_in + 7) >> 3 : 0u) : 0xffffu) << 6) | (unsigned)lane;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 64; j++) rank += (unsigned)__builtin_amdgcn_readlane((int)key, j) < key ? 1 : 0;
        fit_order(c)[(((S + 63) >> 6) - 1 - blk.x) * 64 + rank] = live ? s : -1;
    }
    if (!live) return;
    GnHeader hd;
    hd.m_in = 0;
    hd.nx = hd.ny = hd.nz = hd.mx = hd.my = hd.mz = 0.0f;
    hd.far2 = far2;
    if (fit) {
        const float len = sqrtf(nx * nx + ny * ny + nz * nz); // FF.cpp:866-871
        hd.nx = nx / len; hd.ny = ny / len; hd.nz = nz / len;
        hd.mx = sx / (float)m_in; hd.my = sy / (float)m_in; hd.mz = sz / (float)m_in; // FF.cpp:117-120
        hd.m_in = m_in;
    }
    c->gn_hdr[s] = hd;
}

// ---- the fit: four seeds per wave, sixteen lanes per seed
// LDS columns per seed: p0 | p1 | p2 | residual, padded with +0.0f up to the longest list of the four (a
// running sum that starts at +0.0 stays bit-identical when +0.0 is added, and a padded element's product is +0.0).
constexpr int kFitSeeds = 4, kFitLanes = 16, kFitCols = 4;
constexpr int kFitStride = kGnCap + 4; // 236 floats: successive columns 16 B x 59 apart -> shifted by 11 x 16 B mod 256
// LDS per wave decides how many waves of this kernel a CU holds (16.8 KB: nine), and it is sized for the longest list
// a window can give (232) while nearly every group of four seeds stays far below that.  Launches batched over
// handles -- enough waves to fill the machine several times -- therefore run the fit in two tiers: groups whose longest
// list fits kFitSmallCap in a kernel with columns of that length (9.6 KB and fewer registers: sixteen waves per CU);
// that kernel queues the few others (c->worklist, free by now; count in c->fit_big_count), and a second launch of a
// handful of workgroups in the full-length form works the queue off -- normally it finds it empty.  Same arithmetic,
// element for element; which tier a group takes changes nothing in its result.
constexpr int kFitSmallStride = kFitSmallCap + 4; // 124 floats: columns 16 B x 31 apart -> shifted by 15 x 16 B mod 256
constexpr int kFitLargeBlocks = 128;              // workgroups per handle working the queue off (they leave at once when it is empty; 16 until round 5: a feed with large superpixels -- quantised grey levels, profiles/r05_* -- queues a hundred groups per frame, and sixteen workgroups took them seven deep)
enum FitTier { kFitAll = 0, kFitSmall = 1, kFitLarge = 2 };
// accumulator of lane gl of a group: (X column, Y column); columns 0..2 = p, 3 = residual, 4 = the homogeneous 1 -- a
// shared block of eight 1.0f read at stride 0 instead of a column per seed (LDS per wave decides how many waves a CU
// holds, and this kernel is short of waves).  gl 0..8 = H(a,b), a <= b, without H(3,3); gl 9 = H(3,3) = 2 x (number of
// core elements), an integer that needs no sum; gl 10..13 = J(a); gl 14, 15 idle (they stream ones and are ignored)
__constant__ const signed char kFitX[16] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 4, 3, 3, 3, 3, 4, 4};
__constant__ const signed char kFitY[16] = {0, 1, 2, 4, 1, 2, 4, 2, 4, 4, 0, 1, 2, 4, 4, 4};
__constant__ const signed char kFitRow[16] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3, 0, 1, 2, 3, 0, 0};  // H: row a | J: a
__constant__ const signed char kFitColI[16] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3, 0, 0, 0, 0, 0, 0}; // H: column b

// Ordered double sum of this lane's accumulator over the padded lists (m8 = longest of the four, rounded up to 8).
// Blocks of 8 whose residuals are in the Huber core for all four seeds take the plain path.  Otherwise every element
// adds (double)(X*Y) * scale, scale = 1 for a core element, else hr/2 in a Jacobian lane and 0 in a Hessian lane,
// where the residual column holds +-1 instead of r for an upper / lower tail element (0 for a NaN residual): the
// tail term +-(hr/2)*(double)Y of a Jacobian lane is (double)(+-1*Y) * (hr/2) exactly, a core term times 1.0 is
// itself, and a Hessian lane adds +-0.  Branch-free and without per-element class logic; checked against the
// three-way form on 8 M random elements on the host.
// Scaling by two commutes with every rounding, so the sums are carried halved: a core term is (double)(X*Y)
// instead of (double)((2*X)*Y), a tail term +-(hr/2)*(double)Y, and the result is doubled once at the end --
// bit-identical (no overflow / underflow anywhere near these magnitudes), one multiply less per element.
// all four seeds' residuals in the Huber core (the usual case after the first step): no masks, and the next block's
// operands are fetched while this block's adds run -- a wave of this kernel has a SIMD almost to itself, so the LDS
// latency is not hidden by other waves
// (xs, ys: 1 = the operand advances with the element index, 0 = it is the shared block of ones)
__device__ __forceinline__ double fit_ordered_sum_core(const float *xc, const float *yc, int xs, int ys, int m8) {
    const float4 *x4 = reinterpret_cast<const float4 *>(xc), *y4 = reinterpret_cast<const float4 *>(yc);
    float4 xa = x4[0], xb = x4[1], ya = y4[0], yb = y4[1];
    double acc = 0.0;
    for (int b = 8; b <= m8; b += 8) {
        const int nb = b < m8 ? b >> 2 : 0; // (the last round re-reads block 0 and drops it)
        const float4 pxa = x4[nb * xs], pxb = x4[nb * xs + 1], pya = y4[nb * ys], pyb = y4[nb * ys + 1];
        acc += (double)(xa.x * ya.x); acc += (double)(xa.y * ya.y); acc += (double)(xa.z * ya.z); acc += (double)(xa.w * ya.w);
        acc += (double)(xb.x * yb.x); acc += (double)(xb.y * yb.y); acc += (double)(xb.z * yb.z); acc += (double)(xb.w * yb.w);
        xa = pxa; xb = pxb; ya = pya; yb = pyb;
    }
    return 2.0 * acc;
}

__device__ __forceinline__ double fit_ordered_sum(const float *xc, const float *yc, int xs, int ys, int m8,
                                                  const unsigned long long noncore[4], bool is_j, double hr) {
    if (__ballot((noncore[0] | noncore[1] | noncore[2] | noncore[3]) != 0) == 0) return fit_ordered_sum_core(xc, yc, xs, ys, m8);
    double acc = 0.0;
    const double k_lane = is_j ? 0.5 * hr : 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int lim = m8 - k * 64 < 64 ? m8 - k * 64 : 64;
        if (lim <= 0) break;
        for (int j = 0; j < lim; j += 8) { // 8 at a time: two operand columns, register budget
            const int b = k * 64 + j;
            const float4 xa = *reinterpret_cast<const float4 *>(xc + b * xs), xb = *reinterpret_cast<const float4 *>(xc + b * xs + 4);
            const float4 ya = *reinterpret_cast<const float4 *>(yc + b * ys), yb = *reinterpret_cast<const float4 *>(yc + b * ys + 4);
            const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const float yv[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
            const unsigned n8 = (unsigned)(noncore[k] >> j) & 0xffu; // this lane's seed
            if (__ballot(n8 != 0) == 0) {
#pragma unroll
                for (int q = 0; q < 8; q++) acc += (double)(xv[q] * yv[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double v = (double)(xv[q] * yv[q]);
                    const double scale = ((n8 >> q) & 1u) ? k_lane : 1.0;
                    acc += v * scale;
                }
            }
        }
    }
    return 2.0 * acc;
}

// x += x rotated within its row of sixteen lanes (DPP row_ror: CTRL = 0x120 + lanes), for a lane's quarter of a Jacobian sum
// and the two values its exactness test needs
template <int CTRL> __device__ __forceinline__ void quarter_fold(double &part, float &t_sum, uint32_t &t_min) {
    const long long pb = __double_as_longlong(part);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)pb, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(pb >> 32), CTRL, 0xf, 0xf, false);
    part += __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
    t_sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t_sum), CTRL, 0xf, 0xf, false));
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t_min, CTRL, 0xf, 0xf, false);
    t_min = o < t_min ? o : t_min;
}

template <int TIER> struct FitShape {
    static constexpr int kStride = TIER == kFitSmall ? kFitSmallStride : kFitStride;
    static constexpr int kChunks = TIER == kFitSmall ? (kFitSmallCap + 63) / 64 : 4; // 64-element chunks a list can span
};

// the group of seeds s0 .. s0+3 on one wave
template <int TIER> __device__ __forceinline__ void fit_group(const DeviceCtx *__restrict__ c, int s0,
                                                             float (*s_col)[kFitCols][FitShape<TIER>::kStride], float *s_ones,
                                                             double (*s_solver)[52], float4 *s_plane) {
    constexpr int kChunks = FitShape<TIER>::kChunks;
    const int lane = lane_id(), g = lane >> 4, gl = lane & (kFitLanes - 1);
    const int S = c->n_seed;
    // the seed in slot s0 + g: batched launches take the seeds in the order the stage before left (fit_order)
    const int s = TIER == kFitAll ? s0 + g : (s0 + g < S ? fit_order(c)[s0 + g] : -1);
    const bool live = TIER == kFitAll ? s < S : s >= 0;
    stamp(c, 4, s0, 0, lane);
    const FrameParams &fp = frame_params(c);
    const double hr = c->huber;
    const float hr_above = flt_above(hr); // the Huber class tests in fp32 (dsm_math.h)
    GnHeader hd;
    hd.m_in = 0;
    float4 core = make_float4(0, 0, 0, 0);
    // Everything a lane will want from memory is asked for at once, before anything is waited for: the seed's header, and
    // this lane's window row (labels, depths, the rays of the sixteen columns) for the gather below -- whether the seed has
    // a list at all is in the header, but a wave of this kernel lives as long as its round trips take (a third of a
    // wave's life was waiting: for the header, then for the rows, then for one ray per inlier column, each in turn).
    LabelQuad row_lab[4];
    float4 row_dp[4];
    float row_rx[kWin], row_ry = 0.0f;
    int wx0 = 0;
    bool row_in = false;
    // (with them this lane's part in the sums and in the tabled 4x4 inverse, dsm_math.h kInv4: tables in memory too)
    int d2[4] = {0, 0, 0, 0}, oe[7] = {0, 0, 0, 0, 0, 0, 1};
    if (gl < 12)
#pragma unroll
        for (int q = 0; q < 4; q++) d2[q] = kInv4.det2[gl][q];
#pragma unroll
    for (int q = 0; q < 7; q++) oe[q] = kInv4.out[gl][q];
    const int xcol = kFitX[gl], ycol = kFitY[gl], h_row = kFitRow[gl], h_col = kFitColI[gl];
    if (live) {
        hd = c->gn_hdr[s];
        core = c->core[s];
        const float *dep = frame_depth(c, fp);
        const int w = c->w, h = c->h, pitch = c->pitch;
        int gx, gy;
        seed_cell(c, s, gx, gy);
        wx0 = gx * kCell + kCell / 2 - kCell;
        const int y = gy * kCell + kCell / 2 - kCell + gl;
        row_in = (unsigned)y < (unsigned)h;
        const unsigned row = (unsigned)__mul24(y < 0 ? 0 : (y > h - 1 ? h - 1 : y), pitch);
        row_ry = ld_off(c->ray_y, (unsigned)(y < 0 ? 0 : (y > h ? h : y)) << 2);
#pragma unroll
        for (int q = 0; q < 4; q++) { // window quads redirected into the row where they leave it (masked below)
            const int xq = wx0 + 4 * q;
            const unsigned o4 = (row + (unsigned)(xq < 0 ? 0 : (xq > pitch - 4 ? pitch - 4 : xq))) << 2;
            row_lab[q] = label_quad(c->label, o4 >> 2);
            row_dp[q] = ld_vec<float4>(dep, o4);
        }
#pragma unroll
        for (int j = 0; j < kWin; j++) {
            const int x = wx0 + j;
            row_rx[j] = ld_off(c->ray_x, (unsigned)(x < 0 ? 0 : (x > w ? w : x)) << 2);
        }
    }
    const int m = hd.m_in;
    int mg[kFitSeeds];
#pragma unroll
    for (int q = 0; q < kFitSeeds; q++) mg[q] = __builtin_amdgcn_readlane(m, q * kFitLanes);
    int m_max = mg[0];
#pragma unroll
    for (int q = 1; q < kFitSeeds; q++) m_max = mg[q] > m_max ? mg[q] : m_max;
    if (TIER == kFitSmall && m_max > c->fit_small_cap) { // does not fit this tier's columns: queue it for the other
        if (lane == 0) c->worklist[atomicAdd(c->fit_big_count, 1)] = s0 / kFitSeeds;
        return;
    }
    const int m8 = (m_max + 7) & ~7;
    float nx = hd.nx, ny = hd.ny, nz = hd.nz, nb = 0.0f;
    stamp(c, 4, s0, 1, lane);

    if (m_max > 0) {
        // ---- lists into LDS: the columns zeroed up to m8, then the sixteen lanes of a seed gather its centred inlier points,
        // one window row each, in window row-major order (FF.cpp:846-861, 121-126: the points k_seed_stats summed)
        if (lane < 8) s_ones[lane] = 1.0f;
        {
            const int i4 = lane * 4; // m8 <= 232: one 16-byte chunk per lane and column
            if (i4 < m8) {
#pragma unroll
                for (int q = 0; q < kFitSeeds; q++)
#pragma unroll
                    for (int col = 0; col < kFitCols; col++) *reinterpret_cast<float4 *>(&s_col[q][col][i4]) = make_float4(0, 0, 0, 0);
            }
        }
        wave_lds_sync();
        if (m > 0) {
            const int w = c->w;
            const float md = core.w;
            unsigned inl = 0; // this row's inliers, bit j = window column j
#pragma unroll
            for (int j = 0; j < kWin; j++) {
                const float d = comp(row_dp[j >> 2], j & 3);
                const bool ok = row_in && (unsigned)(wx0 + j) < (unsigned)w && comp(row_lab[j >> 2], j & 3) == (unsigned)s && d > flt_below(0.05) &&
                                fabsf(md - d) < hr_above;
                inl |= ok ? 1u << j : 0u;
            }
            // where this row's points start in the seed's list: exclusive prefix of the row counts over the group's 16 lanes
            const int cnt = __popc(inl);
            int pre = cnt;
            pre += __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xf, 0xf, false); // row_shr:1 .. 8: Hillis-Steele within the row of 16
            pre += __builtin_amdgcn_update_dpp(0, pre, 0x112, 0xf, 0xf, false);
            pre += __builtin_amdgcn_update_dpp(0, pre, 0x114, 0xf, 0xf, false);
            pre += __builtin_amdgcn_update_dpp(0, pre, 0x118, 0xf, 0xf, false);
            int pos = pre - cnt;
#pragma unroll
            for (int j = 0; j < kWin; j++) {
                if ((inl >> j) & 1u) { // (column wx0 + j is in [0, w) for an inlier: its ray is the unclamped one)
                    const float d = comp(row_dp[j >> 2], j & 3);
                    s_col[g][0][pos] = row_rx[j] * d - hd.mx;
                    s_col[g][1][pos] = row_ry * d - hd.my;
                    s_col[g][2][pos] = d - hd.mz;
                    pos++;
                }
            }
        }
        double *SA = s_solver[g], *SD = SA + 16, *SO = SA + 28, *SJ = SA + 44, *SU = SA + 48;
        const bool is_j = gl >= 10 && gl < 14;
        const float *xc = xcol == 4 ? s_ones : s_col[g][xcol], *yc = ycol == 4 ? s_ones : s_col[g][ycol];
        const int xs = xcol == 4 ? 0 : 1, ys = ycol == 4 ? 0 : 1;
        wave_lds_sync();
        // this lane's points of every list (element k*64+lane of seed q), for the residuals
        float pq[kFitSeeds][kChunks][3];
#pragma unroll
        for (int q = 0; q < kFitSeeds; q++)
#pragma unroll
            for (int k = 0; k < kChunks; k++) {
                const int i = k * 64 + lane;
#pragma unroll
                for (int col = 0; col < 3; col++) pq[q][k][col] = (k * 64 < mg[q] && i < m8) ? s_col[q][col][i] : 0.0f;
            }
        stamp(c, 4, s0, 2, lane);
        unsigned long long h_masks[4] = {0, 0, 0, 0}; // class masks (this lane's seed) the cached inverse was built from
        bool h_all_core = false;                      // ... and whether they were empty for all four seeds (wave-uniform)
        for (int it = 0; it < 5; it++) {
            if (it == 1) stamp(c, 4, s0, 3, lane);
            // residuals and Huber classes of every seed's list, lane-parallel.  The class masks are wave-uniform values (a
            // ballot each); a seed's lanes take theirs only on a step that has an outlier somewhere (any_out)
            unsigned long long out_mask[kFitSeeds][kChunks];
            bool any_out = false;
            // every seed's plane: each group's first lane leaves its four floats in LDS, every lane reads all sixteen (four
            // 16-byte reads instead of sixteen cross-lane moves: the instructions of this kernel are what it is short of)
            if (gl == 0) s_plane[g] = make_float4(nx, ny, nz, nb);
            wave_lds_sync();
            float pn[kFitSeeds][4];
#pragma unroll
            for (int q = 0; q < kFitSeeds; q++) {
                const float4 v = s_plane[q];
                pn[q][0] = v.x; pn[q][1] = v.y; pn[q][2] = v.z; pn[q][3] = v.w;
            }
            // the residuals of all four lists, 64 elements of a list at a time; the residual column carries r for a core element
            // and the tail sign for an outlier, 0 for a NaN residual (huber_class32 as selects, see fit_ordered_sum) -- formed only
            // where a chunk has an outlier at all (wave-uniform: after the first step practically never)
#pragma unroll
            for (int q = 0; q < kFitSeeds; q++) {
#pragma unroll
                for (int k = 0; k < kChunks; k++) {
                    if (k == 0 || k * 64 < mg[q]) { // wave-uniform
                        const int i = k * 64 + lane;
                        const bool valid = i < mg[q];
                        const float r = pq[q][k][0] * pn[q][0] + pq[q][k][1] * pn[q][1] + pq[q][k][2] * pn[q][2] + pn[q][3];
                        const bool in_core = fabsf(r) < hr_above;
                        const unsigned long long mask = __ballot(valid && !in_core);
                        out_mask[q][k] = mask;
                        if (mask == 0) {
                            if (valid) s_col[q][3][i] = r;
                        } else {
                            const float tail_v = r >= hr_above ? 1.0f : (r <= -hr_above ? -1.0f : 0.0f);
                            if (valid) s_col[q][3][i] = in_core ? r : tail_v;
                            any_out = true;
                        }
                    } else {
                        out_mask[q][k] = 0;
                    }
                }
            }
            unsigned long long noncore[4] = {0, 0, 0, 0};
            if (any_out) {
#pragma unroll
                for (int q = 0; q < kFitSeeds; q++)
#pragma unroll
                    for (int k = 0; k < kChunks; k++)
                        if (g == q) noncore[k] = out_mask[q][k];
            }
            wave_lds_sync();
            // The Hessian sums read nothing but the points and which elements are in the Huber core: while the class
            // masks of all four seeds stay what they were when H was last summed (from the second step on they are
            // normally all-core), H, its damped inverse and the determinant are bit for bit the same, and the
            // inverse still sits in LDS: only J is new.
            bool reuse_inverse = it > 0 && h_all_core && !any_out; // (all empty then and now: nothing to compare)
            if (it > 0 && !h_all_core && any_out) { // (outliers then and now: the same ones?  One side empty and the other not: they differ)
                bool same = true;
#pragma unroll
                for (int k = 0; k < 4; k++) same = same && noncore[k] == h_masks[k];
                reuse_inverse = __ballot(!same) == 0;
            }
            // ... and then only four of a seed's sixteen lanes have a sum to take.  Round 6: all sixteen take a QUARTER of one
            // -- lane gl = component (gl & 3) of J, blocks of four elements gl >> 2, + 4, + 8, ... -- wherever the order of a sum
            // provably does not matter: every term is the reference's own fp32 product widened to double, and while the
            // magnitudes of a sum's terms span less than 2^29 no addition of any order rounds (dsm_math.h, gn_sum_is_exact:
            // 99.8 % of these steps; tests/hostemu.cpp).  A lane carries sum|t| and the smallest non-zero |t| beside its
            // quarter; if every sum of all four seeds passes, J is the ordered J bit for bit -- else the ordered sums run.
            bool have_j = false;
            if (reuse_inverse && !any_out) {
                const int a = gl & 3;
                const float *rc = s_col[g][3], *pc = a == 3 ? s_ones : s_col[g][a];
                const int ps = a == 3 ? 0 : 1;
                double part = 0.0;
                float t_sum = 0.0f;
                uint32_t t_min = 0xffffffffu;
                for (int b = (gl >> 2) * 4; b < m8; b += 16) { // (blocks of four: the quarters differ by at most one block)
                    const float4 xa = *reinterpret_cast<const float4 *>(rc + b), ya = *reinterpret_cast<const float4 *>(pc + b * ps);
                    const float t[4] = {xa.x * ya.x, xa.y * ya.y, xa.z * ya.z, xa.w * ya.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        part += (double)t[q];
                        t_sum += fabsf(t[q]);
                        const uint32_t key = gn_min_key(t[q]);
                        t_min = key < t_min ? key : t_min;
                    }
                }
                // the four quarters of a component sit 4 lanes apart within the seed's row of sixteen: row_ror 8, then 4
                // (every lane ends with the same bits: x + y == y + x)
                quarter_fold<0x128>(part, t_sum, t_min);
                quarter_fold<0x124>(part, t_sum, t_min);
                if (__ballot(!gn_sum_is_exact(t_sum, gn_min_key_value(t_min))) == 0) {
                    if (gl < 4) SJ[gl] = 2.0 * part;
                    have_j = true;
                }
            }
            double acc = 0.0;
            if (!have_j) {
                acc = fit_ordered_sum(xc, yc, xs, ys, m8, noncore, is_j, hr);
                if (gl >= 10 && gl < 14) SJ[gl - 10] = acc;
            }
            if (!reuse_inverse) {
#pragma unroll
                for (int k = 0; k < 4; k++) h_masks[k] = noncore[k];
                h_all_core = !any_out;
                // damped solve, FF.cpp:172-180: one lane per 2x2 determinant, per adjugate entry, per row -- per seed
                if (gl < 10) {
                    // H(3,3) += 2 per core element (FF.cpp:150): an integer, no sum needed
                    const int n_core = m - (__popcll(noncore[0]) + __popcll(noncore[1]) + __popcll(noncore[2]) + __popcll(noncore[3]));
                    const double hv = gl == 9 ? 2.0 * (double)n_core : acc;
                    const double v = h_row == h_col ? hv + 5 : hv; // +5 on the diagonal
                    SA[h_col * 4 + h_row] = v;
                    SA[h_row * 4 + h_col] = v;
                }
                wave_lds_sync();
                if (gl < 12) SD[gl] = SA[d2[0]] * SA[d2[1]] - SA[d2[2]] * SA[d2[3]];
                wave_lds_sync();
                double Dv[12];
#pragma unroll
                for (int t = 0; t < 12; t++) Dv[t] = SD[t];
                const double inv_det = 1.0 / inv4_det(Dv);
                const double sg = (double)oe[6];
                const double t1 = sg * (SA[oe[0]] * SD[oe[1]]), t2 = sg * (SA[oe[2]] * SD[oe[3]]), t3 = sg * (SA[oe[4]] * SD[oe[5]]);
                SO[gl] = ((t1 - t2) + t3) * inv_det;
            }
            wave_lds_sync();
            if (gl < 4) SU[gl] = ((SO[gl] * SJ[0] + SO[4 + gl] * SJ[1]) + SO[8 + gl] * SJ[2]) + SO[12 + gl] * SJ[3];
            wave_lds_sync();
            nx = (float)((double)nx - SU[0]);
            ny = (float)((double)ny - SU[1]);
            nz = (float)((double)nz - SU[2]);
            nb = (float)((double)nb - SU[3]);
            wave_lds_sync();
        }
    }

    stamp(c, 4, s0, 4, lane);
    if (kWaveStamps && c->stamps && lane == 0) c->stamps[((int64_t)4 * c->n_seed + s0) * 8 + 7] = m_max;
    // ---- the fitted plane goes to k_seed_finish (the seed record and the surfel it would create are a few hundred
    // double-typed instructions per seed: there a lane per seed, here they would run with 4 of 64 lanes)
    if (live && gl == 0 && m > 0) c->plane[s] = make_float4(nx, ny, nz, nb);
    if (g == 0) stamp(c, 4, s0, 5, lane);
}

// The seed record (FF.cpp:872-914: plane to normal / position / view angle) and the per-seed part of initialize_surfels
// (FF.cpp:315-361, up to the `fused` test that k_frame_tail applies), one thread per seed.
template <bool BATCH> __global__ __launch_bounds__(256) void k_seed_finish(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const int s = blk.x * 256 + threadIdx.x;
    if (s >= c->n_seed) return;
    const FrameParams &fp = frame_params(c);
    const Intrinsics K = c->k;
    const GnHeader hd = c->gn_hdr[s];
    const float4 core = c->core[s];
    dsm_seed out;
    out.x = core.x; out.y = core.y;
    out.size = 0; out.norm_x = out.norm_y = out.norm_z = 0;
    out.posi_x = out.posi_y = out.posi_z = 0;
    out.view_cos = 0;
    out.mean_depth = core.w;
    out.mean_intensity = core.z;
    out.fused = 0;
    out.stable = (uint8_t)(c->tmin[s] == kIntMax ? 1 : 0);
    out.pad_[0] = out.pad_[1] = 0;
    out.min_eigen_value = out.max_eigen_value = 0;
    if (hd.m_in > 0) {
        const float4 pl = c->plane[s];
        float nx = pl.x, ny = pl.y, nz = pl.z, nb = pl.w;
        plane_finish(nx, ny, nz, nb, hd.mx, hd.my, hd.mz);
        const SeedGeom sg = seed_geometry(K, core.x, core.y, core.w, nx, ny, nz, nb);
        out.norm_x = sg.nx; out.norm_y = sg.ny; out.norm_z = sg.nz;
        out.posi_x = sg.px; out.posi_y = sg.py; out.posi_z = sg.pz;
        out.mean_depth = sg.mean_depth;
        out.view_cos = sg.view_cos;
        out.size = sqrtf(hd.far2);
    }
    c->seeds[s] = out;
    SeedView sd;
    sd.size = out.size; sd.nx = out.norm_x; sd.ny = out.norm_y; sd.nz = out.norm_z;
    sd.px = out.posi_x; sd.py = out.posi_y; sd.pz = out.posi_z;
    sd.view_cos = out.view_cos; sd.mean_depth = out.mean_depth; sd.mean_intensity = out.mean_intensity;
    const bool ok = seed_spawns(sd, false);
    if (ok) {
        const Surfel e = spawn_surfel(K, fp.ref_idx, fp.pose, sd);
        dsm_surfel o;
        o.px = e.px; o.py = e.py; o.pz = e.pz; o.nx = e.nx; o.ny = e.ny; o.nz = e.nz;
        o.size = e.size; o.color = e.color; o.weight = e.weight;
        o.update_times = e.update_times; o.last_update = e.last_update;
        c->spawn_rec[s] = o;
    }
    c->spawn_ok[s] = ok ? 1 : 0;
    c->fused_flag[s] = 0;
    c->seed_weight[s] = depth_weight(out.mean_depth); // FF.cpp:274: what a surfel fusing into this seed weighs it with
}

template <bool BATCH, int TIER> __global__ __launch_bounds__(64) void k_seed_fit(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    __shared__ __attribute__((aligned(16))) float s_col[kFitSeeds][kFitCols][FitShape<TIER>::kStride];
    __shared__ __attribute__((aligned(16))) float s_ones[8];
    __shared__ double s_solver[kFitSeeds][52]; // per seed: [16] damped H | [12] 2x2 dets | [16] inverse | [4] J | [4] update
    __shared__ float4 s_plane[kFitSeeds];      // per seed: the plane of the step being taken
    if (TIER == kFitLarge) {
        const int n_big = c->fit_big_count[0];
        for (int e = blk.x; e < n_big; e += kFitLargeBlocks) {
            fit_group<TIER>(c, c->worklist[e] * kFitSeeds, s_col, s_ones, s_solver, s_plane);
            wave_lds_sync();
        }
    } else {
        const int n_groups = (c->n_seed + kFitSeeds - 1) / kFitSeeds;
        fit_group<TIER>(c, (n_groups - 1 - blk.x) * kFitSeeds, s_col, s_ones, s_solver, s_plane); // bottom rows (long lists) first, see seed_of_block
    }
}


} // namespace dsm
