// dsm_k_superpixel.h -- the superpixel sweeps: k_init_seeds, k_assign, k_resolve, k_apply_labels, k_update_seeds (lane and
// wave forms, k_update_seeds_rest), k_commit_seeds.  FF.cpp:364-629.  Included by dsm_kernels.hip.
#pragma once
#include "dsm_k_common.h"

namespace dsm {

// ------------------------------------------------------------------------------ init seeds
// FF.cpp:577-629.  A seed whose centre pixel has no depth takes the first depth > 0.01 of its clipped 16x16 window
// in row-major order (FF.cpp:600-626) -- whole image regions (sky) need that at once.  Sixteen lanes per seed, lane r
// holding window row r as four 16-byte loads issued together with the centre pixel (speculatively: whether the scan
// is needed is only known once the centre has arrived, and a second dependent round trip costs more than the 1 KB
// per seed read from L2); the first hit is the lowest lane with one: one ballot per wave.
constexpr int kInitLanes = 16, kInitSeedsPerBlock = 256 / kInitLanes;
template <bool BATCH> __global__ __launch_bounds__(256) void k_init_seeds(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const int tid = threadIdx.x, lane = lane_id();
    const int r = tid & (kInitLanes - 1);
    const int s = blk.x * kInitSeedsPerBlock + tid / kInitLanes;
    if (blk.x == 0 && tid < kSweeps * kWorkers) c->first_empty[tid] = kIntMax;
    if (blk.x == 0 && tid == 0) c->work_count[0] = c->fit_big_count[0] = 0;
    if (blk.x == 0 && tid < 2 * kSweeps) c->rest_count[tid] = 0;
    // first kernel of the frame: resolve the params ring once and publish the result (FrameCur)
    const FrameParams &fp = c->params[(unsigned)(c->cursor[0] * c->cursor_mul + c->cursor_add) % (unsigned)c->n_params];
    const uint8_t *img = c->img_base + (int64_t)fp.slot * c->slot_elems;
    const float *dep = c->depth_base + (int64_t)fp.slot * c->slot_elems;
    if (blk.x == 0 && tid < 64) {
        FrameCur *wc = c->cur;
        const int t = tid;
        if (t < 16) wc->p.pose[t] = fp.pose[t];
        else if (t < 32) wc->p.inv[t - 16] = fp.inv[t - 16];
        else if (t == 32) { wc->p.ref_idx = fp.ref_idx; wc->p.slot = fp.slot; }
        else if (t == 33) wc->img = img;
        else if (t == 34) wc->dep = dep;
    }
    const int w = c->w, h = c->h, pitch = c->pitch;
    const bool live = s < c->n_seed;
    const int sc = live ? s : 0;
    const int gx = sc % c->gw, gy = sc / c->gw;
    int ix = gx * kCell + kCell / 2, iy = gy * kCell + kCell / 2;
    if (ix > w - 1) ix = w - 1;
    if (iy > h - 1) iy = h - 1;
    const int wx0 = gx * kCell + kCell / 2 - kCell, wy0 = gy * kCell + kCell / 2 - kCell;
    const int x_lo = wx0 < 0 ? 0 : wx0, x_hi = wx0 + 2 * kCell > w - 1 ? w - 1 : wx0 + 2 * kCell;
    const int y_lo = wy0 < 0 ? 0 : wy0, y_hi = wy0 + 2 * kCell > h - 1 ? h - 1 : wy0 + 2 * kCell;
    const int y = wy0 + r;
    const bool row_in = y >= y_lo && y < y_hi;
    float md = dep[iy * pitch + ix];
    const float mi = (float)img[iy * pitch + ix];
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = wx0 + 4 * q; // multiple of 4: 16-byte aligned, and never straddles x = 0
        v[q] = (row_in && x >= 0) ? *reinterpret_cast<const float4 *>(dep + y * pitch + x) : make_float4(0, 0, 0, 0);
    }
    // first hit of this row
    bool hit = false;
    float first = 0.0f;
#pragma unroll
    for (int q = 3; q >= 0; q--) {
        const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int t = 3; t >= 0; t--) {
            const int x = wx0 + 4 * q + t;
            if (row_in && x >= x_lo && x < x_hi && e[t] > flt_below(0.01)) { hit = true; first = e[t]; }
        }
    }
    // first row with a hit among the 16 lanes of this seed
    const unsigned long long m = __ballot(hit);
    const unsigned rows = (unsigned)(m >> (lane & ~(kInitLanes - 1))) & 0xffffu;
    const int src = (lane & ~(kInitLanes - 1)) + (rows ? __ffs((int)rows) - 1 : 0);
    const float scanned = __shfl(first, src);
    if (md < flt_above(0.01) && rows) md = scanned; // (double)md < 0.01
    if (!live || r != 0) return;
    c->core[s] = make_float4((float)ix, (float)iy, mi, md);
    c->inv_depth[s] = 1.0 / (double)md;
    c->tmin[s] = -1; // fused = stable = false
}

// The same with ONE LANE PER SEED, for launches batched over many handles: sixteen lanes per seed are 440 workgroups per
// handle, each a chain of three dependent trips to memory (cursor -> params -> pixels) that ends in one 16-byte store per
// sixteen lanes -- 56 000 workgroups per launch of 128 handles, 153 us of wave turnover.  Here a lane reads its seed's
// centre pixel, and only a wave that holds a seed without depth there walks windows: every such lane its own, row by row
// from the first until it has found the lowest row and column with a depth (FF.cpp:600-626).
template <bool BATCH> __global__ __launch_bounds__(256) void k_init_seeds_lanes(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const int tid = threadIdx.x;
    const int s = blk.x * 256 + tid;
    if (blk.x == 0 && tid < kSweeps * kWorkers) c->first_empty[tid] = kIntMax;
    if (blk.x == 0 && tid == 0) c->work_count[0] = c->fit_big_count[0] = 0;
    if (blk.x == 0 && tid < 2 * kSweeps) c->rest_count[tid] = 0;
    // first kernel of the frame: resolve the params ring once and publish the result (FrameCur)
    const FrameParams &fp = c->params[(unsigned)(c->cursor[0] * c->cursor_mul + c->cursor_add) % (unsigned)c->n_params];
    const uint8_t *img = c->img_base + (int64_t)fp.slot * c->slot_elems;
    const float *dep = c->depth_base + (int64_t)fp.slot * c->slot_elems;
    if (blk.x == 0 && tid < 64) {
        FrameCur *wc = c->cur;
        const int t = tid;
        if (t < 16) wc->p.pose[t] = fp.pose[t];
        else if (t < 32) wc->p.inv[t - 16] = fp.inv[t - 16];
        else if (t == 32) { wc->p.ref_idx = fp.ref_idx; wc->p.slot = fp.slot; }
        else if (t == 33) wc->img = img;
        else if (t == 34) wc->dep = dep;
    }
    const int w = c->w, h = c->h, pitch = c->pitch;
    const bool live = s < c->n_seed;
    const int sc = live ? s : 0;
    int gx, gy;
    seed_cell(c, sc, gx, gy);
    int ix = gx * kCell + kCell / 2, iy = gy * kCell + kCell / 2;
    if (ix > w - 1) ix = w - 1;
    if (iy > h - 1) iy = h - 1;
    float md = dep[iy * pitch + ix];
    const float mi = (float)img[iy * pitch + ix];
    const bool need = live && md < flt_above(0.01); // (double)md < 0.01
    if (__ballot(need) != 0) {
        const int wx0 = gx * kCell + kCell / 2 - kCell, wy0 = gy * kCell + kCell / 2 - kCell;
        const int x_lo = wx0 < 0 ? 0 : wx0, x_hi = wx0 + 2 * kCell > w - 1 ? w - 1 : wx0 + 2 * kCell;
        const int y_lo = wy0 < 0 ? 0 : wy0, y_hi = wy0 + 2 * kCell > h - 1 ? h - 1 : wy0 + 2 * kCell;
        // rows from the first on, two at a time, until every lane that looks has found its pixel: the first hit in row-major
        // order (a row is scanned from its last column down, so that its first hit remains).  Most seeds without a depth
        // at their centre find one in the first rows; only a wave that holds an empty window (sky) walks all sixteen.
        bool pending = need;
#pragma unroll 1
        for (int r0 = 0; r0 < 2 * kCell && __ballot(pending) != 0; r0 += 2) {
            float4 v[2][4];
            bool row_in[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int y = wy0 + r0 + k;
                row_in[k] = pending && y >= y_lo && y < y_hi;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int x = wx0 + 4 * q; // multiple of 4: 16-byte aligned, and never straddles x = 0
                    v[k][q] = (row_in[k] && x >= 0) ? *reinterpret_cast<const float4 *>(dep + y * pitch + x) : make_float4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
                bool hit = false;
                float first = 0.0f;
#pragma unroll
                for (int q = 3; q >= 0; q--) {
                    const float e[4] = {v[k][q].x, v[k][q].y, v[k][q].z, v[k][q].w};
#pragma unroll
                    for (int t = 3; t >= 0; t--) {
                        const int x = wx0 + 4 * q + t;
                        if (row_in[k] && x >= x_lo && x < x_hi && e[t] > flt_below(0.01)) { hit = true; first = e[t]; }
                    }
                }
                if (pending && hit) { md = first; pending = false; }
            }
        }
    }
    if (!live) return;
    c->core[s] = make_float4((float)ix, (float)iy, mi, md);
    c->inv_depth[s] = 1.0 / (double)md;
    c->tmin[s] = -1; // fused = stable = false
}

// ------------------------------------------------------------------------------ assign
// One thread per column of FOUR pixels (a 4 x 4 quadrant of a cell shares its <= 2 x 2 candidate seeds: they are fetched
// once per thread), a 64x16-pixel tile per block; the <=10x4 seeds a tile can pick from are staged in LDS.  FIRST sweep:
// every pixel is evaluated (all labels 0, seed 0 unstable) so the pick is the label.  Later sweeps: the pick goes to
// `cand`, and the sequential skip rule is resolved through tmin (see k_resolve).
constexpr int kTileW = 64, kTileCellsX = kTileW / kCell + 2;
template <int COLS> struct AssignTile { // COLS pixels per thread: 4 in launches batched over many handles, 1 where latency counts
    static constexpr int kH = 4 * COLS, kCellsY = (kH + kCell - 1) / kCell + 2;
};

// The reference scans pixels in row-major order; a pixel is skipped iff its current seed is still
// `stable` when the scan reaches it, and every evaluated pixel clears `stable` of the seed it
// picks.  With T[s] = first pixel key at which s is cleared this reads
//     evaluated(p)  <=>  T[label(p)] < p ,      T[s] = min { p : evaluated(p), pick(p) = s } ,
// whose least fixed point from above (T = -1 for unstable seeds, +inf for stable ones) is reached
// by repeated atomicMin.  Every pixel whose seed was unstable applies its own atomicMin directly;
// only pixels whose old and new seeds were both stable (a short list: borders between two seeds
// that stopped moving) can still change the picture; k_resolve iterates that list to the fixed point.
__device__ void resolve_worklist(const DeviceCtx *c, const label_t *label_in) {
    const int n = c->work_count[0];
    if (n == 0) return;
    for (;;) {
        int changed = 0;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int p = c->worklist[i];
            const int l = label_in[p], pk = c->cand[p]; // (both seeds of a listed pixel exist)
            if (load_coherent(&c->tmin[l]) < p && load_coherent(&c->tmin[pk]) > p) {
                atomicMin(&c->tmin[pk], p);
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
}

template <bool FIRST, bool BATCH, int COLS> __global__ __launch_bounds__(256) void k_assign(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    constexpr int kColumn = COLS, kTileH = AssignTile<COLS>::kH, kTileCellsY = AssignTile<COLS>::kCellsY;
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    // The seeds a tile can pick from, cell (cx0 + i, cy0 + j) at index j * kTileCellsX + i.  A cell OUTSIDE the grid holds a
    // copy of the nearest one inside: the filtered pick reads its four candidates at fixed offsets from its first one,
    // wherever the tile lies (such a candidate is out of play, dsm_math.h pick_col); the typed pick asks for cells inside only.
    constexpr int kCells = kTileCellsX * kTileCellsY;
    __shared__ float4 s_core[kCells];
    __shared__ double s_inv[kCells];
    // ... and for the filtered pick as one array per field (a pair of candidates = one ds_read2_b32, straight into the register
    // pair of a packed operation): x / 4, y / 4, mean intensity, 20 / mean depth rounded to float (0 without one), mean depth
    __shared__ float s_f[5][kCells];
    // the tile's pixels whose pick the fp32 filter leaves open (tile row << 6 | tile column), for the dense pass below
    __shared__ unsigned short s_open[kTileW * kTileH];
    __shared__ int s_n_open;
    const FrameParams &fp = frame_params(c);
    const uint8_t *img = frame_image(c, fp);
    const float *dep = frame_depth(c, fp);
    const label_t *label_in = c->label; // the previous sweep's image (sweep >= 1)
    const int w = c->w, h = c->h, pitch = c->pitch, gw = c->gw, gh = c->gh;
    const int bx = blk.x * kTileW, by = blk.y * kTileH;
    const int cx0 = bx / kCell - 1, cy0 = by / kCell - 1;
    const int tid = threadIdx.x;
    if (tid == 0) s_n_open = 0;
    if (tid < kCells) {
        int gx = cx0 + tid % kTileCellsX, gy = cy0 + tid / kTileCellsX;
        gx = gx < 0 ? 0 : (gx > gw - 1 ? gw - 1 : gx);
        gy = gy < 0 ? 0 : (gy > gh - 1 ? gh - 1 : gy);
        const float4 v = c->core[gy * gw + gx];
        const double inv = c->inv_depth[gy * gw + gx];
        s_core[tid] = v;
        s_inv[tid] = inv;
        s_f[0][tid] = v.x * 0.25f;
        s_f[1][tid] = v.y * 0.25f;
        s_f[2][tid] = v.z;
        s_f[3][tid] = seed_s20(v.w, inv);
        s_f[4][tid] = v.w;
    }
    __syncthreads();
    // what becomes of a pixel once its pick is known (FF.cpp:442-451 and the stable-skip bookkeeping, see resolve_worklist).
    // tl = tmin of the pixel's old seed (-1 never changes; >= 0 only moves among values >= 0: it may be read at any time),
    // tp = tmin of the picked seed, read coherently after the pick was known.
    auto settle = [&](int p, int l, int pick, int tl, int tp) {
        if (pick < 0) { // every candidate cost >= the reference's 1e6 sentinel: it would index seeds[-1]
            atomicOr(c->status, kStatusBadPick);
            if (FIRST) label_put(c->label, (unsigned)p, 0); else label_put(c->cand, (unsigned)p, l);
        } else if (FIRST) {
            label_put(c->label, (unsigned)p, pick);
        } else {
            label_put(c->cand, (unsigned)p, pick);
            if (tl == -1) {
                // the old seed was unstable at sweep start: this pixel is evaluated whatever happens
                // elsewhere, so its pick loses `stable` no later than at p
                if (tp > p) atomicMin(&c->tmin[pick], p);
            } else if (pick != l && tp != -1) {
                // old and new seed both stable at sweep start: whether this pixel is evaluated depends
                // on the scan order -- resolved below
                const int slot = atomicAdd(c->work_count, 1);
                c->worklist[slot] = p;
            }
        }
    };
    auto tmin_of = [&](int s) { // tmin[s] by a 32-bit offset from the uniform base, coherently (other workgroups lower it)
        return __hip_atomic_load(reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(c->tmin) + ((unsigned)s << 2)), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
    };
    const int x = bx + (tid & (kTileW - 1)), y0 = by + (tid / kTileW) * kColumn; // y0 is a multiple of 4: one quadrant row
    if (x < w && y0 < h) { // (no early return: every thread meets the barrier below)
        // the column's pixels, one round trip; the stable-skip state of their old seeds right behind them
        float pix_i[kColumn], pix_d[kColumn];
        int lab[kColumn], tl[kColumn], pick[kColumn]; // (lab: the label plane's 16 bits as they are -- kNoLabel equals no pick)
        bool live[kColumn], sure[kColumn];
        const unsigned p0 = (unsigned)(__mul24(y0, pitch) + x);
#pragma unroll
        for (int r = 0; r < kColumn; r++) {
            const unsigned p = y0 + r < h ? p0 + (unsigned)(r * pitch) : p0, p4 = p << 2; // byte offsets into the 4-byte planes, see ld_off
            pix_i[r] = (float)ld_off(img, p);
            pix_d[r] = ld_off(dep, p4);
            lab[r] = FIRST ? 0 : (int)ld_off(label_in, p << 1);
        }
        if (!FIRST) {
            const unsigned s_last = (unsigned)c->n_seed - 1u; // (a pixel without a label is never settled: any seed's entry will do)
#pragma unroll
            for (int r = 0; r < kColumn; r++) tl[r] = ld_off(c->tmin, ((unsigned)lab[r] < s_last ? (unsigned)lab[r] : s_last) << 2);
        }
        // (candidate 0's cell is at most one cell left of / above the tile: inside the staged halo, as are the other three)
        const int li0 = __mul24(((y0 >> 3) - ((y0 & 7) < 4 ? 1 : 0)) - cy0, kTileCellsX) + (((x >> 3) - ((x & 7) < 4 ? 1 : 0)) - cx0);
        const PickCol quad = pick_col(x, y0, gw, gh, [&](int k, int, int, float &sx4, float &sy4, float &si, float &sd, float &s20) {
            const int li = li0 + (k >> 1) + (k & 1) * kTileCellsX;
            sx4 = s_f[0][li]; sy4 = s_f[1][li]; si = s_f[2][li]; s20 = s_f[3][li]; sd = s_f[4][li];
        });
        // ---- the four picks: the argmin from fp32 costs with error bounds where that is decisive (dsm_math.h, pick_seed_fast),
        // straight-line for the whole column (rows beyond the image repeat row 0's pixel and are dropped below)
#pragma unroll
        for (int r = 0; r < kColumn; r++) {
            const int y = y0 + r;
            live[r] = y < h && has_candidate_cell(x, y, gw, gh);
            const FastPick fpk = pick_seed_fast(quad, y, pix_i[r], pix_d[r], gw);
            pick[r] = fpk.seed;
            sure[r] = fpk.sure;
            // ragged border beyond every cell's reach: label -1, once per frame (no later stage changes these pixels:
            // every seed window ends before them, and k_apply_labels keeps a -1)
            if (FIRST && y < h && !live[r]) label_put(c->label, p0 + (unsigned)(r * pitch), -1);
        }
        // ---- a pixel the filter leaves open goes onto the tile's list: the reference's typed arithmetic is several hundred
        // instructions, and run here it would run for the whole wave whenever ONE of its 64 pixels is open -- 16 % of the
        // wave-rows on smooth synthetic depth, and most of them on the reference's own kind of input (kitti_publisher: depth
        // = bf / quantised disparity, a few grey levels), where a tenth of the first sweep's pixels sit exactly between two
        // seeds of equal intensity whose inverse depths lie on the disparity lattice.
#pragma unroll
        for (int r = 0; r < kColumn; r++) {
            const bool open = live[r] && !sure[r];
            const unsigned long long m = __ballot(open);
            if (m) {
                const int rk = rank_below(m);
                int base = 0;
                if (open && rk == 0) base = atomicAdd(&s_n_open, __popcll(m));
                base = __builtin_amdgcn_readlane(base, __ffsll((long long)m) - 1);
                if (open) s_open[base + rk] = (unsigned short)((((tid / kTileW) * kColumn + r) << 6) | (tid & (kTileW - 1)));
            }
            live[r] = live[r] && !open;
        }
        // ---- the settled ones: the picked seeds' stable-skip state for the whole column in one round trip, then the stores
        int tp[kColumn];
        if (!FIRST) {
#pragma unroll
            for (int r = 0; r < kColumn; r++) tp[r] = tmin_of(live[r] ? pick[r] : 0); // (a settled pick is a seed index)
        }
#pragma unroll
        for (int r = 0; r < kColumn; r++)
            if (live[r]) settle((int)p0 + r * pitch, lab[r], pick[r], FIRST ? 0 : tl[r], FIRST ? 0 : tp[r]);
    }
    __syncthreads();
    // ---- the open pixels of the tile, packed: 64 to a wave whatever rows and columns they came from
    const int n_open = s_n_open;
    for (int i = tid; i < n_open; i += 256) {
        const unsigned e = s_open[i];
        const int ox = bx + (int)(e & 63u), oy = by + (int)(e >> 6);
        const unsigned p = (unsigned)(__mul24(oy, pitch) + ox);
        const float pi = (float)ld_off(img, p), pd = ld_off(dep, p << 2);
        const int l = FIRST ? 0 : label_at(label_in, p);
        const int pick = pick_seed(ox, oy, pi, pd, gw, gh, [&](int gx, int gy, float &sx, float &sy, float &si, bool &has_d, double &inv_d) {
            const int li = __mul24(gy - cy0, kTileCellsX) + (gx - cx0);
            const float4 v = s_core[li];
            sx = v.x; sy = v.y; si = v.z;
            has_d = v.w > 0;
            inv_d = s_inv[li];
        });
        settle((int)p, l, pick, FIRST ? 0 : ld_off(c->tmin, (unsigned)l << 2), FIRST || pick < 0 ? 0 : tmin_of(pick));
    }
}

// One workgroup iterates the worklist to the fixed point.  (Folding this into k_assign behind a
// "last block done" ticket costs a device-scope release per workgroup -- an L2 write-back on this
// multi-XCD part -- and was 10x slower than the extra launch.)
template <bool BATCH> __global__ __launch_bounds__(256) void k_resolve(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    resolve_worklist(c, c->label);
}

// Ordered sum of one Huber-Newton pass (FF.cpp:536-549): element i adds lt[i] = 2*r if its residual is
// in the Huber core, else a = (float)((double)a +- hr).  tail/pos are wave-uniform bit masks per
// 64 elements; blocks without outliers take the plain path.
__device__ __forceinline__ float huber_ordered_sum(const float *lt, int nd, const unsigned long long tail[4],
                                                   const unsigned long long pos[4], double hr) {
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int lim = nd - k * 64 < 64 ? nd - k * 64 : 64;
        if (lim <= 0) break;
        for (int j = 0; j < lim; j += kBlk) {
            const Blk16 v = load_blk(lt + k * 64 + j);
            const unsigned t16 = (unsigned)(tail[k] >> j) & 0xffffu, p16 = (unsigned)(pos[k] >> j) & 0xffffu;
            if (t16 == 0) {
#pragma unroll
                for (int q = 0; q < kBlk; q++) a += v.e[q];
            } else {
#pragma unroll
                for (int q = 0; q < kBlk; q++) {
                    const float a_core = a + v.e[q];
                    const float a_tail = (float)((double)a + (((p16 >> q) & 1u) ? hr : -1 * hr));
                    a = ((t16 >> q) & 1u) ? a_tail : a_core;
                }
            }
        }
    }
    return a;
}

// Huber-Newton passes it0 .. 4 of one seed's robust mean depth by one whole wave (FF.cpp:530-556), starting from md.
// dl[0..nd) = the member depths in order, dl and lt padded with +0.0f to a multiple of kBlk.  The loop-carried part of a
// pass is only the ordered fp32 sum of the per-element terms; residuals and their classification are lane-parallel.
__device__ __forceinline__ float huber_passes_wave(const float *dl, float *lt, int nd, float md, int it0, double hr, int lane) {
    const float hr_above = flt_above(hr); // the Huber class tests in fp32 (dsm_math.h)
    float dk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) dk[k] = (k * 64 + lane < nd) ? dl[k * 64 + lane] : 0.0f;
    const int nk = (nd + 63) >> 6;
    for (int it = it0; it < 5; it++) {
        unsigned long long tail[4] = {0, 0, 0, 0}, pos[4] = {0, 0, 0, 0};
        int n_core = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= nk) break;
            const int idx = k * 64 + lane;
            const bool valid = idx < nd;
            const float r = md - dk[k];
            const bool core = valid && fabsf(r) < hr_above; // (double)r < hr && (double)r > -hr
            if (valid) lt[idx] = 2 * r;
            tail[k] = __ballot(valid && !core);
            pos[k] = __ballot(valid && !core && r > 0);
            n_core += __popcll(__ballot(core));
        }
        wave_lds_sync();
        const float a = huber_ordered_sum(lt, nd, tail, pos, hr);
        const float b = (float)(2 * n_core); // the reference adds 2.0f per core element: exact
        const float delta = huber_newton_step(a, b);
        md = md + delta;
        wave_lds_sync();
        if (fabsf(delta) < flt_above(0.01)) break; // (double)delta < 0.01 && (double)delta > -0.01
    }
    return md;
}

// ------------------------------------------------------------------------------ update seeds
// One wave per seed.  Lanes cover the 16x16 window (4 pixels each, row-major across k*64+lane).
// Counts and coordinate/intensity sums are integers (exact in the reference's fp32 accumulators,
// any order); the depth sum and the Huber-Newton passes are fp32 sums in window row-major order,
// so member depths are compacted in order into LDS and summed sequentially.
constexpr int kWin = 2 * kCell; // 16

// The label image of a sweep >= 1 is  new(p) = T[old(p)] < p ? pick(p) : old(p)  (see k_assign): k_apply_labels forms it,
// once per pixel and in place, before the seeds are updated.
// Second half of update_seeds for one seed (one wave): the sums of its members are in the lanes' registers, the
// member depths > 0.1 in window row-major order in dl[0..nd).
__device__ __forceinline__ void update_seed_finish(const DeviceCtx *__restrict__ c, int sweep, int s, int lane, int wx0, int wy0,
                                                   const float4 old, float *dl, float *lt, int cnt, int sdx, int sdy, int si, int nd) {
    stamp(c, sweep, s, 2, lane);
    // integer sums (exact in the reference's fp32 accumulators), two per wave reduction: the member count (<= 256) above
    // the intensity sum (<= 256 * 255 < 2^16), and the window offsets 12 bits each, shifted back by cnt * window origin
    const int cnt_si = wave_sum(si | (cnt << 16));
    cnt = cnt_si >> 16;
    si = cnt_si & 0xffff;
    if (cnt == 0) { // FF.cpp:516-517: the worker returns, abandoning the rest of its chunk
        if (lane == 0) atomicMin(&c->first_empty[sweep * kWorkers + chunk_of(c->n_seed, s)], s);
        return;
    }
    const int packed = wave_sum(sdx | (sdy << 16));
    const int sx = (packed & 0xffff) + cnt * wx0, sy = (packed >> 16) + cnt * wy0;
    wave_lds_sync();
    const float fn = (float)cnt;
    const float mi = (float)si / fn, mx = (float)sx / fn, my = (float)sy / fn;
    const float moved = fabsf(old.z - mi) + fabsf(old.x - mx) + fabsf(old.y - my);
    const int stable = moved < flt_above(0.2) ? 1 : 0; // (double)moved < 0.2, in fp32 (dsm_math.h, flt_above)
    stamp(c, sweep, s, 3, lane);
    float md = 0.0f;
    // The kernel ends with its slowest wave, and that is a wave with a long list (its ordered sums are serial chains
    // of nd adds, up to six of them): let it issue ahead of the short ones sharing its SIMD.
    wave_priority(nd);
    if (nd > 0) {
        pad_column(dl, nd, lane);
        pad_column(lt, nd, lane); // pad slots stay +0.0f: the passes only write valid slots
        wave_lds_sync();
        md = ordered_sum(dl, nd) / (float)nd;
        stamp(c, sweep, s, 4, lane);
        if (!mean_depth_is_settled(md)) md = huber_passes_wave(dl, lt, nd, md, 0, c->huber, lane);
    }
    stamp(c, sweep, s, 5, lane);
    if (kWaveStamps && c->stamps && lane == 0) c->stamps[((int64_t)sweep * c->n_seed + s) * 8 + 7] = nd;
    if (lane == 0) {
        c->core_stage[s] = make_float4(mx, my, mi, md);
        c->stable_stage[s] = stable;
    }
}

// update_seeds for ONE seed by one whole wave (lanes cover the 16x16 window, 4 pixels each): the form every seed took
// until round 3.  Today it serves launches for one handle or a few, and the seeds whose depth list outgrows the
// lane-per-seed kernel's longest LDS rows (below).  s is wave-uniform; dl / lt are two lists of 256 floats in LDS owned
// by this wave.
__device__ __forceinline__ void update_seed_wave(const DeviceCtx *__restrict__ c, int sweep, int s, float *dl, float *lt) {
    const int lane = lane_id();
    stamp(c, sweep, s, 0, lane);
    const FrameParams &fp = frame_params(c);
    const uint8_t *img = frame_image(c, fp);
    const float *dep = frame_depth(c, fp);
    const label_t *lbl = c->label;
    const int w = c->w, h = c->h, pitch = c->pitch;
    int gx, gy;
    seed_cell(c, s, gx, gy);
    const int wx0 = gx * kCell + kCell / 2 - kCell, wy0 = gy * kCell + kCell / 2 - kCell;
    const int t_self = c->tmin[s];
    if (t_self == kIntMax) return; // stable: FF.cpp:479-480
    const float4 old = c->core[s]; // needed only after the sums: issued with the window loads, not behind them
    stamp(c, sweep, s, 1, lane);
    int cnt = 0, sdx = 0, sdy = 0, si = 0, nd = 0;
    int lab[4], pi[4];
    float pd[4];
    bool pimg[4];
    // pixel key of this lane's first window pixel; the other three are 4, 8, 12 rows further down (keys are
    // non-negative wherever they are used: a pixel outside the image reads pixel 0 and is masked out)
    const int x = wx0 + (lane & (kWin - 1)), y0 = wy0 + (lane >> 4);
    const bool x_in = x >= 0 && x < w;
    const int key0 = __mul24(y0, pitch) + x, row4 = 4 * pitch;
#pragma unroll
    for (int k = 0; k < 4; k++) { // independent loads, one round trip
        const int y = y0 + 4 * k;
        pimg[k] = x_in && y >= 0 && y < h;
        const int pk = pimg[k] ? key0 + k * row4 : 0;
        const unsigned o4 = (unsigned)pk << 2;
        lab[k] = (int)ld_off(lbl, (unsigned)pk << 1); // (16 bits: kNoLabel equals no seed)
        pd[k] = ld_off(dep, o4);
        pi[k] = (int)ld_off(img, (unsigned)pk);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idx = k * 64 + lane;
        const int y = y0 + 4 * k;
        // statistics window clipped to [0, w-1) x [0, h-1): the last row and column never contribute
        const bool mem = pimg[k] && x < w - 1 && y < h - 1 && lab[k] == s;
        const float d = mem ? pd[k] : 0.0f;
        if (mem) {
            cnt += 1; sdx += idx & (kWin - 1); sdy += idx >> 4; si += pi[k];
        }
        const bool dv = mem && d > flt_below(0.1); // FF.cpp:508, (double)d > 0.1
        const unsigned long long m = __ballot(dv);
        if (dv) dl[nd + rank_below(m)] = d;
        nd += __popcll(m);
    }
    update_seed_finish(c, sweep, s, lane, wx0, wy0, old, dl, lt, cnt, sdx, sdy, si, nd);
}

// ---- the label image of a sweep >= 1, one thread per eight pixels of a row (16 bytes of each plane):  new(p) = T[old(p)] < p ? pick(p) : old(p)
// with T = tmin after k_resolve (see k_assign), IN PLACE: a pixel's new label needs nothing but its own old one, and most
// pixels keep theirs -- only quads in which a label changes are stored.  Until round 4 every seed's window walk formed
// the new labels on the fly for the 256 pixels of its window -- every pixel four times over, each time behind a gather of
// tmin[old label] by 64 lanes that hold 64 different seeds -- and the registers of that (two more row planes, the
// gathered tmin) held the lane-per-seed kernel to one wave per SIMD.  Here a pixel is resolved once, and neighbouring
// pixels mostly share their old label: a wave's gather touches a handful of lines.  Pixels beyond every cell's reach keep
// their -1 (no seed, no tmin).
// A thread takes the eight pixels of kApplyRows consecutive rows; tmin is fetched only for a pixel whose pick differs from its
// label (the others keep theirs whatever tmin says: from the second sweep on that is nearly all of them).
constexpr int kApplyRows = 2;
template <bool BATCH> __global__ __launch_bounds__(256) void k_apply_labels(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const int pitch = c->pitch;
    const int xq = blk.x * 64 + (threadIdx.x & 63), y0 = (blk.y * 4 + (threadIdx.x >> 6)) * kApplyRows;
    if (8 * xq >= pitch || y0 >= c->h) return;
    uint4 lab[kApplyRows], cd[kApplyRows];
#pragma unroll
    for (int r = 0; r < kApplyRows; r++) {
        const int key0 = __mul24(min(y0 + r, c->h - 1), pitch) + 8 * xq; // (a row past the image: the last row again, not stored)
        lab[r] = ld_vec<uint4>(c->label, (unsigned)key0 << 1);
        cd[r] = ld_vec<uint4>(c->cand, (unsigned)key0 << 1);
    }
#pragma unroll
    for (int r = 0; r < kApplyRows; r++) {
        if (y0 + r >= c->h) break;
        const int key0 = __mul24(y0 + r, pitch) + 8 * xq;
        const unsigned lw[4] = {lab[r].x, lab[r].y, lab[r].z, lab[r].w}, cw[4] = {cd[r].x, cd[r].y, cd[r].z, cd[r].w};
        unsigned l[8], pk[8], o[8];
        int t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            l[j] = (j & 1) ? lw[j >> 1] >> 16 : lw[j >> 1] & 0xffffu;
            pk[j] = (j & 1) ? cw[j >> 1] >> 16 : cw[j >> 1] & 0xffffu;
            t[j] = (l[j] != (unsigned)kNoLabel && pk[j] != l[j]) ? ld_off(c->tmin, l[j] << 2) : kIntMax;
        }
        bool changed = false;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            o[j] = t[j] < key0 + j ? pk[j] : l[j];
            changed = changed || o[j] != l[j];
        }
        if (changed)
            *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(c->label) + ((unsigned)key0 << 1)) =
                make_uint4(o[0] | o[1] << 16, o[2] | o[3] << 16, o[4] | o[5] << 16, o[6] | o[7] << 16);
    }
}

// One Huber-Newton pass (FF.cpp:536-553) of up to 64 seeds at once, one chain per lane: a = ordered sum of 2*r over the
// Huber core, +-hr (added in double) per tail element; returns the Newton step -a / (b + 10), b = 2 * (core elements).
// fetch(i) = element i of this lane's list (i is wave-uniform; any value beyond the list's end); lim = the list's
// length, 0 for a lane that does not take part.  Lanes past the end of their list add r = +0: a + 0 is a, bit for bit (a is
// never -0).  Branch-free: in a wave of 64 lists some lane nearly always holds a tail element, and a wave-uniform
// branch per element costs more than the double-typed add it would skip.
template <typename Fetch> __device__ __forceinline__ float huber_pass_lanes(Fetch fetch, int lim, float md, double hr) {
    const float hr_above = flt_above(hr); // the Huber class tests in fp32 (dsm_math.h)
    const unsigned hr_lo = (unsigned)__double_as_longlong(hr), hr_hi = (unsigned)(__double_as_longlong(hr) >> 32);
    float a = 0.0f;
    int n_tail = 0;
    float d8[8];
#pragma unroll
    for (int q = 0; q < 8; q++) d8[q] = fetch(q);
    for (int i = 0; __ballot(i < lim) != 0; i += 8) {
        float n8[8];
#pragma unroll
        for (int q = 0; q < 8; q++) n8[q] = fetch(i + 8 + q); // next block, in flight during this one
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float r = i + q < lim ? md - d8[q] : 0.0f;
            const bool core = fabsf(r) < hr_above; // (double)r < hr && (double)r > -hr
            const float a_core = a + 2 * r;
            // (float)((double)a + (r > 0 ? hr : -1 * hr)): the constant's sign bit by select, its low word is shared
            const double step = __longlong_as_double((long long)(((unsigned long long)(r > 0 ? hr_hi : hr_hi ^ 0x80000000u) << 32) | hr_lo));
            const float a_tail = (float)((double)a + step);
            a = core ? a_core : a_tail;
            n_tail += core ? 0 : 1;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) d8[q] = n8[q];
    }
    const float b = (float)(2 * (lim - n_tail)); // the reference adds 2.0f per core element: exact
    return huber_newton_step(a, b);
}

// The same pass over a list held in REGISTERS (k_update_seeds_rest: the list of a queued seed is read once and serves
// four passes).  Same operations in the same order as huber_pass_lanes; the loop is unrolled so that every v[] index is a
// constant, and leaves at the first block of eight beyond the longest list of the wave.
constexpr int kRestRegs = 128; // >= kLaneCap
__device__ __forceinline__ float huber_pass_regs(const float (&v)[kRestRegs], int lim, float md, double hr) {
    const float hr_above = flt_above(hr);
    const unsigned hr_lo = (unsigned)__double_as_longlong(hr), hr_hi = (unsigned)(__double_as_longlong(hr) >> 32);
    float a = 0.0f;
    int n_tail = 0;
#pragma unroll
    for (int i = 0; i < kRestRegs; i += 8) {
        if (__ballot(i < lim) == 0) break;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float r = i + q < lim ? md - v[i + q] : 0.0f;
            const bool core = fabsf(r) < hr_above; // (double)r < hr && (double)r > -hr
            const float a_core = a + 2 * r;
            const double step = __longlong_as_double((long long)(((unsigned long long)(r > 0 ? hr_hi : hr_hi ^ 0x80000000u) << 32) | hr_lo));
            const float a_tail = (float)((double)a + step);
            a = core ? a_core : a_tail;
            n_tail += core ? 0 : 1;
        }
    }
    const float b = (float)(2 * (lim - n_tail)); // the reference adds 2.0f per core element: exact
    return huber_newton_step(a, b);
}

// ---- update_seeds, ONE LANE PER SEED: a wave takes 64 consecutive seeds (launches batched over handles).
// The wave-per-seed form above spends most of its instructions on work one lane could do: window addressing, the
// ballot / rank compaction and two wave reductions are repeated by every wave for every window, and the ordered sums of
// a Huber-Newton pass are serial chains of adds executed by all 64 lanes (546 VALU wave-instructions per seed,
// profiles/r02_pmc_sq_batch8.md) -- and batched launches are bound by VALU issue, not by bytes.  Here every lane walks
// its own seed's 16x16 window in row-major order (16-byte loads, rows fetched three ahead), keeps the integer sums and the
// ordered depth sum in registers, compacts its member depths in order into its own LDS row ([element][lane]:
// conflict-free whatever the lanes' list lengths), and runs the first Huber-Newton pass as 64 independent chains: one
// v_add serves 64 seeds.  Same operations on the same operands in the same order as the reference, seed by seed.  The
// label image it reads is the sweep's own (k_apply_labels): 170 registers, two waves per SIMD where the form that
// applied the labels inside the walk (round 3: two more row planes, a gathered tmin per pixel) held one.
// What the first Huber pass does not finish goes to k_update_seeds_rest through two queues: the 13 % of the seeds that
// need more passes, packed 64 to a wave again, and the seeds whose list does not fit the 123 depths a lane keeps in LDS (a
// superpixel averages 53, the longest of 64 neighbours ~95; 0.05 % of all seeds have more), which get a wave
// of their own.  Same arithmetic on every path, so which one a seed takes changes nothing in its result.
// (Round 4 measured the occupancy lever of VERDICT r03 in this form: rows of 79 depths -- 20 KB, eight waves per CU
// instead of five -- with a second lane-per-seed pass over the 10 % longer lists, long rows, seeds taken from a queue:
// bit-exact, and slower in every configuration on one box, 26.5 k against 28.8-30.6 k frames/s for 32 subsequences in 4
// batches, 30.5 k against 31.7 k for 128: the second pass is a full window walk again and sits between two launches
// that wait for it.  tools/_exp/r04_update_twotier.patch.)
constexpr int kLaneCap = kRestListCap; // rows of rest_list; k_update_seeds keeps kLaneCap + 1 rows in LDS: 32 KB per wave, five waves per CU
// rest_count[2 * sweep + ...] (zeroed by k_init_seeds) / where the queues live in `worklist` (free between k_resolve and
// the next k_assign): entries of seeds that need more Huber passes (int4, from 0) | seeds queued for a wave of their own
enum { kQueueRest = 0, kQueueWave = 1 };
__device__ __forceinline__ int32_t *queue_wave(const DeviceCtx *c) { return c->worklist + 4 * c->n_seed; }
// After the sweeps the same words hold the order in which k_seed_fit of a batched launch takes the seeds, four per wave
// (k_seed_stats: by length of list within every 64 seeds; k_seed_points: as they come).
__device__ __forceinline__ int32_t *fit_order(const DeviceCtx *c) { return c->worklist + 4 * c->n_seed; }

struct LaneRow { // one window row of one lane: 16 labels, depths, intensities
    LabelQuad lab[4];
    float4 dp[4];
    unsigned im[4];
};

template <bool BATCH> __global__ __launch_bounds__(64) void k_update_seeds(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    constexpr int CAP = kLaneCap - 3; // the longest list kept here: 124 rows + the four a quad may add before the end is clamped
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    __shared__ __attribute__((aligned(16))) float s_list[(CAP + 4) * 64]; // [element][lane] + four spare rows: 32 KB
    const int lane = lane_id();
    const int S = c->n_seed;
    const FrameParams &fp = frame_params(c);
    const uint8_t *img = frame_image(c, fp);
    const float *dep = frame_depth(c, fp);
    const label_t *lbl = c->label;
    const int w = c->w, h = c->h, pitch = c->pitch;
    // bottom rows first, see seed_of_block
    const int s = (((S + 63) >> 6) - 1 - blk.x) * 64 + lane;
    const bool live = s < S;
    const int sc = live ? s : S - 1;
    int gx, gy;
    seed_cell(c, sc, gx, gy);
    const int wx0 = gx * kCell + kCell / 2 - kCell, wy0 = gy * kCell + kCell / 2 - kCell;
    const int t_self = c->tmin[sc];
    const float4 old = c->core[sc];
    const bool stats = live && t_self != kIntMax; // stable seeds keep their state: FF.cpp:479-480
    const unsigned s_match = stats ? (unsigned)s : (unsigned)kNoSeed;
    // the four quads of a window row, as pixel offsets within the row; a quad wholly outside the row (x < 0 at
    // the left border, x >= pitch where the pitch equals the width) is redirected to an in-range one and masked below
    int qx[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = wx0 + 4 * q;
        qx[q] = x < 0 ? 0 : (x > pitch - 4 ? pitch - 4 : x);
    }
    // statistics window clipped to [0, w-1) x [0, h-1): the last row and column never contribute.  What a label of window
    // column j is compared with: the seed, or no label at all where the column is outside
    unsigned s_col[kWin];
#pragma unroll
    for (int j = 0; j < kWin; j++) s_col[j] = (unsigned)(wx0 + j) < (unsigned)(w - 1) ? s_match : (unsigned)kNoSeed;

    auto load_row = [&](int r) {
        LaneRow R;
        int y = wy0 + r;
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        const unsigned row = (unsigned)__mul24(y, pitch);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned o = row + (unsigned)qx[q], o4 = o << 2;
            R.lab[q] = label_quad(lbl, o);
            R.dp[q] = ld_vec<float4>(dep, o4);
            R.im[q] = ld_vec<unsigned>(img, o);
        }
        return R;
    };

    int acc_ci = 0;  // member count << 16 | intensity sum  (<= 225 members, 225 * 255 < 2^16)
    int colcnt[kWin]; // members per window column (their column sum is sum_j j * colcnt[j]: one add-with-carry per pixel)
#pragma unroll
    for (int j = 0; j < kWin; j++) colcnt[j] = 0;
    int acc_y = 0, cnt_prev = 0; // sum of the members' window rows, from the member count of every row
    // member depths > 0.1 in window row-major order: element i of this lane at s_list[i * 64 + lane]; `tail` = byte address
    // of the list's end
    const unsigned lane4 = (unsigned)lane << 2, tail_cap = ((unsigned)CAP << 8) + lane4;
    unsigned tail = lane4;
    float sum = 0.0f; // their sequential fp32 sum, FF.cpp:511

    // one window row of every lane's seed: membership, sums, depth list
    auto process_row = [&](const LaneRow &A, int r) {
        const int y = wy0 + r;
        // Branch-free within the row (every lane is a different seed: a branch per pixel only adds exec-mask bookkeeping),
        // and every per-pixel condition is ONE vector compare whose lane mask the next instruction consumes: the column's
        // validity sits in the value the label is compared with (s_col), the row's in the exec mask of the whole row, and
        // the depth test reads the depth already masked by membership.  (Conditions combined as lane masks cost two
        // scalar instructions per pixel between two vector ones, and a wave of this kernel mostly has its SIMD to itself:
        // nothing hides the hand-over.)  The depth is stored at the list's end unconditionally and the end advances
        // only past a member depth > 0.1 (a later store overwrites a rejected one); the end is clamped to row CAP once
        // per quad -- a quad adds at most four rows, the spare ones -- and sticks there: a list that reaches CAP is `over`.
        if ((unsigned)y < (unsigned)(h - 1)) {
#pragma unroll
            for (int j = 0; j < kWin; j++) {
                const bool mem = comp(A.lab[j >> 2], j & 3) == s_col[j];
                const int pi = (int)((A.im[j >> 2] >> (8 * (j & 3))) & 0xffu);
                acc_ci += mem ? pi | 0x10000 : 0;
                colcnt[j] += mem ? 1 : 0;
                const float d = comp(A.dp[j >> 2], j & 3);
                const float dm = mem ? d : 0.0f;
                const bool dv = dm > flt_below(0.1); // FF.cpp:508, (double)d > 0.1
                if ((j & 3) == 0) tail = tail < tail_cap ? tail : tail_cap;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(s_list) + tail) = d;
                tail += dv ? 256u : 0u;
                sum += dv ? dm : 0.0f; // (+0.0f: the running sum of positive depths is never -0)
                if ((j & 3) == 3) {
                    // pin the accumulators per quad: left alone, the optimiser reassociates the integer sums of the unrolled
                    // pixels into one tree and keeps every lane mask alive for it (they spill to VGPR lanes)
                    asm volatile("" : "+v"(acc_ci), "+v"(sum), "+v"(tail), "+v"(colcnt[j - 3]), "+v"(colcnt[j - 2]), "+v"(colcnt[j - 1]), "+v"(colcnt[j]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const int cnt_now = acc_ci >> 16;
        acc_y += r * (cnt_now - cnt_prev);
        cnt_prev = cnt_now;
    };

    // Four row buffers in rotation: a row's loads are issued three rows before it is worked on.  The loop is NOT unrolled
    // further: every wave runs this code once per four rows, and a fully unrolled window (50 KB of straight-line code) is
    // paced by instruction fetch, not by the SIMD -- measured 82 us per launch against 20 us for the wave-per-seed kernel
    // it replaces.
    LaneRow B0 = load_row(0), B1 = load_row(1), B2 = load_row(2), B3;
#pragma unroll 1
    for (int r = 0; r < kWin; r += 4) {
        B3 = load_row(r + 3);
        __builtin_amdgcn_sched_barrier(0);
        process_row(B0, r);
        if (r + 4 < kWin) B0 = load_row(r + 4);
        __builtin_amdgcn_sched_barrier(0);
        process_row(B1, r + 1);
        if (r + 4 < kWin) B1 = load_row(r + 5);
        __builtin_amdgcn_sched_barrier(0);
        process_row(B2, r + 2);
        if (r + 4 < kWin) B2 = load_row(r + 6);
        __builtin_amdgcn_sched_barrier(0);
        process_row(B3, r + 3);
    }

    // ---- per-lane finish: means, stability, robust mean depth (FF.cpp:514-556)
    const int cnt = acc_ci >> 16, si = acc_ci & 0xffff;
    const int nd = (int)((tail - lane4) >> 8);
    const bool empty = stats && cnt == 0;
    if (empty) atomicMin(&c->first_empty[sweep * kWorkers + chunk_of(S, s)], s); // FF.cpp:516-517: the worker returns, abandoning the rest of its chunk
    // (a list with a +inf depth needs no pass and no complete list: its robust mean is +inf, dsm_math.h mean_depth_is_settled)
    const bool settled = sum == __builtin_inff();
    const bool over = stats && nd >= CAP && !settled;
    const bool fin = stats && cnt > 0 && !over;
    int acc_x = 0;
#pragma unroll
    for (int j = 1; j < kWin; j++) acc_x += j * colcnt[j];
    const int sx = acc_x + cnt * wx0, sy = acc_y + cnt * wy0;
    const float fn = (float)cnt;
    const float mi = (float)si / fn, mx = (float)sx / fn, my = (float)sy / fn;
    const float moved = fabsf(old.z - mi) + fabsf(old.x - mx) + fabsf(old.y - my);
    const int stable = moved < flt_above(0.2) ? 1 : 0; // (double)moved < 0.2, in fp32 (dsm_math.h, flt_above)
    float md = 0.0f;
    bool run = fin && nd > 0;
    if (run) md = sum / (float)nd;
    run = run && !settled;
    const double hr = c->huber;
    wave_lds_sync();
    // ---- the FIRST Huber-Newton pass of all 64 seeds, one chain per lane.  87 % of all seeds are done after it
    // (|delta| < 0.01: FF.cpp:554).
    if (__ballot(run) != 0) {
        const float delta = huber_pass_lanes([&](int i) { return s_list[(i < CAP ? i : CAP) * 64 + lane]; }, run ? nd : 0, md, hr);
        if (run) md = md + delta;
        if (fabsf(delta) < flt_above(0.01)) run = false; // (double)delta < 0.01 && (double)delta > -0.01
    }
    // ---- the rest goes to k_update_seeds_rest.  Seeds that need more passes (13 %; 3 % need all five) are PACKED there,
    // 64 to a wave: refining them here leaves sixty lanes idle for four more passes (45 us per launch, measured), and
    // taking them one after the other by the whole wave is worse (150 us: they are the expensive seeds, long lists full
    // of tail elements).  A queue entry is (seed, length, mean so far); the list moves to rest_list[entry / 64][i][entry % 64].
    // Seeds whose list outgrew its LDS row (0.05 %) are queued for a wave of their own.
    const unsigned long long rm = __ballot(run), om = __ballot(over);
    if (rm | om) {
        int base_r = 0, base_o = 0;
        if (lane == 0) {
            if (rm) base_r = atomicAdd(&c->rest_count[2 * sweep + kQueueRest], __popcll(rm));
            if (om) base_o = atomicAdd(&c->rest_count[2 * sweep + kQueueWave], __popcll(om));
        }
        base_r = __builtin_amdgcn_readfirstlane(base_r);
        base_o = __builtin_amdgcn_readfirstlane(base_o);
        if (over) queue_wave(c)[base_o + rank_below(om)] = s;
        const int q = base_r + rank_below(rm);
        if (run) reinterpret_cast<int4 *>(c->worklist)[q] = make_int4(s, nd, __float_as_int(md), 0);
        const unsigned dst0 = (((unsigned)(q >> 6) * kLaneCap) << 8) + ((unsigned)(q & 63) << 2);
        const int lim = run ? nd : 0;
        for (int i = 0; __ballot(i < lim) != 0; i += 4) {
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (i + t < lim) st_off(c->rest_list, dst0 + ((unsigned)(i + t) << 8), s_list[(i + t) * 64 + lane]);
        }
    }
    if (fin) { // (for a queued seed everything but the depth is final)
        c->core_stage[s] = make_float4(mx, my, mi, md);
        c->stable_stage[s] = stable;
    }
}

// What k_update_seeds left in its queues.  Workgroups (one wave each) below n_dense = ceil(S / 64): passes 2..5 of the
// queued seeds, 64 to a wave, one chain per lane over the lists in rest_list (coalesced: 64 lanes read 64 consecutive
// floats per element).  The workgroups after them: seeds whose list did not fit an LDS row, gathered and refined from
// scratch by one wave each.
constexpr int kRestOverBlocks = 128; // (32 until round 5: on the reference's kind of input hundreds of seeds per frame outgrow their LDS row, see kFitLargeBlocks)
constexpr int kLaneBatch = 8; // handles per launch from which the lane-per-seed kernels are used (launch_frame)
template <bool BATCH> __global__ __launch_bounds__(64) void k_update_seeds_rest(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    __shared__ __attribute__((aligned(16))) float s_depth[kWin * kWin], s_term[kWin * kWin];
    const int lane = lane_id();
    const int S = c->n_seed, n_dense = (S + 63) >> 6;
    if (blk.x >= n_dense) {
        const int n_over = c->rest_count[2 * sweep + kQueueWave];
        for (int e = blk.x - n_dense; e < n_over; e += kRestOverBlocks) {
            update_seed_wave(c, sweep, __builtin_amdgcn_readfirstlane(queue_wave(c)[e]), s_depth, s_term);
            wave_lds_sync();
        }
        return;
    }
    const int n = c->rest_count[2 * sweep + kQueueRest];
    if (blk.x * 64 >= n) return;
    const int q = blk.x * 64 + lane;
    const bool live = q < n;
    const int4 ent = reinterpret_cast<const int4 *>(c->worklist)[live ? q : blk.x * 64];
    const int s = ent.x, nd = ent.y;
    float md = __int_as_float(ent.z);
    const double hr = c->huber;
    const unsigned src0 = (((unsigned)blk.x * kLaneCap) << 8) + ((unsigned)lane << 2);
    const int n_max = __builtin_amdgcn_readfirstlane(wave_max_int(live ? nd : 0));
    // The lists into registers, all loads in flight at once: this kernel is pure latency (a few waves per handle between
    // two stages that wait for it), and with the list re-read from memory by every pass -- one block of eight ahead --
    // each of up to 64 blocks waited for most of a trip to the L2: 28 us, whatever the batch.
    static_assert(kRestRegs >= kLaneCap, "list registers");
    float v[kRestRegs];
#pragma unroll
    for (int b = 0; b < kRestRegs; b += 16) {
        if (b < n_max) {
#pragma unroll
            for (int q = 0; q < 16; q++) v[b + q] = ld_off(c->rest_list, src0 + ((unsigned)(b + q < n_max ? b + q : n_max - 1) << 8));
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) v[b + q] = 0.0f;
        }
    }
    bool run = live;
#pragma unroll 1
    for (int it = 1; it < 5; it++) {
        if (__ballot(run) == 0) break;
        const float delta = huber_pass_regs(v, run ? nd : 0, md, hr);
        if (run) md = md + delta;
        if (fabsf(delta) < flt_above(0.01)) run = false; // (double)delta < 0.01 && (double)delta > -0.01
    }
    if (live) c->core_stage[s].w = md;
}

// One wave per seed for ALL seeds: the launch for one handle or a few (frame groups), where what counts is the kernel's
// latency -- it ends with its slowest wave (~20 us), the lane-per-seed pair above with the slowest of its two stages each
// (~45 us) -- and not the instructions issued, which is what bounds launches batched over many handles (kLaneBatch).
// Same results, bit for bit.
template <bool BATCH> __global__ __launch_bounds__(256) void k_update_seeds_wave(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    __shared__ __attribute__((aligned(16))) float s_depth[4][kWin * kWin];
    __shared__ __attribute__((aligned(16))) float s_term[4][kWin * kWin];
    const int wv = threadIdx.x >> 6;
    // (one seed per wave: the index lives in a scalar register, and so does every address formed from it)
    const int s = __builtin_amdgcn_readfirstlane(seed_of_block(blk.x, wv, c->gw, c->gh));
    if (s < 0) return;
    update_seed_wave(c, sweep, s, s_depth[wv], s_term[wv]);
}

// Seeds at or after the first pixel-less unstable seed of their worker chunk keep their old state.
template <bool BATCH> __global__ __launch_bounds__(256) void k_commit_seeds(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const int s = blk.x * 256 + threadIdx.x;
    if (s == 0) c->work_count[0] = 0;
    if (s >= c->n_seed) return;
    if (c->tmin[s] == kIntMax) return;
    int t = -1;
    if (s < c->first_empty[sweep * kWorkers + chunk_of(c->n_seed, s)]) {
        const float4 v = c->core_stage[s];
        c->core[s] = v;
        c->inv_depth[s] = 1.0 / (double)v.w;
        if (c->stable_stage[s]) t = kIntMax;
    }
    c->tmin[s] = t;
}


} // namespace dsm
