// dsm_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the per-frame surfel-fusion hot path.
//
// Reference functions covered ("FF.cpp" = surfel_fusion/src/fusion_functions.cpp, "SM.cpp" =
// surfel_fusion/src/surfel_map.cpp of the reference):
//   k_init_seeds    initialize_seeds_kernel          FF.cpp:577-629
//   k_assign        update_pixels_kernel             FF.cpp:389-453 (+ calculate_cost 364-387)
//   k_resolve       the sequential `stable` skip rule of FF.cpp:400,445,450 as a fixed point
//   k_update_seeds  update_seeds_kernel              FF.cpp:468-562 (+ the new label image of the sweep)
//   k_commit_seeds  the early `return` of FF.cpp:516-517 (per worker chunk)
//   k_pixel_normals calculate_pixels_norms_kernel (the normals that are read)   FF.cpp:664-712
//   k_seed_stats    calculate_spaces / calculate_sp_depth_norms up to the fit's starting point
//                                                    FF.cpp:644-662, 792-871, 104-120
//   k_seed_fit      get_huber_norm's Gauss-Newton steps, the seed's plane / position / view angle
//                                                    FF.cpp:128-188, 872-914
//                   and the per-seed part of initialize_surfels, FF.cpp:315-361
//   k_fuse_surfels  fuse_surfels_kernel              FF.cpp:190-313
//   k_frame_tail    initialize_surfels (the `fused` test and the ordered list), FF.cpp:315-361;
//                   SurfelMap::fuse_map refill + swap-with-last, SM.cpp:1077-1109
//   k_warp          warp_{active,inactive}_surfels_cpu_kernel          SM.cpp:681-789
//   k_mark_key / k_scan_marks / k_extract_marked   move_add_surfels removal, SM.cpp:1476-1497
//
// Build with -ffp-contract=off: results are required to match the CPU reference bit for bit.
// The work is stencil / gather / ordered reduction: no MFMA (nothing is a dense contraction).  The superpixel kernels
// are bound by VALU instruction issue (mixed fp32 / fp64 scalar-style arithmetic in the reference's order), the
// map-sized ones (k_fuse_surfels, k_warp) by HBM.  Every frame kernel exists twice: for one handle (context in the
// kernel arguments) and, BATCH, for several handles advancing in lockstep (context array, handle = low bits of the
// dispatch index: one handle per XCD) -- see DESIGN.md section 4.
#include "dsm_device.h"

namespace dsm {

// ------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }
// number of set bits of m in lanes below mine
__device__ __forceinline__ int rank_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// Wave-wide integer sum / float max by DPP (no LDS crossbar): Hillis-Steele within each row of 16
// (row_shr 1,2,4,8), then row_bcast15 / row_bcast31 carry the row totals; lane 63 ends with the total.
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_int(int v) { // v >= 0 in every lane
#define DSM_DPP_MAXI(ctrl, rows) v = max(v, __builtin_amdgcn_update_dpp(0, v, ctrl, rows, 0xf, false))
    DSM_DPP_MAXI(0x111, 0xf); DSM_DPP_MAXI(0x112, 0xf); DSM_DPP_MAXI(0x114, 0xf); DSM_DPP_MAXI(0x118, 0xf);
    DSM_DPP_MAXI(0x142, 0xa); DSM_DPP_MAXI(0x143, 0xc);
#undef DSM_DPP_MAXI
    return __builtin_amdgcn_readlane(v, 63);
}
// v >= 0 in every lane (identity +0.0f)
__device__ __forceinline__ float wave_max(float v) {
#define DSM_DPP_MAX(ctrl, rows) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xf, false)))
    DSM_DPP_MAX(0x111, 0xf); DSM_DPP_MAX(0x112, 0xf); DSM_DPP_MAX(0x114, 0xf); DSM_DPP_MAX(0x118, 0xf);
    DSM_DPP_MAX(0x142, 0xa); DSM_DPP_MAX(0x143, 0xc);
#undef DSM_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// order LDS traffic of one wave: a lane's reads after this see every lane's writes before it
// (the LDS queue of a wave is FIFO; this only stops the compiler from moving accesses across).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int load_coherent(const int32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Ordered fp32 sum a = (((0 + l[0]) + l[1]) + ...) of an LDS list: the loads are block-fetched 16 at a
// time (four ds_read_b128) so that only the adds are loop-carried.  l is 16-byte aligned and padded
// with +0.0f up to a multiple of kBlk (a running sum that starts at +0.0f can never be -0.0f, so adding
// +0.0f is the identity, bit for bit).
constexpr int kBlk = 16;
struct Blk16 {
    float e[16];
};
__device__ __forceinline__ Blk16 load_blk(const float *l) {
    const float4 a = *reinterpret_cast<const float4 *>(l), b = *reinterpret_cast<const float4 *>(l + 4);
    const float4 c = *reinterpret_cast<const float4 *>(l + 8), d = *reinterpret_cast<const float4 *>(l + 12);
    return Blk16{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w}};
}
__device__ __forceinline__ float ordered_sum(const float *l, int n) {
    float a = 0.0f;
    for (int i = 0; i < n; i += kBlk) {
        const Blk16 v = load_blk(l + i);
#pragma unroll
        for (int q = 0; q < kBlk; q++) a += v.e[q];
    }
    return a;
}
// zero the padding slots [n, round_up(n, kBlk)) of a column
__device__ __forceinline__ void pad_column(float *l, int n, int lane) {
    if (lane < kBlk && n + lane < ((n + kBlk - 1) & ~(kBlk - 1))) l[n + lane] = 0.0f;
}

// issue priority of this wave by the length of its list (s_setprio takes an immediate; n is wave-uniform)
__device__ __forceinline__ void wave_priority(int n) {
    if (n > 96) __builtin_amdgcn_s_setprio(3);
    else if (n > 64) __builtin_amdgcn_s_setprio(2);
}

// debug: record the shader clock of phase `ph` of seed s in per-seed kernel `kid` (lane 0 only)
__device__ __forceinline__ void stamp(const DeviceCtx *c, int kid, int s, int ph, int lane) {
    if (c->stamps && lane == 0) c->stamps[((int64_t)kid * c->n_seed + s) * 8 + ph] = clock64();
}

// A block of the wave-per-seed kernels is 4 consecutive seeds; returns the seed of wave `wv`, or -1 outside the grid.
// Bottom rows first: in driving scenes they are the expensive seeds (near ground, every pixel has depth, long lists),
// the top rows are sky and leave after the gather.  Workgroups are dispatched in index order and the grid does not fit
// the machine at once, so what is dispatched last must be what finishes fastest.  (An XCD-local order -- vertical
// strips of the seed grid per XCD -- cut the fabric reads 3x and was slower: profiles/r01_xcd_mapping.md,
// tools/_exp/r02_experiments.patch.)
__device__ __forceinline__ int seed_of_block(int b, int wv, int gw, int gh) {
    const int n_blocks = (gw * gh + 3) >> 2;
    const int s = (n_blocks - 1 - b) * 4 + wv;
    return s < gw * gh ? s : -1;
}

// Launches batched over handles (grid z = handle): which handle and which block of it this workgroup takes.
// Workgroups go to the XCDs round-robin in dispatch order (x fastest, then z), so with the handle taken from the
// low bits of the dispatch index a batch of eight puts each handle on ONE XCD: the overlapping windows of a frame
// then meet in one L2 instead of being fetched over the fabric by all eight.
struct BlockOf { int z, x, y; };
template <bool BATCH> __device__ __forceinline__ BlockOf block_of() {
    if (!BATCH) return {0, (int)blockIdx.x, (int)blockIdx.y};
    const unsigned l = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), r = l / gridDim.z;
    return {(int)(l % gridDim.z), (int)(r % gridDim.x), (int)(r / gridDim.x)};
}

// A pointer loaded from memory is a generic pointer to the compiler: loads through it are flat_load (address-space check
// per access, 64-bit vector address arithmetic, and a wait that couples them to the LDS queue) instead of global_load.
// Kernel arguments are known to be global; the context of a batched launch, read from the batch's array, is not --
// it is copied out once with every pointer rebuilt as a global one (through an integer: a plain cast there and back
// is folded away before the address-space inference sees it).
template <typename T> __device__ __forceinline__ T *as_global(T *p) {
    return (T *)(__attribute__((address_space(1))) T *)(unsigned long long)p;
}
__device__ __forceinline__ DeviceCtx load_ctx(const DeviceCtx *src) {
    DeviceCtx o = *as_global(src);
#define DSM_G(f) o.f = as_global(o.f)
    DSM_G(ray_x);
    DSM_G(ray_y);
    DSM_G(img_base);
    DSM_G(depth_base);
    DSM_G(label);
    DSM_G(cand);
    DSM_G(core);
    DSM_G(inv_depth);
    DSM_G(core_stage);
    DSM_G(stable_stage);
    DSM_G(tmin);
    DSM_G(first_empty);
    DSM_G(worklist);
    DSM_G(work_count);
    DSM_G(fit_big_count);
    DSM_G(rest_count);
    DSM_G(rest_list);
    DSM_G(gn_hdr);
    DSM_G(normals);
    DSM_G(plane);
    DSM_G(seeds);
    DSM_G(spawn_rec);
    DSM_G(spawn_ok);
    DSM_G(fused_flag);
    DSM_G(spawn_idx);
    DSM_G(local);
    DSM_G(fresh);
    DSM_G(n_local);
    DSM_G(n_local_next);
    DSM_G(n_new);
    DSM_G(hole_mask);
    DSM_G(wave_prefix);
    DSM_G(holes);
    DSM_G(n_holes);
    DSM_G(hole_chunk);
    DSM_G(params);
    DSM_G(cursor);
    DSM_G(status);
    DSM_G(cur);
    DSM_G(stamps);
    DSM_G(seed_weight);
#undef DSM_G
    return o;
}

// Element at a 32-bit BYTE offset from a wave-uniform base: compiles to global_load v, v_off, s[base] -- the offset is the
// vector address.  Indexing with an int (p[y * pitch + x]) costs a sign extension, a 64-bit shift and a 64-bit add in
// the vector ALU per access, and a 64-bit multiply-add where the index is formed; the per-seed kernels make a dozen
// such accesses per lane and are bound by instruction issue.
template <typename T> __device__ __forceinline__ T ld_off(const T *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename T> __device__ __forceinline__ void st_off(T *base, unsigned byte_off, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
template <typename T> __device__ __forceinline__ T ld_vec(const void *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ int comp(const int4 &v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }
// Label planes (label_t, 16 bits per pixel).  One pixel as the reference's int (-1 = no superpixel) / four pixels of a
// row as they lie in memory, and pixel t of the four as its 16 bits (compared with a seed index as they are: kNoLabel
// equals none, and "no seed" on the other side is a value above 16 bits).
__device__ __forceinline__ int label_at(const label_t *plane, unsigned pixel) {
    const int l = (int)ld_off(plane, pixel << 1);
    return l == kNoLabel ? -1 : l;
}
__device__ __forceinline__ void label_put(label_t *plane, unsigned pixel, int l) { st_off(plane, pixel << 1, (label_t)l); } // (-1 -> kNoLabel)
typedef uint2 LabelQuad;
__device__ __forceinline__ LabelQuad label_quad(const label_t *plane, unsigned pixel) { return ld_vec<LabelQuad>(plane, pixel << 1); }
__device__ __forceinline__ unsigned comp(const LabelQuad &v, int t) { return t == 0 ? v.x & 0xffffu : t == 1 ? v.x >> 16 : t == 2 ? v.y & 0xffffu : v.y >> 16; }
constexpr int kNoSeed = 0x10000; // compared with 16 bits of a label plane: equals no label
__device__ __forceinline__ float comp(const float4 &v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }
// grid cell of seed s (s < 65 536: dsm_create): the quotient by multiplication with the reciprocal the host rounded up
__device__ __forceinline__ void seed_cell(const DeviceCtx *c, int s, int &gx, int &gy) {
    gy = c->gw > 1 ? (int)__umulhi((unsigned)s, c->gw_magic) : s;
    gx = s - (int)__umul24((unsigned)gy, (unsigned)c->gw);
}

__device__ __forceinline__ const FrameParams &frame_params(const DeviceCtx *c) { return c->cur->p; }
__device__ __forceinline__ const uint8_t *frame_image(const DeviceCtx *c, const FrameParams &) { return as_global(c->cur->img); }
__device__ __forceinline__ const float *frame_depth(const DeviceCtx *c, const FrameParams &) { return as_global(c->cur->dep); }

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
// from the last to the first so that the lowest row and column with a depth is what remains (FF.cpp:600-626).
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
        bool hit = false;
        float first = 0.0f;
#pragma unroll 4
        for (int r = 2 * kCell - 1; r >= 0; r--) {
            const int y = wy0 + r;
            const bool row_in = need && y >= y_lo && y < y_hi;
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int x = wx0 + 4 * q; // multiple of 4: 16-byte aligned, and never straddles x = 0
                v[q] = (row_in && x >= 0) ? *reinterpret_cast<const float4 *>(dep + y * pitch + x) : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int q = 3; q >= 0; q--) {
                const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                for (int t = 3; t >= 0; t--) {
                    const int x = wx0 + 4 * q + t;
                    if (row_in && x >= x_lo && x < x_hi && e[t] > flt_below(0.01)) { hit = true; first = e[t]; }
                }
            }
        }
        if (need && hit) md = first;
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
    __shared__ float4 s_core[kTileCellsX * kTileCellsY];
    __shared__ double s_inv[kTileCellsX * kTileCellsY];
    __shared__ float s_invf[kTileCellsX * kTileCellsY]; // the same rounded to float, for the filtered pick
    const FrameParams &fp = frame_params(c);
    const uint8_t *img = frame_image(c, fp);
    const float *dep = frame_depth(c, fp);
    const label_t *label_in = c->label; // the previous sweep's image (sweep >= 1)
    const int w = c->w, h = c->h, pitch = c->pitch, gw = c->gw, gh = c->gh;
    const int bx = blk.x * kTileW, by = blk.y * kTileH;
    const int cx0 = bx / kCell - 1, cy0 = by / kCell - 1;
    const int tid = threadIdx.x;
    if (tid < kTileCellsX * kTileCellsY) {
        const int gx = cx0 + tid % kTileCellsX, gy = cy0 + tid / kTileCellsX;
        if (gx >= 0 && gx < gw && gy >= 0 && gy < gh) {
            s_core[tid] = c->core[gy * gw + gx];
            const double inv = c->inv_depth[gy * gw + gx];
            s_inv[tid] = inv;
            s_invf[tid] = (float)inv;
        }
    }
    __syncthreads();
    const int x = bx + (tid & (kTileW - 1)), y0 = by + (tid / kTileW) * kColumn; // y0 is a multiple of 4: one quadrant row
    if (x >= w || y0 >= h) return;
    // the column's pixels, one round trip
    float pix_i[kColumn], pix_d[kColumn];
    int lab[kColumn];
    const unsigned p0 = (unsigned)(__mul24(y0, pitch) + x);
#pragma unroll
    for (int r = 0; r < kColumn; r++) {
        const unsigned p = y0 + r < h ? p0 + (unsigned)(r * pitch) : p0, p4 = p << 2; // byte offsets into the 4-byte planes, see ld_off
        pix_i[r] = (float)ld_off(img, p);
        pix_d[r] = ld_off(dep, p4);
        lab[r] = FIRST ? 0 : label_at(label_in, p);
    }
    const PickQuad quad = pick_quad(x, y0, gw, gh, [&](int gx, int gy, float &sx, float &sy, float &si, float &sd, float &inv_f) {
        const int li = __mul24(gy - cy0, kTileCellsX) + (gx - cx0);
        const float4 v = s_core[li];
        sx = v.x; sy = v.y; si = v.z; sd = v.w;
        inv_f = s_invf[li];
    });
#pragma unroll
    for (int r = 0; r < kColumn; r++) {
        const int y = y0 + r;
        if (y >= h) break;
        const int p = (int)p0 + r * pitch;
        if (!has_candidate_cell(x, y, gw, gh)) {
            // ragged border beyond every cell's reach: label -1, once per frame (no later stage changes these pixels:
            // every seed window ends before them, and k_apply_labels keeps a -1)
            if (FIRST) label_put(c->label, (unsigned)p, -1);
            continue;
        }
        // the argmin from fp32 costs with error bounds where that is decisive (dsm_math.h, pick_seed_fast); the few
        // near-ties of a wave take the reference's typed arithmetic
        int pick = pick_seed_fast(quad, x, y, pix_i[r], pix_d[r], gw);
        if (pick == kPickUnsure)
            pick = pick_seed(x, y, pix_i[r], pix_d[r], gw, gh,
                             [&](int gx, int gy, float &sx, float &sy, float &si, bool &has_d, double &inv_d) {
                                 const int li = __mul24(gy - cy0, kTileCellsX) + (gx - cx0);
                                 const float4 v = s_core[li];
                                 sx = v.x; sy = v.y; si = v.z;
                                 has_d = v.w > 0;
                                 inv_d = s_inv[li];
                             });
        const int l = lab[r];
        if (pick < 0) { // every candidate cost >= the reference's 1e6 sentinel: it would index seeds[-1]
            atomicOr(c->status, kStatusBadPick);
            if (FIRST) label_put(c->label, (unsigned)p, 0); else label_put(c->cand, (unsigned)p, l);
        } else if (FIRST) {
            label_put(c->label, (unsigned)p, pick);
        } else {
            label_put(c->cand, (unsigned)p, pick);
            const int tl = ld_off(c->tmin, (unsigned)l << 2); // -1 never changes; >= 0 only moves among values >= 0
            if (tl == -1) {
                // the old seed was unstable at sweep start: this pixel is evaluated whatever happens
                // elsewhere, so its pick loses `stable` no later than at p
                if (load_coherent(&c->tmin[pick]) > p) atomicMin(&c->tmin[pick], p);
            } else if (pick != l && c->tmin[pick] != -1) {
                // old and new seed both stable at sweep start: whether this pixel is evaluated depends
                // on the scan order -- resolved below
                const int slot = atomicAdd(c->work_count, 1);
                c->worklist[slot] = p;
            }
        }
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
        md = huber_passes_wave(dl, lt, nd, md, 0, c->huber, lane);
    }
    stamp(c, sweep, s, 5, lane);
    if (c->stamps && lane == 0) c->stamps[((int64_t)sweep * c->n_seed + s) * 8 + 7] = nd;
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
template <bool BATCH> __global__ __launch_bounds__(256) void k_apply_labels(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int sweep) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const int pitch = c->pitch;
    const int xq = blk.x * 64 + (threadIdx.x & 63), y = blk.y * 4 + (threadIdx.x >> 6);
    if (8 * xq >= pitch || y >= c->h) return;
    const int key0 = __mul24(y, pitch) + 8 * xq;
    const uint4 lab = ld_vec<uint4>(c->label, (unsigned)key0 << 1), cd = ld_vec<uint4>(c->cand, (unsigned)key0 << 1);
    const unsigned lw[4] = {lab.x, lab.y, lab.z, lab.w}, cw[4] = {cd.x, cd.y, cd.z, cd.w};
    unsigned l[8], o[8];
    int t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        l[j] = (j & 1) ? lw[j >> 1] >> 16 : lw[j >> 1] & 0xffffu;
        t[j] = l[j] != (unsigned)kNoLabel ? ld_off(c->tmin, l[j] << 2) : kIntMax;
    }
    bool changed = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned pk = (j & 1) ? cw[j >> 1] >> 16 : cw[j >> 1] & 0xffffu;
        o[j] = t[j] < key0 + j ? pk : l[j];
        changed = changed || o[j] != l[j];
    }
    if (changed)
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(c->label) + ((unsigned)key0 << 1)) =
            make_uint4(o[0] | o[1] << 16, o[2] | o[3] << 16, o[4] | o[5] << 16, o[6] | o[7] << 16);
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
// label image it reads is the sweep's own (k_apply_labels): 193 registers, two waves per SIMD where the form that
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
    const bool over = stats && nd >= CAP;
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
constexpr int kRestOverBlocks = 32;
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
    if (c->stamps && lane == 0) c->stamps[((int64_t)3 * c->n_seed + s) * 8 + 7] = n;
}

// ---- seed statistics without a wave per seed
// k_pixel_normals, one thread per pixel: the forward-difference normal (FF.cpp:664-712) of every pixel that is a depth
// inlier of its own superpixel (FF.cpp:846-850: member, depth > 0.05, |mean depth - depth| < HUBER_RANGE), written into
// a 12 B/pixel plane; the other pixels' entries are stale and never read.  (calculate_pixels_norms computes all of them;
// only these are ever read, FF.cpp:852-857.)
template <bool BATCH> __global__ __launch_bounds__(256) void k_pixel_normals(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    const FrameParams &fp = frame_params(c);
    const float *dep = frame_depth(c, fp);
    const int w = c->w, h = c->h, pitch = c->pitch;
    const int x = blk.x * 64 + (threadIdx.x & 63), y = blk.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const unsigned p = (unsigned)(__mul24(y, pitch) + x), p4 = p << 2;
    const float d = ld_off(dep, p4);
    const int l = label_at(c->label, p);
    // only the depth inliers of their own superpixel are ever read (k_seed_stats asks for exactly those): nothing is
    // stored for any other pixel; an inlier on the image border has no normal (FF.cpp:670-677) and stores zeros
    if (!(l >= 0 && d > flt_below(0.05))) return;                      // (double)d > 0.05
    const float md = ld_off(reinterpret_cast<const float *>(c->core), ((unsigned)l << 4) + 12u);
    const bool interior = x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2;
    float d_right = 0.0f, d_down = 0.0f;
    if (interior) { // (neighbours fetched before the inlier test is known: one round trip)
        d_right = ld_off(dep, p4 + 4u);
        d_down = ld_off(dep, p4 + ((unsigned)pitch << 2));
    }
    const float rx0 = ld_off(c->ray_x, (unsigned)x << 2), rx1 = ld_off(c->ray_x, ((unsigned)x << 2) + 4u);
    const float ry0 = ld_off(c->ray_y, (unsigned)y << 2), ry1 = ld_off(c->ray_y, ((unsigned)y << 2) + 4u);
    if (!(fabsf(md - d) < flt_above(c->huber))) return;
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    if (interior) pixel_normal_rays(rx0, rx1, ry0, ry1, d, d_right, d_down, nx, ny, nz);
    float *o = reinterpret_cast<float *>(reinterpret_cast<char *>(c->normals) + p * 12u);
    o[0] = nx; o[1] = ny; o[2] = nz;
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
constexpr int kFitLargeBlocks = 16;               // workgroups per handle working the queue off
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

template <int TIER> struct FitShape {
    static constexpr int kStride = TIER == kFitSmall ? kFitSmallStride : kFitStride;
    static constexpr int kChunks = TIER == kFitSmall ? (kFitSmallCap + 63) / 64 : 4; // 64-element chunks a list can span
};

// the group of seeds s0 .. s0+3 on one wave
template <int TIER> __device__ __forceinline__ void fit_group(const DeviceCtx *__restrict__ c, int s0,
                                                             float (*s_col)[kFitCols][FitShape<TIER>::kStride], float *s_ones,
                                                             double (*s_solver)[52]) {
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
        for (int it = 0; it < 5; it++) {
            if (it == 1) stamp(c, 4, s0, 3, lane);
            // residuals and Huber classes of every seed's list, lane-parallel; the class masks of a seed stay with its lanes
            unsigned long long noncore[4] = {0, 0, 0, 0};
            float pn[kFitSeeds][4]; // every seed's plane, wave-uniform
#pragma unroll
            for (int q = 0; q < kFitSeeds; q++) {
                pn[q][0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nx), q * kFitLanes));
                pn[q][1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ny), q * kFitLanes));
                pn[q][2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nz), q * kFitLanes));
                pn[q][3] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nb), q * kFitLanes));
            }
            // elements 0..63 of all four lists without a branch in between: four independent instruction streams for
            // the scheduler to interleave (a wave of this kernel has few neighbours to hide its latencies behind)
#pragma unroll
            for (int q = 0; q < kFitSeeds; q++) {
                const bool valid = lane < mg[q];
                const float r = pq[q][0][0] * pn[q][0] + pq[q][0][1] * pn[q][1] + pq[q][0][2] * pn[q][2] + pn[q][3];
                // Huber class (huber_class32) as selects: the residual column carries r for a core element and the tail
                // sign for an outlier, 0 for a NaN residual (see fit_ordered_sum)
                const bool in_core = fabsf(r) < hr_above;
                const float tail_v = r >= hr_above ? 1.0f : (r <= -hr_above ? -1.0f : 0.0f);
                if (valid) s_col[q][3][lane] = in_core ? r : tail_v;
                const unsigned long long mask = __ballot(valid && !in_core);
                if (g == q) noncore[0] = mask;
            }
#pragma unroll
            for (int q = 0; q < kFitSeeds; q++) {
#pragma unroll
                for (int k = 1; k < kChunks; k++) {
                    if (k * 64 < mg[q]) { // wave-uniform
                        const int i = k * 64 + lane;
                        const bool valid = i < mg[q];
                        const float r = pq[q][k][0] * pn[q][0] + pq[q][k][1] * pn[q][1] + pq[q][k][2] * pn[q][2] + pn[q][3];
                        const bool in_core = fabsf(r) < hr_above;
                        const float tail_v = r >= hr_above ? 1.0f : (r <= -hr_above ? -1.0f : 0.0f);
                        if (valid) s_col[q][3][i] = in_core ? r : tail_v;
                        const unsigned long long mask = __ballot(valid && !in_core);
                        if (g == q) noncore[k] = mask;
                    }
                }
            }
            wave_lds_sync();
            const double acc = fit_ordered_sum(xc, yc, xs, ys, m8, noncore, is_j, hr);
            // The Hessian sums read nothing but the points and which elements are in the Huber core: while the class
            // masks of all four seeds stay what they were when H was last summed (from the second step on they are
            // normally all-core), H, its damped inverse and the determinant are bit for bit the same, and the
            // inverse still sits in LDS: only J is new.
            bool same = it > 0;
#pragma unroll
            for (int k = 0; k < 4; k++) same = same && noncore[k] == h_masks[k];
            const bool reuse_inverse = __ballot(!same) == 0;
            if (gl >= 10 && gl < 14) SJ[gl - 10] = acc;
            if (!reuse_inverse) {
#pragma unroll
                for (int k = 0; k < 4; k++) h_masks[k] = noncore[k];
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
    if (c->stamps && lane == 0) c->stamps[((int64_t)4 * c->n_seed + s0) * 8 + 7] = m_max;
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
    if (TIER == kFitLarge) {
        const int n_big = c->fit_big_count[0];
        for (int e = blk.x; e < n_big; e += kFitLargeBlocks) {
            fit_group<TIER>(c, c->worklist[e] * kFitSeeds, s_col, s_ones, s_solver);
            wave_lds_sync();
        }
    } else {
        const int n_groups = (c->n_seed + kFitSeeds - 1) / kFitSeeds;
        fit_group<TIER>(c, (n_groups - 1 - blk.x) * kFitSeeds, s_col, s_ones, s_solver); // bottom rows (long lists) first, see seed_of_block
    }
}

// ------------------------------------------------------------------------------ fuse surfels
// One lane per surfel.  Pure gather: a surfel reads one depth pixel, one label and one seed and rewrites
// only itself; the single shared write is the idempotent `fused` flag of the seed.
// The 44-byte records are an array of structures: a wave moves 64 of them (176 16-byte vectors) through its part of
// the LDS tile with fully coalesced loads, a lane owns one record at a stride of 11 dwords (odd: conflict-free), and
// the 64 are stored back -- again coalesced -- only if one of them changed.  This is the stage that scales
// with the map: 88 B per live surfel.  For a map whose surfels are all in view (bench.py's fuse_8M) the time is one
// quarter streaming the records in (55 us of 222 at 8 M surfels: 6.4 TB/s), one fifth the store-back, and the rest
// the gathers: a wave's 64 surfels touch ~65 cache lines of label / depth / seed data, more bytes than its records.
// Deleted slots are reported as one ballot per wave (hole bitmap for the compaction).
constexpr int kRecDw = sizeof(dsm_surfel) / 4; // 11

// coalesced copy of `cnt` consecutive records between global memory and LDS (records start 16-byte aligned)
__device__ __forceinline__ void records_to_lds(float *s_rec, const dsm_surfel *src, int cnt, int tid) {
    const int n_dw = cnt * kRecDw;
    const float *s1 = reinterpret_cast<const float *>(src);
    for (int v = tid; v * 4 < n_dw; v += 256) {
        if (v * 4 + 4 <= n_dw) reinterpret_cast<float4 *>(s_rec)[v] = reinterpret_cast<const float4 *>(s1)[v];
        else
            for (int e = v * 4; e < n_dw; e++) s_rec[e] = s1[e];
    }
}
__device__ __forceinline__ void rec_store(float4 *p, const float4 &v);
template <int NT = 256> __device__ __forceinline__ void records_from_lds(dsm_surfel *dst, const float *s_rec, int cnt, int tid) {
    const int n_dw = cnt * kRecDw;
    float *d1 = reinterpret_cast<float *>(dst);
    for (int v = tid; v * 4 < n_dw; v += NT) {
        if (v * 4 + 4 <= n_dw) rec_store(reinterpret_cast<float4 *>(d1) + v, reinterpret_cast<const float4 *>(s_rec)[v]);
        else
            for (int e = v * 4; e < n_dw; e++) d1[e] = s_rec[e];
    }
}

// The same copy split in two: the loads of a block of records are issued into registers one loop trip ahead and landed
// in LDS when the trip starts.  A block that loads, works and stores in turn has bytes in flight for a fraction of its
// life only, and HBM bandwidth is bytes in flight over latency; with the next block's records on their way during the
// gathers, the arithmetic and the store-back, a CU keeps about twice as many.
// (three named vectors, not an array: an aggregate indexed in a loop ends up in scratch memory here)
struct RecRegs {
    float4 v0, v1, v2; // 256 records = 704 16-byte vectors: 2.75 per thread
};
// The map-sized kernels stream every record once per launch: non-temporal loads and stores (no reuse worth a cache line:
// k_warp at 8 M surfels 170 -> 157 us, 4.15 -> 4.5 TB/s; the headline, whose maps are re-read one frame later from
// whatever cache still holds them, is unchanged).  The builtins want a native vector type, not HIP's float4 struct.
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rec_load(const float4 *p) {
    const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void rec_store(float4 *p, const float4 &v) {
    v4f_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_t *>(p));
}
// (NT threads share the copy: a workgroup's 256 with 256 records, or a wave's 64 with 64 records -- 2.75 vectors per thread either way)
template <int NT = 256> __device__ __forceinline__ RecRegs records_issue(const dsm_surfel *src, int cnt, int tid) {
    const int n_dw = cnt * kRecDw;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    // unconditional (a vector beyond the block re-reads vector 0): no branch to wait behind
    RecRegs p;
    p.v0 = rec_load(s4 + (tid * 4 + 4 <= n_dw ? tid : 0));
    p.v1 = rec_load(s4 + ((tid + NT) * 4 + 4 <= n_dw ? tid + NT : 0));
    p.v2 = rec_load(s4 + ((tid + 2 * NT) * 4 + 4 <= n_dw ? tid + 2 * NT : 0));
    return p;
}
__device__ __forceinline__ void records_land_one(float *s_rec, const float4 &val, const float *s1, int n_dw, int v) {
    if (v * 4 + 4 <= n_dw) reinterpret_cast<float4 *>(s_rec)[v] = val;
    else if (v * 4 < n_dw) // ragged last vector of the array
        for (int e = v * 4; e < n_dw; e++) s_rec[e] = s1[e];
}
template <int NT = 256> __device__ __forceinline__ void records_land(float *s_rec, const RecRegs &p, const dsm_surfel *src, int cnt, int tid) {
    const int n_dw = cnt * kRecDw;
    const float *s1 = reinterpret_cast<const float *>(src);
    records_land_one(s_rec, p.v0, s1, n_dw, tid);
    records_land_one(s_rec, p.v1, s1, n_dw, tid + NT);
    records_land_one(s_rec, p.v2, s1, n_dw, tid + 2 * NT);
}

template <bool BATCH> __global__ __launch_bounds__(256) void k_fuse_surfels(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    __shared__ __attribute__((aligned(16))) float s_rec[256 * kRecDw];
    const FrameParams &fp = frame_params(c);
    const float *dep = frame_depth(c, fp);
    const int M = c->n_local[0];
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    // a large map's deleted slots are listed by several workgroups of k_frame_tail: they need the holes per chunk of the
    // bitmap, and the map size this frame started with (the tail's first workgroup moves n_local)
    const bool big_map = M > kTailFastWords * 64;
    if (big_map && blk.x == 0 && tid == 0) c->hole_chunk[c->n_hole_chunk + 1] = M;
    FuseConst fc;
    fc.k = c->k; fc.far_d = c->far_d; fc.near_d = c->near_d;
    fc.baseline = c->baseline; fc.disp_err = c->disp_err; fc.min_tol = c->min_tol;
    fc.w = c->w; fc.h = c->h;
    fuse_const_prepare(fc);
    const int ref_idx = fp.ref_idx;
    // the two matrices once, into scalar registers: read through `fp` inside the loop they are fetched again every trip
    // (the compiler cannot rule out that the stores to the map alias them), a dependent round trip before a surfel can
    // even be projected
    float inv[16], pose[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        inv[q] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fp.inv[q])));
        pose[q] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fp.pose[q])));
    }
    // A WAVE moves its own 64 records (2 816 B = 176 vectors, 16-byte aligned) through its own quarter of the LDS tile and
    // never waits for the other three: no workgroup barrier, the waves of a CU drift apart and their loads, gathers and
    // stores overlap instead of marching in step.
    const int stride = gridDim.x * 256;
    float *s_w = s_rec + wv * 64 * kRecDw;
    const int first = blk.x * 256 + wv * 64;
    RecRegs ahead = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (first < M) ahead = records_issue<64>(c->local + first, M - first < 64 ? M - first : 64, lane);
    for (int base = first; base < M; base += stride) {
        const int cnt = M - base < 64 ? M - base : 64;
        records_land<64>(s_w, ahead, c->local + base, cnt, lane);
        wave_lds_sync();
        if (base + stride < M) ahead = records_issue<64>(c->local + base + stride, M - base - stride < 64 ? M - base - stride : 64, lane);
        bool hole = false, changed = false;
        if (lane < cnt) {
            float *r = s_w + lane * kRecDw;
            Surfel e;
            e.px = r[0]; e.py = r[1]; e.pz = r[2]; e.nx = r[3]; e.ny = r[4]; e.nz = r[5];
            e.size = r[6]; e.color = r[7]; e.weight = r[8];
            e.update_times = __float_as_int(r[9]); e.last_update = __float_as_int(r[10]);
            int ui, vi;
            float pc[3], nc[3];
            FuseOutcome oc = fuse_project(fc, ref_idx, inv, e, ui, vi, pc, nc);
            if (oc == kFuseNeedPixel) {
                const unsigned p4 = (unsigned)(__mul24(vi, c->pitch) + ui) << 2; // byte offsets, see ld_off
                const int sidx = label_at(c->label, p4 >> 2);
                const float pix_depth = ld_off(dep, p4);
                SeedView sd = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // label -1 (ragged border): the all-zero seed, see has_candidate_cell
                float w1 = 0.0f;
                if (sidx >= 0) {
                    w1 = ld_off(c->seed_weight, (unsigned)sidx << 2);
                    const float *sf = reinterpret_cast<const float *>(c->seeds);
                    const unsigned so = __umul24((unsigned)sidx, (unsigned)sizeof(dsm_seed));
                    static_assert(offsetof(dsm_seed, size) == 8 && offsetof(dsm_seed, norm_x) == 12 && offsetof(dsm_seed, posi_x) == 24 &&
                                      offsetof(dsm_seed, view_cos) == 36 && offsetof(dsm_seed, mean_depth) == 40 &&
                                      offsetof(dsm_seed, mean_intensity) == 44,
                                  "Superpixel_seed layout (elements.h:5-20)");
                    sd.size = ld_off(sf, so + 8); sd.nx = ld_off(sf, so + 12); sd.ny = ld_off(sf, so + 16); sd.nz = ld_off(sf, so + 20);
                    sd.px = ld_off(sf, so + 24); sd.py = ld_off(sf, so + 28); sd.pz = ld_off(sf, so + 32);
                    sd.view_cos = ld_off(sf, so + 36); sd.mean_depth = ld_off(sf, so + 40); sd.mean_intensity = ld_off(sf, so + 44);
                }
                oc = fuse_update(fc, ref_idx, pose, e, pc, nc, pix_depth, sd, w1);
                // the seed's `fused` mark: idempotent, but ~60 surfels fuse into a seed and a byte store into a line that
                // thousands of lanes are writing is a read-modify-write in L2 -- look first (a stale 0 only repeats the store)
                if (oc == kFuseFused && c->fused_flag[sidx] == 0) { c->seeds[sidx].fused = 1; c->fused_flag[sidx] = 1; }
            }
            if (oc == kFuseDeleted) {
                r[9] = __int_as_float(0);
                changed = true;
            } else if (oc == kFuseFused) {
                r[0] = e.px; r[1] = e.py; r[2] = e.pz; r[3] = e.nx; r[4] = e.ny; r[5] = e.nz;
                r[6] = e.size; r[7] = e.color; r[8] = e.weight;
                r[9] = __int_as_float(e.update_times); r[10] = __int_as_float(e.last_update);
                changed = true;
            }
            hole = e.update_times == 0;
        }
        const unsigned long long m = __ballot(hole);
        if (lane == 0) {
            c->hole_mask[base >> 6] = m;
            if (big_map && m) atomicAdd(&c->hole_chunk[base / (64 * kTailChunkWords)], __popcll(m)); // (deletions are rare)
        }
        wave_lds_sync();
        if (__ballot(changed) != 0) records_from_lds<64>(c->local + base, s_w, cnt, lane); // (stored back only if a surfel of the 64 changed)
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------ block scan helper
// exclusive prefix sum of one int per thread over a 1024-thread block; returns the block total
__device__ __forceinline__ int block_scan_1024(int v, int &excl, int *s_wave /* [17] */) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads(); // s_wave reuse across calls
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 16; i++) { const int t = s_wave[i]; s_wave[i] = run; run += t; }
        s_wave[16] = run;
    }
    __syncthreads();
    excl = s_wave[wv] + inc - v;
    return s_wave[16];
}

// ------------------------------------------------------------------------------ new surfels
// initialize_surfels: seeds in index order -> ordered stream compaction by one workgroup.
// initialize_surfels (FF.cpp:315-361) as an ordered stream compaction by one workgroup.  k_seed_planes left
// the would-be surfel of every qualifying seed in spawn_rec / spawn_ok; what remains is the `fused` test,
// the ordered list of creating seeds (spawn_idx) and, without compaction, the copy into `fresh`.
constexpr int kMaxSeedRounds = 64; // seeds <= 64 * 1024 (checked by dsm_create)

__device__ __forceinline__ int tail_spawn_list(const DeviceCtx *__restrict__ c, int *s_cnt /* [kMaxSeedRounds*16+1] */) {
    const int S = c->n_seed;
    const int rounds = (S + 1023) / 1024;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    unsigned long long mine = 0;
    for (int r0 = 0; r0 < rounds; r0 += 8) { // two byte loads per seed, 8 rounds per batch
        unsigned char ok[8], fu[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int s = (r0 + q) * 1024 + threadIdx.x;
            ok[q] = c->spawn_ok[s < S ? s : 0];
            fu[q] = c->fused_flag[s < S ? s : 0];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = r0 + q;
            if (r >= rounds) break;
            const bool spawn = r * 1024 + (int)threadIdx.x < S && ok[q] && !fu[q];
            if (spawn) mine |= 1ull << r;
            const unsigned long long m = __ballot(spawn);
            if (lane == 0) s_cnt[r * 16 + wv] = __popcll(m);
        }
    }
    __syncthreads();
    // exclusive scan of the rounds*16 wave counts (seed order = round-major, then wave), by wave 0
    if (wv == 0) {
        int run = 0;
        for (int base = 0; base < rounds * 16; base += 64) {
            const int i = base + lane;
            const int v = i < rounds * 16 ? s_cnt[i] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (i < rounds * 16) s_cnt[i] = run + inc - v;
            run += __shfl(inc, 63);
        }
        if (lane == 0) s_cnt[kMaxSeedRounds * 16] = run;
    }
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        const bool spawn = (mine >> r) & 1ull;
        const unsigned long long m = __ballot(spawn);
        if (spawn) c->spawn_idx[s_cnt[r * 16 + wv] + rank_below(m)] = r * 1024 + threadIdx.x;
    }
    const int K = s_cnt[kMaxSeedRounds * 16];
    if (threadIdx.x == 0) c->n_new[0] = K;
    return K;
}

// ------------------------------------------------------------------------------ hole scan
// Ascending list of deleted slots (SM.cpp:1078-1083) from the per-wave bitmaps.
// A thread owns kScanWords consecutive bitmap words per round (two 16-byte loads each pair, coalesced across the block): one
// block scan orders 8 192 words = 524 288 surfels, so a 2 M-surfel map takes four rounds and an 8 M one sixteen (one word
// per thread and round: 31 and 122).
// One round: the kTailChunkWords words from `base`, holes before them = `run`; returns the holes of the round.
__device__ __forceinline__ int tail_hole_round(const DeviceCtx *__restrict__ c, int *s_wave /* [17] */, int n_word, int base, int run) {
    const int v0 = base + (int)threadIdx.x * kScanWords;
    unsigned long long m[kScanWords];
#pragma unroll
    for (int q = 0; q < kScanWords; q += 2) { // (the allocation holds cap / 64 + 1 words, rounded up by dev_alloc's slack)
        ulonglong2 two = make_ulonglong2(0, 0);
        if (v0 + q < n_word) two = *reinterpret_cast<const ulonglong2 *>(c->hole_mask + v0 + q);
        m[q] = two.x;
        m[q + 1] = v0 + q + 1 < n_word ? two.y : 0ull;
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < kScanWords; q++) cnt += __popcll(m[q]);
    int excl;
    const int total = block_scan_1024(cnt, excl, s_wave);
    int o = run + excl;
#pragma unroll
    for (int q = 0; q < kScanWords; q++) {
        if (v0 + q < n_word) {
            c->wave_prefix[v0 + q] = o;
            unsigned long long w = m[q];
            while (w) {
                const int b = __ffsll((long long)w) - 1;
                c->holes[o++] = (v0 + q) * 64 + b;
                w &= w - 1;
            }
        }
    }
    return total;
}
__device__ __forceinline__ int tail_hole_scan(const DeviceCtx *__restrict__ c, int *s_wave /* [17] */, int M) {
    const int n_word = (M + 63) >> 6;
    int run = 0;
    for (int base = 0; base < n_word; base += kTailChunkWords) run += tail_hole_round(c, s_wave, n_word, base, run);
    if (threadIdx.x == 0) c->n_holes[0] = run;
    return run;
}
// sum of the first n per-chunk hole counts of k_fuse_surfels (whole workgroup)
__device__ __forceinline__ int tail_chunk_holes(const DeviceCtx *__restrict__ c, int *s_wave /* [17] */, int n) {
    int part = 0, excl;
    for (int i = threadIdx.x; i < n; i += 1024) part += c->hole_chunk[i];
    return block_scan_1024(part, excl, s_wave);
}

// ------------------------------------------------------------------------------ compaction
// SM.cpp:1087-1109 in parallel-exact form.  D = holes ascending (k of them), K new surfels.
//   new j          -> D[k-1-j] while j < k, else appended in order;
//   if K < k, the r = k-K smallest holes remain.  Taken in descending order H[i] = D[r-1-i], step i
//   copies the element at index M-1-i (the then-last element) into H[i] and shrinks the array.  A
//   source index that is itself a remaining hole H[j] (j < i) was overwritten in step j by the
//   element at M-1-j: follow that chain to a live element.  Targets >= M-r are cut off anyway.
// Every target is written by exactly one thread and no thread reads a slot another one writes
// (sources are live slots >= M-r or prepared new surfels), so the copy is done in place.
__device__ __forceinline__ bool is_hole(const DeviceCtx *c, int i, int &rank) {
    const unsigned long long m = c->hole_mask[i >> 6];
    const int b = i & 63;
    rank = c->wave_prefix[i >> 6] + __popcll(m & ((1ull << b) - 1ull));
    return (m >> b) & 1ull;
}

__device__ __forceinline__ void tail_compact(const DeviceCtx *__restrict__ c, int M, int K, int k) {
    const int tid = threadIdx.x, nthr = 1024;
    dsm_surfel *local = c->local;
    const dsm_surfel *rec = c->spawn_rec;
    const int32_t *idx = c->spawn_idx; // new surfel j = rec[idx[j]]
    int new_m;
    if (K >= k) {
        new_m = M + (K - k);
        if (new_m > c->cap) { // cannot append: report, keep what fits
            if (tid == 0) atomicOr(c->status, kStatusCapacity);
            new_m = c->cap;
        }
        for (int j = tid; j < K; j += nthr) {
            const int tgt = j < k ? c->holes[k - 1 - j] : M + (j - k);
            if (tgt < c->cap) local[tgt] = rec[idx[j]];
        }
    } else {
        const int r = k - K, cut = M - r;
        new_m = cut;
        for (int j = tid; j < K; j += nthr) local[c->holes[k - 1 - j]] = rec[idx[j]];
        for (int i = tid; i < r; i += nthr) {
            const int tgt = c->holes[r - 1 - i];
            if (tgt >= cut) continue;
            int src = M - 1 - i, rank;
            bool hole;
            while ((hole = is_hole(c, src, rank)) && rank < r) src = M - 1 - (r - 1 - rank);
            local[tgt] = hole ? rec[idx[k - 1 - rank]] : local[src];
        }
    }
    if (tid == 0) c->n_local_next[0] = new_m;
}

// Frame tail in one workgroup: new surfels (ordered), deleted-slot list, order-exact compaction, then
// commit the map size and bump the params cursor.  The phases are separated by a workgroup-scope fence +
// barrier because later phases read what earlier ones (same workgroup) wrote to global memory.
//
// Fast path (S <= 8192 seeds, M <= 262144 surfels: every KITTI / VGA frame): the kernel is a chain of dependent trips to
// memory, so everything it needs is fetched in ONE trip -- the two byte planes of the spawn test and this thread's four
// words of the hole bitmap (thread t owns words 4t .. 4t+3: one block scan orders all holes) -- the spawn list and the
// refill targets stay in LDS, and the only second trip is the prepared records themselves.  The rare K < k frame (more
// deleted slots than new surfels: swap-with-last chains) and larger frames / maps take the general path below.
constexpr int kTailFastSeeds = 8192; // (kTailFastWords: dsm_device.h)

__device__ __forceinline__ bool frame_tail_fast(const DeviceCtx *__restrict__ c, int with_compaction, int *s_idx /* [8192] */,
                                                int *s_refill /* [8192] */, int *s_cnt /* [129] */, int *s_wave /* [17] */, int &M_out) {
    const int S = c->n_seed, tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    // ---- one trip: the map size, the spawn planes and the hole bitmap (words beyond the map are dropped once the size
    // is known; the bitmap allocation holds cap / 64 + 1 words)
    unsigned char ok[8], fu[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int sd = q * 1024 + tid;
        ok[q] = c->spawn_ok[sd < S ? sd : 0];
        fu[q] = c->fused_flag[sd < S ? sd : 0];
    }
    unsigned long long mk[4] = {0, 0, 0, 0};
    if (with_compaction) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (4 * tid + q <= c->cap / 64) mk[q] = c->hole_mask[4 * tid + q];
    }
    const int M = c->n_local[0];
    M_out = M;
    if (M > kTailFastWords * 64) return false;
    const int n_word = (M + 63) >> 6;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (4 * tid + q >= n_word) mk[q] = 0;
    // ---- spawn list (seed order = round-major, then thread) into LDS
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const bool spawn = q * 1024 + tid < S && ok[q] && !fu[q];
        if (spawn) mine |= 1u << q;
        const unsigned long long m = __ballot(spawn);
        if (lane == 0) s_cnt[q * 16 + wv] = __popcll(m);
    }
    __syncthreads();
    if (wv == 0) { // exclusive scan of the 128 wave counts by wave 0
        int run = 0;
#pragma unroll
        for (int base = 0; base < 128; base += 64) {
            const int v = s_cnt[base + lane];
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            s_cnt[base + lane] = run + inc - v;
            run += __shfl(inc, 63);
        }
        if (lane == 0) s_cnt[128] = run;
    }
    // ---- holes: one scan over the per-thread counts (thread t's words precede thread t+1's)
    int excl = 0, k = 0;
    if (with_compaction) {
        const int cnt = __popcll(mk[0]) + __popcll(mk[1]) + __popcll(mk[2]) + __popcll(mk[3]);
        k = block_scan_1024(cnt, excl, s_wave); // (its barriers also publish s_cnt)
    } else {
        __syncthreads();
    }
    const int K = s_cnt[128];
    if (with_compaction && K < k) return false; // general path (nothing has been written yet)
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const bool spawn = (mine >> q) & 1u;
        const unsigned long long m = __ballot(spawn);
        if (spawn) s_idx[s_cnt[q * 16 + wv] + rank_below(m)] = q * 1024 + tid;
    }
    if (with_compaction) { // new surfel j goes to hole D[k-1-j] (SM.cpp:1087-1102): the hole of rank o takes j = k-1-o
        int o = excl;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned long long m = mk[q];
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                s_refill[k - 1 - o] = (4 * tid + q) * 64 + b;
                o++;
                m &= m - 1;
            }
        }
    }
    __syncthreads();
    // ---- the prepared records to their places
    const dsm_surfel *rec = c->spawn_rec;
    if (with_compaction) {
        int new_m = M + (K - k);
        if (new_m > c->cap) { // cannot append: report, keep what fits
            if (tid == 0) atomicOr(c->status, kStatusCapacity);
            new_m = c->cap;
        }
        for (int j = tid; j < K; j += 1024) {
            const int tgt = j < k ? s_refill[j] : M + (j - k);
            if (tgt < c->cap) c->local[tgt] = rec[s_idx[j]];
        }
        if (tid == 0) {
            c->n_holes[0] = k;
            c->n_local[0] = new_m;
        }
    } else { // FusionFunctions::fuse_initialize_map hands the new surfels back separately
        for (int j = tid; j < K; j += 1024) c->fresh[j] = rec[s_idx[j]];
    }
    if (tid == 0) {
        c->n_new[0] = K;
        c->cursor[0] = c->cursor[0] + 1;
    }
    return true;
}

// A LARGE map (more than kTailFastWords * 64 surfels) with compaction is worked by all the workgroups of the launch (one
// per kTailChunkWords words of its bitmap, at most kTailMaxBlocks): the hole list is what grows with the map -- a round
// of the scan per 524 288 surfels, each a chain of trips to memory, sixteen of them at 8 M.  k_fuse_surfels has counted the
// holes of every chunk, so every chunk's place in the list is known up front and the chunks are listed independently;
// workgroup 0 orders the new surfels meanwhile.  Whichever workgroup finishes LAST (a ticket taken behind a device-scope
// fence) sees all the lists and does the compaction and the commit.  Nobody waits for anybody.
template <bool BATCH> __global__ __launch_bounds__(1024) void k_frame_tail(const DeviceCtx ctx, const DeviceCtx *__restrict__ batch, int with_compaction) {
    const BlockOf blk = block_of<BATCH>();
    DeviceCtx batch_ctx;
    if (BATCH) batch_ctx = load_ctx(batch + blk.z);
    const DeviceCtx *__restrict__ c = BATCH ? &batch_ctx : &ctx;
    __shared__ int s_cnt[kMaxSeedRounds * 16 + 1];
    __shared__ int s_wave[17];
    __shared__ int s_idx[kTailFastSeeds], s_refill[kTailFastSeeds];
    __shared__ int s_last;
    const int n_blk = gridDim.x;
    int M = 0, K = 0;
    if (blk.x > 0) { // hole lists of a large map
        M = c->hole_chunk[c->n_hole_chunk + 1]; // (0 unless k_fuse_surfels saw a large map)
        if (!with_compaction || M <= kTailFastWords * 64) return;
        const int n_word = (M + 63) >> 6;
        for (int ch = blk.x - 1; ch * kTailChunkWords < n_word; ch += n_blk - 1) {
            const int before = tail_chunk_holes(c, s_wave, ch);
            tail_hole_round(c, s_wave, n_word, ch * kTailChunkWords, before);
        }
    } else {
        if (c->n_seed <= kTailFastSeeds) {
            if (frame_tail_fast(c, with_compaction, s_idx, s_refill, s_cnt, s_wave, M)) return;
            __syncthreads(); // K < k, or a larger map: start over on the general path
        } else {
            M = c->n_local[0];
        }
        K = tail_spawn_list(c, s_cnt);
        const bool large = M > kTailFastWords * 64;
        if (!(large && with_compaction && n_blk > 1)) { // everything here
            int k = 0;
            if (with_compaction) k = tail_hole_scan(c, s_wave, M);
            __threadfence_block(); // the lists were written by this workgroup (same CU): no device-scope write-back needed
            __syncthreads();
            if (with_compaction) {
                tail_compact(c, M, K, k);
            } else { // FusionFunctions::fuse_initialize_map hands the new surfels back separately
                for (int j = threadIdx.x; j < K; j += 1024) c->fresh[j] = c->spawn_rec[c->spawn_idx[j]];
            }
            __threadfence_block();
            __syncthreads();
            if (threadIdx.x == 0) {
                if (with_compaction) c->n_local[0] = c->n_local_next[0];
                c->cursor[0] = c->cursor[0] + 1;
            }
            if (large) // k_fuse_surfels counted, nobody else looks: back to zero for the next frame
                for (int i = threadIdx.x; i < c->n_hole_chunk + 2; i += 1024) c->hole_chunk[i] = 0;
            return;
        }
    }
    // ---- large map: the last workgroup to get here finishes the frame.  The workgroups sit on different XCDs, whose L2s
    // are not coherent with each other: every wave's stores drained, then one agent-scope release (L2 write-back) before
    // the ticket; the last arriver's agent-scope acquire (drops this CU's L1 and the L2's non-local lines) before it
    // reads what the others wrote, with plain vector loads.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (restated: the compiler may drop the fence's own wait)
        const bool last = __hip_atomic_fetch_add(&c->hole_chunk[c->n_hole_chunk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_blk - 1;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_last = last ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    K = load_coherent(c->n_new);
    const int k = tail_chunk_holes(c, s_wave, c->n_hole_chunk);
    if (threadIdx.x == 0) c->n_holes[0] = k;
    tail_compact(c, M, K, k);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        c->n_local[0] = c->n_local_next[0];
        c->cursor[0] = c->cursor[0] + 1;
    }
    for (int i = threadIdx.x; i < c->n_hole_chunk + 2; i += 1024) c->hole_chunk[i] = 0;
}

// ------------------------------------------------------------------------------ map deformation
// SM.cpp:681-824.  Streaming 3x4 transform of position and normal of every surfel; the one stage of the
// product that is purely HBM-bound (88 B per surfel: the 44-byte AoS record is read and rewritten whole).
// A block moves 256 records = 704 16-byte vectors through LDS with fully coalesced loads and stores; a
// lane then owns one record at stride 11 dwords (odd: conflict-free).  group_offsets == nullptr: one
// matrix for all (the reference's active-map case); otherwise record i uses the matrix of its group.
//
// Inactive store (dsm_store_warp): group_on[g] == 0 leaves group g untouched (SM.cpp:691-695: poses whose
// cam_pose already equals loop_pose are skipped), and `cloud` is the XYZI shadow of the store
// (`inactive_pointcloud`): SM.cpp:742 copies [&front, &back) of the warped points, i.e. every point of a
// keyframe's patch except its last one, which keeps its stale position.
__device__ __forceinline__ int warp_group_of(const int32_t *__restrict__ group_offsets, int n_groups, int i) {
    int lo = 0, hi = n_groups; // last g with offsets[g] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (group_offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}
// One matrix for all travels in the kernel-argument segment (`single`, used when mats == nullptr): no staging buffer,
// nothing for the host to wait for between two calls.
__global__ __launch_bounds__(256) void k_warp(dsm_surfel *__restrict__ surfels, const int32_t *__restrict__ n_ptr,
                                              int32_t n_fixed, const float *__restrict__ mats, const WarpMat single,
                                              const int32_t *__restrict__ group_offsets, int32_t n_groups,
                                              const uint8_t *__restrict__ group_on, float4 *__restrict__ cloud) {
    __shared__ __attribute__((aligned(16))) float s_rec[256 * 11];
    const int n = n_ptr ? n_ptr[0] : n_fixed;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    // a wave moves its own 64 records through its quarter of the tile (see k_fuse_surfels: no workgroup barrier)
    float one[16]; // the one matrix of a launch without groups: wave-uniform, read once
#pragma unroll
    for (int q = 0; q < 16; q++) one[q] = group_offsets ? 0.0f : (mats ? mats[q] : single.m[q]);
    float *s_w = s_rec + wv * 64 * 11;
    // without untouched groups to skip, every block is read: its records are fetched one trip ahead (records_issue)
    const bool stream_all = group_on == nullptr;
    const int stride = gridDim.x * 256;
    const int first = blockIdx.x * 256 + wv * 64;
    RecRegs ahead = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (stream_all && first < n) ahead = records_issue<64>(surfels + first, n - first < 64 ? n - first : 64, lane);
    for (int base = first; base < n; base += stride) {
        const int cnt = n - base < 64 ? n - base : 64;
        if (group_on) { // wave-uniform: skip records that only belong to untouched groups
            const int g0 = warp_group_of(group_offsets, n_groups, base), g1 = warp_group_of(group_offsets, n_groups, base + cnt - 1);
            bool any = false;
            for (int g = g0; g <= g1; g++) any |= group_on[g] != 0;
            if (!any) continue;
        }
        if (stream_all) {
            records_land<64>(s_w, ahead, surfels + base, cnt, lane);
        } else {
            const RecRegs now = records_issue<64>(surfels + base, cnt, lane);
            records_land<64>(s_w, now, surfels + base, cnt, lane);
        }
        wave_lds_sync();
        if (stream_all && base + stride < n)
            ahead = records_issue<64>(surfels + base + stride, n - base - stride < 64 ? n - base - stride : 64, lane);
        if (lane < cnt) {
            bool on = true;
            int g = 0;
            float m[16]; // this record's matrix, in registers (the address of a kernel argument would put it in scratch memory)
            if (group_offsets) {
                g = warp_group_of(group_offsets, n_groups, base + lane);
                if (group_on) on = group_on[g] != 0;
#pragma unroll
                for (int q = 0; q < 16; q++) m[q] = mats[16 * g + q];
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++) m[q] = one[q];
            }
            if (on) {
                float *r = s_w + lane * 11;
                const float p[3] = {r[0], r[1], r[2]}, v[3] = {r[3], r[4], r[5]};
                float o[3], w[3];
                xform_point(m, p, o);
                xform_dir(m, v, w);
                r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
                r[3] = w[0]; r[4] = w[1]; r[5] = w[2];
                if (cloud && base + lane != group_offsets[g + 1] - 1) cloud[base + lane] = make_float4(o[0], o[1], o[2], r[7]);
            }
        }
        wave_lds_sync();
        records_from_lds<64>(surfels + base, s_w, cnt, lane);
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------ active-set maintenance
// SM.cpp:1476-1497: bitmap of live surfels attached to keyframe `key` (reuses the hole bitmap arrays: they are
// rebuilt by every frame), then the same one-workgroup scan, then an ordered copy-out that deletes the slots.
__global__ __launch_bounds__(256) void k_mark_key(const DeviceCtx ctx, int key) {
    const DeviceCtx *__restrict__ c = &ctx;
    const int M = c->n_local[0];
    const int n_wave = (M + 63) >> 6, lane = lane_id();
    const int waves_total = (gridDim.x * 256) >> 6;
    for (int wv = (blockIdx.x * 256 + threadIdx.x) >> 6; wv < n_wave; wv += waves_total) {
        const int i = wv * 64 + lane;
        bool hit = false;
        if (i < M) hit = c->local[i].update_times > 0 && c->local[i].last_update == key;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) c->hole_mask[wv] = m;
    }
}
__global__ __launch_bounds__(1024) void k_scan_marks(const DeviceCtx ctx) {
    const DeviceCtx *__restrict__ c = &ctx;
    __shared__ int s_wave[17];
    tail_hole_scan(c, s_wave, c->n_local[0]); // wave_prefix, holes (= marked indices, ascending), n_holes
}
__global__ __launch_bounds__(256) void k_extract_marked(const DeviceCtx ctx, dsm_surfel *__restrict__ out, int cap,
                                                        float4 *__restrict__ cloud_out) {
    const DeviceCtx *__restrict__ c = &ctx;
    const int k = c->n_holes[0];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < k && j < cap; j += gridDim.x * 256) {
        const int i = c->holes[j];
        const dsm_surfel e = c->local[i];
        out[j] = e;
        if (cloud_out) cloud_out[j] = make_float4(e.px, e.py, e.pz, e.color); // SM.cpp:1483-1488
        c->local[i].update_times = 0;
    }
}
// count only (sizing pass of dsm_store_deactivate): k_mark_key + k_scan_marks leave the count in n_holes
__global__ void k_append(const DeviceCtx ctx, int n) {
    const DeviceCtx *__restrict__ c = &ctx;
    if (threadIdx.x == 0 && blockIdx.x == 0) c->n_local[0] = c->n_local[0] + n;
}

// workgroups of `kernel` the current device holds at once (occupancy x CUs), cached per kernel and device
// Workgroups of `kernel` for a grid-stride pass over the map: what the device holds at once, or `per_cu_wanted` per CU if
// that is fewer.  The map-sized kernels are fastest BELOW full occupancy -- every wave keeps a trip of records in flight,
// and past the bytes in flight that cover the memory latency more of them only queue up behind each other (8 M surfels:
// k_warp 146 us with 8 workgroups per CU, 136 with 4, 195 with 2 -- and, once its matrix stopped living in scratch memory,
// 125 with 3, 120.5 with 4, 118 with 5, 120 with 6; k_fuse_surfels 222 us with 5, 203 with 3, 215 with 2; the same order
// at 2 M).
constexpr int kWarpBlocksPerCu = 5, kFuseBlocksPerCu = 3;
template <typename K> static int resident_blocks(K kernel, int block_size, int per_cu_wanted) {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_size, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached[dev] = (per_cu < per_cu_wanted ? per_cu : per_cu_wanted) * cus;
    }
    return cached[dev];
}

hipError_t launch_warp(dsm_surfel *surfels, const int32_t *n_ptr, int n_fixed, const float *d_mats, const float *single16,
                       const int32_t *d_offsets, int n_groups, int n_upper, hipStream_t st, const uint8_t *d_group_on,
                       float4 *d_cloud) {
    int blocks = (n_upper + 255) / 256;
    if (blocks < 1) blocks = 1;
    const int cap = resident_blocks(k_warp, 256, kWarpBlocksPerCu);
    if (blocks > cap) blocks = cap;
    WarpMat one;
    for (int i = 0; i < 16; i++) one.m[i] = single16 ? single16[i] : 0.0f;
    hipLaunchKernelGGL(k_warp, dim3(blocks), dim3(256), 0, st, surfels, n_ptr, n_fixed, d_mats, one, d_offsets, n_groups,
                       d_group_on, d_cloud);
    return hipGetLastError();
}
hipError_t launch_mark(const DeviceCtx &d, int key, int n_upper, hipStream_t st) {
    int blocks = (n_upper + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mark_key, dim3(blocks), dim3(256), 0, st, d, key);
    hipLaunchKernelGGL(k_scan_marks, dim3(1), dim3(1024), 0, st, d);
    return hipGetLastError();
}
hipError_t launch_extract_marked(const DeviceCtx &d, dsm_surfel *out, int cap, float4 *cloud_out, hipStream_t st) {
    hipLaunchKernelGGL(k_extract_marked, dim3(64), dim3(256), 0, st, d, out, cap, cloud_out);
    return hipGetLastError();
}
hipError_t launch_extract(const DeviceCtx &d, int key, dsm_surfel *out, int cap, int n_upper, hipStream_t st) {
    hipError_t e = launch_mark(d, key, n_upper, st);
    if (e != hipSuccess) return e;
    return launch_extract_marked(d, out, cap, nullptr, st);
}
hipError_t launch_append_count(const DeviceCtx &d, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_append, dim3(1), dim3(64), 0, st, d, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ frame upload
// A frame arrives as tightly packed rows; the frame slots are pitched (rows start 64-element aligned).  A 2-D
// hipMemcpy moves such a frame row by row (hundreds of small DMA transfers: 2.7 ms for 1226x370); one 1-D copy
// into a staging buffer plus this repack takes a few microseconds.
template <typename T> __global__ __launch_bounds__(256) void k_repack_rows(T *__restrict__ dst, int pitch, const T *__restrict__ src, int w, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        dst[(int64_t)y * pitch + x] = src[i];
    }
}
hipError_t launch_repack(uint8_t *d_img, float *d_depth, int pitch, const uint8_t *s_img, const float *s_depth, int w, int h, hipStream_t st) {
    const int n = w * h;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (s_img) hipLaunchKernelGGL(k_repack_rows<uint8_t>, dim3(blocks), dim3(256), 0, st, d_img, pitch, s_img, w, n);
    if (s_depth) hipLaunchKernelGGL(k_repack_rows<float>, dim3(blocks), dim3(256), 0, st, d_depth, pitch, s_depth, w, n);
    return hipGetLastError();
}

// Timed replays only: keep the GPU busy for `ticks` of the 100 MHz wall clock while the host enqueues
// the whole frame, so that the events between kernels do not measure host launch latency.
__global__ void k_delay(long long ticks) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < 2000000; i++) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// ------------------------------------------------------------------------------ launcher
const char *const kStageNames[kNumStages] = {
    "init_seeds", "assign_0",  "update_seeds_0", "commit_seeds_0", "assign_1",    "resolve_1",    "update_seeds_1", "commit_seeds_1",
    "assign_2",   "resolve_2", "update_seeds_2", "commit_seeds_2", "seed_points", "seed_fit", "fuse_surfels", "frame_tail",
};

// d_batch != nullptr: the kernels take their context from d_batch[blockIdx.z], z < n_batch (handles of equal geometry
// advancing in lockstep: one launch per kernel for all of them); hc is then any one of them (grid sizes).
hipError_t launch_frame(const DeviceCtx &hc, int map_upper_bound, int tail_map_bound, bool with_compaction,
                        hipStream_t st, hipEvent_t *ev, int stage_lo, int stage_hi, const DeviceCtx *d_batch, int n_batch) {
    int stage = 0;
    hipError_t err = hipSuccess;
    const bool batched = d_batch != nullptr;
    const unsigned nz = batched ? (unsigned)n_batch : 1u;
// launches only the stages whose index (position in kStageNames) lies in [stage_lo, stage_hi]
#define hipLaunchStage(kernel_single, kernel_batch, grid, block, ...)                                        \
    do {                                                                                                     \
        if (stage - 1 >= stage_lo && stage - 1 <= stage_hi) {                                                \
            const dim3 g_((grid).x, (grid).y, nz);                                                           \
            if (batched) hipLaunchKernelGGL(kernel_batch, g_, block, 0, st, hc, d_batch, ##__VA_ARGS__);   \
            else hipLaunchKernelGGL(kernel_single, g_, block, 0, st, hc, d_batch, ##__VA_ARGS__);          \
        }                                                                                                    \
    } while (0)
#define DSM_MARK()                                                                      \
    do {                                                                                \
        if (ev) {                                                                       \
            err = hipEventRecord(ev[stage], st);                                        \
            if (err != hipSuccess) return err;                                          \
        }                                                                               \
        stage++;                                                                        \
    } while (0)
    const int S = hc.n_seed;
    const dim3 g_seed_thr((S + 255) / 256);
    const dim3 g_seed_wave((S + 3) / 4);
    const dim3 g_seed_lane((S + 63) / 64); // one lane per seed
    const dim3 g_seed_rest((S + 63) / 64 + kRestOverBlocks); // packed queue entries, then the seeds with oversized lists
    // Two forms of the per-seed stages (same results): a wave per seed where the launch's latency counts -- one handle, or
    // the few of a frame group -- and a lane per seed (and four pixels per thread in k_assign) where the instructions
    // issued count: launches batched over kLaneBatch handles or more.
    const bool lanes = batched && n_batch >= kLaneBatch;
    const dim3 g_tile1((hc.w + kTileW - 1) / kTileW, (hc.h + AssignTile<1>::kH - 1) / AssignTile<1>::kH);
    const dim3 g_tile4((hc.w + kTileW - 1) / kTileW, (hc.h + AssignTile<4>::kH - 1) / AssignTile<4>::kH);
    const dim3 g_pix4((hc.w + 63) / 64, (hc.h + 3) / 4); // thread per pixel, 64 x 4 per block
    const dim3 g_row8((hc.pitch / 8 + 63) / 64, (hc.h + 3) / 4); // thread per eight pixels of a row, 512 x 4 per block
    if (ev) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, st, 40000LL); // 400 us
    DSM_MARK();
    if (lanes) hipLaunchStage(k_init_seeds_lanes<true>, k_init_seeds_lanes<true>, g_seed_thr, dim3(256));
    else hipLaunchStage(k_init_seeds<false>, k_init_seeds<true>, dim3((S + kInitSeedsPerBlock - 1) / kInitSeedsPerBlock), dim3(256));
    DSM_MARK();
    for (int sweep = 0; sweep < kSweeps; sweep++) {
        if (sweep == 0) {
            if (lanes) hipLaunchStage((k_assign<true, true, 4>), (k_assign<true, true, 4>), g_tile4, dim3(256), sweep);
            else hipLaunchStage((k_assign<true, false, 1>), (k_assign<true, true, 1>), g_tile1, dim3(256), sweep);
            DSM_MARK();
        } else {
            if (lanes) hipLaunchStage((k_assign<false, true, 4>), (k_assign<false, true, 4>), g_tile4, dim3(256), sweep);
            else hipLaunchStage((k_assign<false, false, 1>), (k_assign<false, true, 1>), g_tile1, dim3(256), sweep);
            DSM_MARK();
            hipLaunchStage(k_resolve<false>, k_resolve<true>, dim3(1), dim3(256), sweep);
            hipLaunchStage(k_apply_labels<false>, k_apply_labels<true>, g_row8, dim3(256), sweep); // (part of the resolve stage: the sweep's label image)
            DSM_MARK();
        }
        if (lanes) {
            hipLaunchStage(k_update_seeds<true>, k_update_seeds<true>, g_seed_lane, dim3(64), sweep);
            hipLaunchStage(k_update_seeds_rest<true>, k_update_seeds_rest<true>, g_seed_rest, dim3(64), sweep);
        } else {
            hipLaunchStage(k_update_seeds_wave<false>, k_update_seeds_wave<true>, g_seed_wave, dim3(256), sweep);
        }
        DSM_MARK();
        hipLaunchStage(k_commit_seeds<false>, k_commit_seeds<true>, g_seed_thr, dim3(256), sweep);
        DSM_MARK();
    }
    if (lanes) {
        hipLaunchStage(k_pixel_normals<true>, k_pixel_normals<true>, g_pix4, dim3(256));
        hipLaunchStage(k_seed_stats<true>, k_seed_stats<true>, g_seed_lane, dim3(64));
    } else {
        hipLaunchStage(k_seed_points<false>, k_seed_points<true>, g_seed_wave, dim3(256));
    }
    DSM_MARK();
    hipLaunchStage((k_seed_fit<false, kFitAll>), (k_seed_fit<true, kFitSmall>), dim3((S + kFitSeeds - 1) / kFitSeeds), dim3(64));
    if (batched) hipLaunchStage((k_seed_fit<false, kFitAll>), (k_seed_fit<true, kFitLarge>), dim3(kFitLargeBlocks), dim3(64));
    hipLaunchStage(k_seed_finish<false>, k_seed_finish<true>, g_seed_thr, dim3(256));
    DSM_MARK();
    // grid-stride over the map with no more workgroups than the device holds at once (the ones that start late would run
    // all their trips after the others have finished theirs), and fewer than that: see resident_blocks
    int fuse_blocks = (map_upper_bound + 255) / 256;
    if (fuse_blocks < 1) fuse_blocks = 1;
    const int fuse_cap = batched ? resident_blocks(k_fuse_surfels<true>, 256, kFuseBlocksPerCu) : resident_blocks(k_fuse_surfels<false>, 256, kFuseBlocksPerCu);
    if (fuse_blocks > fuse_cap) fuse_blocks = fuse_cap;
    hipLaunchStage(k_fuse_surfels<false>, k_fuse_surfels<true>, dim3(fuse_blocks), dim3(256));
    DSM_MARK();
    // (a map that may be beyond the tail's one-workgroup path gets a workgroup per chunk of its hole bitmap on top; they
    // leave at once while the map is small, but starting them is not free -- 8 us per launch for eight handles -- so
    // the callers pass 0 until the map can be that large, dsm_api.hip: tail_bound)
    int tail_blocks = 1;
    if (with_compaction && tail_map_bound > kTailFastWords * 64) {
        tail_blocks = 1 + (tail_map_bound / 64 + kTailChunkWords) / kTailChunkWords;
        if (tail_blocks > kTailMaxBlocks) tail_blocks = kTailMaxBlocks;
    }
    hipLaunchStage(k_frame_tail<false>, k_frame_tail<true>, dim3(tail_blocks), dim3(1024), with_compaction ? 1 : 0);
    DSM_MARK();
    if (ev) { // empty interval: what a pair of event records costs by itself
        err = hipEventRecord(ev[stage], st);
        if (err != hipSuccess) return err;
    }
#undef DSM_MARK
#undef hipLaunchStage
    return hipGetLastError();
}

} // namespace dsm
