// dsm_k_common.h -- what the frame kernels share: wave-level helpers, the context of a batched launch, plane addressing,
// the label planes.  Included by dsm_kernels.hip (one translation unit; see its head for the map of kernels).
#pragma once
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

// debug: record the shader clock of phase `ph` of seed s in per-seed kernel `kid` (lane 0 only).  The shipped library is
// built without them (kWaveStamps false: every call folds away, and dsm_create refuses DSM_FLAG_WAVE_STAMPS);
// tools/wave_stamps.py builds its own instrumented copy with -DDSM_WAVE_STAMPS=1.
__device__ __forceinline__ void stamp(const DeviceCtx *c, int kid, int s, int ph, int lane) {
    if (kWaveStamps && c->stamps && lane == 0) c->stamps[((int64_t)kid * c->n_seed + s) * 8 + ph] = clock64();
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
    DSM_G(grp_dirty);
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


} // namespace dsm
