// dsm_k_map.h -- the map-sized kernels: k_fuse_surfels, k_frame_tail (new surfels, hole scan, order-exact compaction),
// k_warp, the active-set kernels, the frame upload repack.  FF.cpp:190-361, SM.cpp:681-824, 1077-1109, 1476-1497.
// Included by dsm_kernels.hip.
#pragma once
#include "dsm_k_common.h"

namespace dsm {

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
        if (__ballot(changed) != 0) { // (stored back only if a surfel of the 64 changed)
            records_from_lds<64>(c->local + base, s_w, cnt, lane);
            if (lane == 0) c->grp_dirty[base >> 6] = 1;
        }
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
            if (tgt < c->cap) { local[tgt] = rec[idx[j]]; c->grp_dirty[tgt >> 6] = 1; }
        }
    } else {
        const int r = k - K, cut = M - r;
        new_m = cut;
        for (int j = tid; j < K; j += nthr) {
            const int tgt = c->holes[k - 1 - j];
            local[tgt] = rec[idx[j]];
            c->grp_dirty[tgt >> 6] = 1;
        }
        for (int i = tid; i < r; i += nthr) {
            const int tgt = c->holes[r - 1 - i];
            if (tgt >= cut) continue;
            int src = M - 1 - i, rank;
            bool hole;
            while ((hole = is_hole(c, src, rank)) && rank < r) src = M - 1 - (r - 1 - rank);
            local[tgt] = hole ? rec[idx[k - 1 - rank]] : local[src];
            c->grp_dirty[tgt >> 6] = 1;
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
            if (tgt < c->cap) { c->local[tgt] = rec[s_idx[j]]; c->grp_dirty[tgt >> 6] = 1; }
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

// ------------------------------------------------------------------------------ delta download (drop-in calls)
// SurfelMap::fuse_map hands its vector to every frame and gets it back (SM.cpp:1066-1073); a frame changes the surfels it
// fuses, deletes, refills and appends -- in a growing map a fraction of the array.  k_delta_pack gathers the 64-record groups
// that k_fuse_surfels / k_frame_tail flagged (DeviceCtx::grp_dirty) into one contiguous block with their group numbers, so that
// one transfer of what changed replaces the transfer of everything; the flags are cleared on the way.  A thread per group
// looks at its flag; a wave then copies its flagged groups one after the other, 176 16-byte vectors each.
__global__ __launch_bounds__(256) void k_delta_pack(const DeviceCtx ctx, dsm_surfel *__restrict__ buf, int32_t *__restrict__ idx,
                                                    int32_t *__restrict__ count, int cap_groups) {
    const DeviceCtx *__restrict__ c = &ctx;
    const int M = c->n_local[0], n_grp = (M + 63) >> 6, lane = lane_id();
    const int n_flag = c->cap / 64 + 1;
    for (int g0 = (blockIdx.x * 256 + (int)threadIdx.x - lane); g0 < n_flag; g0 += gridDim.x * 256) {
        const int g = g0 + lane;
        bool dirty = false;
        if (g < n_flag && c->grp_dirty[g]) {
            c->grp_dirty[g] = 0;
            dirty = g < n_grp; // (a group beyond the map's new end: cut off by the compaction)
        }
        const unsigned long long m = __ballot(dirty);
        if (m == 0) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(count, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (dirty && base + rank_below(m) < cap_groups) idx[base + rank_below(m)] = g;
        int slot = base;
        for (unsigned long long w = m; w; w &= w - 1, slot++) {
            if (slot >= cap_groups) break;
            const int gg = g0 + (__ffsll((long long)w) - 1);
            const int recs = M - gg * 64 < 64 ? M - gg * 64 : 64;
            const int n_vec = (recs * (int)sizeof(dsm_surfel) + 15) >> 4; // (a group starts 16-byte aligned: 64 x 44 = 176 x 16)
            const float4 *src = reinterpret_cast<const float4 *>(c->local + (size_t)gg * 64);
            float4 *dst = reinterpret_cast<float4 *>(buf + (size_t)slot * 64);
            for (int v = lane; v < n_vec; v += 64) dst[v] = src[v];
        }
    }
}
hipError_t launch_delta_pack(const DeviceCtx &d, dsm_surfel *buf, int32_t *idx, int32_t *count, int cap_groups, int n_upper, hipStream_t st) {
    (void)n_upper; // (every flag is looked at: a frame that shrank the map leaves flags beyond its new end, which are dropped)
    int blocks = (d.cap / 64 + 1 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_delta_pack, dim3(blocks), dim3(256), 0, st, d, buf, idx, count, cap_groups);
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
// ... and n of them at once (the asynchronous uploads: frames back to back, tight on the source side, slot after slot on the other)
template <typename T> __global__ __launch_bounds__(256) void k_repack_frames(T *__restrict__ dst, int pitch, int64_t dst_frame, const T *__restrict__ src, int w, int n) {
    const T *s = src + (int64_t)blockIdx.y * n;
    T *d = dst + (int64_t)blockIdx.y * dst_frame;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        d[(int64_t)y * pitch + x] = s[i];
    }
}
hipError_t launch_repack_frames(uint8_t *d_img, float *d_depth, int pitch, int64_t slot_elems, const uint8_t *s_img, const float *s_depth, int w, int h,
                                int frames, hipStream_t st) {
    const int n = w * h;
    int blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (s_img) hipLaunchKernelGGL(k_repack_frames<uint8_t>, dim3(blocks, frames), dim3(256), 0, st, d_img, pitch, slot_elems, s_img, w, n);
    if (s_depth) hipLaunchKernelGGL(k_repack_frames<float>, dim3(blocks, frames), dim3(256), 0, st, d_depth, pitch, slot_elems, s_depth, w, n);
    return hipGetLastError();
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


} // namespace dsm
