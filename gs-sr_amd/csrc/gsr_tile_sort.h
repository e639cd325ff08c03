// gsr_tile_sort.h -- order ONE tile's list by (depth bits, gaussian id), all 256 threads of a workgroup together.
//
// The reference's single 64-bit radix sort (3DGS rasterizer_impl.cu:300-308: key = tile << 32 | depth bits, stable => ties by gaussian id) is
// reproduced here per tile, after a sort on the tile id alone has grouped the instances (gsr_binning.hip).  Two callers:
//   * k_blend_fwd (gsr_blend.hip), as its prologue: the sort's dependent loads (ids -> depth keys) hide behind the blending of the
//     other resident tiles instead of being a latency-bound launch of their own (GSR_TILE_SORT=fused, default);
//   * k_tile_depth_sort (gsr_binning.hip), one WAVE per tile for short lists (GSR_TILE_SORT=kernel: the round-3 form, kept for A/B).
// Both leave the same bytes in point_list.
#pragma once
#include "gsr_common.h"

__device__ __forceinline__ uint32_t tds_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t tds_wave_incl_scan(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if ((int)tds_lane() >= d) v += t;
    }
    return v;
}
// inclusive scan over the 256 threads of the workgroup; lds: 17 words
__device__ __forceinline__ uint32_t tds_block_incl_scan(uint32_t v, uint32_t* lds, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = tds_wave_incl_scan(v);
    if (lane == 63) lds[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int w = 0; w < 4; w++) { const uint32_t t = lds[w]; lds[w] = run; run += t; } lds[16] = run; }
    __syncthreads();
    s += lds[wave];
    *total = lds[16];
    __syncthreads();
    return s;
}

// one compare-exchange of the bitonic network on 64-bit (depth << 32 | id) words
__device__ __forceinline__ void tds_cmpx(unsigned long long* s, uint32_t t, uint32_t j, uint32_t k)
{
    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), q = i | j;
    const unsigned long long a = s[i], b = s[q];
    const bool up = (i & k) == 0u;
    if ((a > b) == up) { s[i] = b; s[q] = a; }
}

// stable LSD radix sort of (keys_a, ids_a)[0, n) by key in GLOBAL memory, 8 bits per pass, ping-ponging with (keys_b, ids_b); the four passes
// end in the a arrays.  Slow and correct: only lists too long for the LDS network come here.  hist: LDS [256], cnt: LDS [4][256], lds17: LDS [17].
// any_order: the list does not arrive in ascending id order (one-pass bucket sort on the tile id, gsr_binning.hip): four passes on the id digits
// first, so that the four stable passes on the key leave equal keys in id order, as the reference's single stable sort of an id-ordered emission does.
__device__ inline void tds_global_radix(uint32_t* __restrict__ ids_a, uint32_t* __restrict__ keys_a, uint32_t* __restrict__ ids_b, uint32_t* __restrict__ keys_b,
                                        uint32_t n, uint32_t* hist, uint32_t* cnt, uint32_t* lds17, bool any_order = false)
{
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const uint64_t lt = (1ull << tds_lane()) - 1ull;
    const int id_passes = any_order ? 4 : 0;             // workgroup-uniform; the total stays even: the result ends in the a arrays
    for (int pass = 0; pass < 4 + id_passes; pass++) {
        const uint32_t* ki = (pass & 1) ? keys_b : keys_a; const uint32_t* vi = (pass & 1) ? ids_b : ids_a;
        uint32_t* ko = (pass & 1) ? keys_a : keys_b; uint32_t* vo = (pass & 1) ? ids_a : ids_b;
        const bool on_id = pass < id_passes;
        const uint32_t* di = on_id ? vi : ki;            // the array this pass takes its digit from
        const int shift = 8 * (on_id ? pass : pass - id_passes);
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t e = tid; e < n; e += 256u) atomicAdd(&hist[(di[e] >> shift) & 255u], 1u);
        __syncthreads();
        {
            const uint32_t v = hist[tid];
            uint32_t tot;
            const uint32_t incl = tds_block_incl_scan(v, lds17, &tot);
            hist[tid] = incl - v;                    // start of digit tid's run
        }
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += 256u) {
            cnt[tid] = 0; cnt[256 + tid] = 0; cnt[512 + tid] = 0; cnt[768 + tid] = 0;
            __syncthreads();
            const uint32_t e = c0 + tid;
            const bool valid = e < n;
            const uint32_t key = valid ? ki[e] : 0u, val = valid ? vi[e] : 0u;
            const uint32_t d = ((on_id ? val : key) >> shift) & 255u;
            uint64_t peers = __ballot(valid);
            if (!valid) peers = ~peers;
            for (int b = 0; b < 8; b++) { const bool bit = (d >> b) & 1u; const uint64_t m = __ballot(bit); peers &= bit ? m : ~m; }
            const uint32_t before = (uint32_t)__popcll(peers & lt);
            if (valid && before == 0) cnt[wave * 256 + d] = (uint32_t)__popcll(peers);
            __syncthreads();
            uint32_t pos = 0;
            if (valid) { pos = hist[d] + before; for (uint32_t w = 0; w < wave; w++) pos += cnt[w * 256 + d]; }
            __syncthreads();
            hist[tid] += cnt[tid] + cnt[256 + tid] + cnt[512 + tid] + cnt[768 + tid];
            if (valid) { ko[pos] = key; vo[pos] = val; }
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
    }
}

// Rank by counting, n <= 256 NQ entries, 256 threads: thread t holds entries t, t + 256, ... and counts, over ALL entries (broadcast LDS reads, 8 keys
// per iteration, no dependency between iterations), how many sort in front of each, on the 32-bit depth keys alone.  Entries with EQUAL keys then
// collide on their rank, which a per-rank counter detects, and only then (rare: bit-identical view depths in one tile) the count is redone with the
// id as tie-break.  The ranks are a permutation: each id goes straight to its place.  lds: 3 * 256 NQ + 1 words.  Ends with a barrier.
// (O(n^2) compares: 2 VALU per (entry, key) -- cheaper than a barrier-laden bitonic network up to ~512 entries, which is where the method pipelines'
// lists end: mean 200-225, 4-16 % of the tiles between 257 and ~360, tools/tile_list_lengths.py.)
template <int NQ>
__device__ __forceinline__ void tds_rank_sort_wg(uint32_t* lds, uint32_t* list, uint32_t n, const uint32_t* __restrict__ depth_key)
{
    constexpr uint32_t CAP = 256u * NQ;
    uint32_t* kk = lds;                                     // [CAP] keys, 0xFFFFFFFF behind the list (real keys are positive float bits)
    uint32_t* ii = kk + CAP;                                // [CAP] ids
    uint32_t* fl = ii + CAP;                                // [CAP] how many entries took each rank; fl[CAP]: "some ranks collided"
    const uint32_t t = threadIdx.x;
    uint32_t key[NQ], id[NQ], rank[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t e = t + 256u * q;
        key[q] = 0xFFFFFFFFu; id[q] = 0u; rank[q] = 0u;
        if (e < n) { id[q] = list[e]; key[q] = depth_key[id[q]]; }
        kk[e] = key[q]; ii[e] = id[q]; fl[e] = 0u;
    }
    if (t == 0) fl[CAP] = 0u;
    __syncthreads();
    if ((t & ~63u) < n) {                                   // wave-uniform: a wave without entries only keeps the barriers
        for (uint32_t e = 0; e < n; e += 8u) {
            const uint4 a = *reinterpret_cast<const uint4*>(kk + e), b = *reinterpret_cast<const uint4*>(kk + e + 4);
#pragma unroll
            for (int q = 0; q < NQ; q++)
                rank[q] += (a.x < key[q] ? 1u : 0u) + (a.y < key[q] ? 1u : 0u) + (a.z < key[q] ? 1u : 0u) + (a.w < key[q] ? 1u : 0u)
                         + (b.x < key[q] ? 1u : 0u) + (b.y < key[q] ? 1u : 0u) + (b.z < key[q] ? 1u : 0u) + (b.w < key[q] ? 1u : 0u);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (t + 256u * q < n && atomicAdd(&fl[rank[q]], 1u) != 0u) fl[CAP] = 1u;
    __syncthreads();
    if (fl[CAP] != 0u) {
#pragma unroll
        for (int q = 0; q < NQ; q++) rank[q] = 0u;
        if (t < n)
            for (uint32_t e = 0; e < n; e++) {
                const uint32_t k2 = kk[e], i2 = ii[e];
#pragma unroll
                for (int q = 0; q < NQ; q++) rank[q] += ((k2 < key[q]) | ((k2 == key[q]) & (i2 < id[q]))) ? 1u : 0u;
            }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (t + 256u * q < n) list[rank[q]] = id[q];
    __syncthreads();
}

// Rank by counting INSIDE depth buckets, n <= 256 NQ entries, 256 threads.  The plain rank count above compares every entry with every key of
// the list (n^2 compares: ~10 us of VALU per launch at the headline size, 100 us at lists of 600).  Here the entries are first dealt into up to
// 128 buckets by the top bits of (key - smallest key of the list) -- a counting pass with LDS atomics, a 128-entry scan, a scatter -- and every entry
// then ranks itself among the members of its own bucket only (a few entries; by key, ties by id).  Buckets are in key order, so
// rank = bucket start + rank inside the bucket.  Worst case (all keys in one bucket, e.g. bit-identical depths) it degenerates to the plain count.
// lds: 2 * 256 NQ + 264 words (keys and ids stay in registers until they are dealt into the buckets).  Ends with a barrier.
template <int NQ>
__device__ __forceinline__ void tds_bucket_rank_wg(uint32_t* lds, uint32_t* list, uint32_t n, const uint32_t* __restrict__ depth_key)
{
    constexpr uint32_t CAP = 256u * NQ;
    uint32_t* k2 = lds;                     // [CAP] keys in bucket order
    uint32_t* i2 = k2 + CAP;                // [CAP] ids in bucket order
    uint32_t* bcnt = i2 + CAP;              // [128] entries per bucket, then [129] bucket starts (in place, shifted by the scan)
    uint32_t* bst = bcnt + 128;             // [129]
    uint32_t* mm = bst + 132;               // [2] smallest / largest key
    const uint32_t t = threadIdx.x, lane = t & 63u;
    uint32_t key[NQ], id[NQ], b[NQ], slot[NQ];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t e = t + 256u * q;
        key[q] = 0u; id[q] = 0u;
        if (e < n) { id[q] = list[e]; key[q] = depth_key[id[q]]; lo = min(lo, key[q]); hi = max(hi, key[q]); }
    }
    if (t < 128u) bcnt[t] = 0u;
    if (t == 0u) { mm[0] = 0xFFFFFFFFu; mm[1] = 0u; }
    __syncthreads();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64)); }
    if (lane == 0u && lo <= hi) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
    __syncthreads();
    const uint32_t kmin = mm[0], range = mm[1] - kmin;
    const int shift = range ? max(0, 32 - (int)__builtin_clz(range) - 7) : 0;      // (range >> shift) <= 127
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        b[q] = 0u; slot[q] = 0u;
        if (t + 256u * q < n) { b[q] = (key[q] - kmin) >> shift; slot[q] = atomicAdd(&bcnt[b[q]], 1u); }
    }
    __syncthreads();
    if (t < 64u) {                          // exclusive scan of the 128 bucket counts by one wave, two buckets per lane
        const uint32_t c0 = bcnt[2u * t], c1 = bcnt[2u * t + 1u];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += v; }
        const uint32_t ex = incl - (c0 + c1);
        bst[2u * t] = ex; bst[2u * t + 1u] = ex + c0;
        if (t == 63u) bst[128] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (t + 256u * q < n) { const uint32_t p = bst[b[q]] + slot[q]; k2[p] = key[q]; i2[p] = id[q]; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (t + 256u * q < n) {
            const uint32_t s0 = bst[b[q]], s1 = bst[b[q] + 1u];
            uint32_t rank = s0;
            for (uint32_t j = s0; j < s1; j++) {
                const uint32_t kj = k2[j], ij = i2[j];
                rank += ((kj < key[q]) | ((kj == key[q]) & (ij < id[q]))) ? 1u : 0u;
            }
            list[rank] = id[q];
        }
    __syncthreads();
}

// What the long-list fallback needs besides the list itself
struct TdsScratch {
    uint32_t* tile_keys;        // [R] tile id of every instance (constant inside a tile's range): lent to the radix fallback and rewritten
    uint32_t* keys;             // [R] free ping-pong half of the binning arena
    uint32_t* ids;              // [R]
};

// The tile's list `list[0, n)` (any order) -> ascending (depth_key[id], id).  Called by ALL 256 threads of the workgroup (n is block-uniform);
// `lds` is LDS_BYTES of workgroup LDS the caller lends (>= 6 KiB; everything in it is overwritten); ends with a barrier, after which the list
// in global memory is visible to every thread of the workgroup.
//   n <= 512: rank by counting (tds_rank_sort_wg).
//   n <= largest power of two of 64-bit words that fits `lds`: bitonic network on (depth << 32 | id).
//   longer: four-pass radix sort in global memory.
template <int LDS_BYTES>
__device__ __forceinline__ void tds_sort_tile_wg(void* lds, uint32_t* list, uint32_t n, uint32_t first, uint32_t tile,
                                                 const uint32_t* __restrict__ depth_key, TdsScratch sc, bool buckets = true, bool any_order = false)
{
    static_assert(LDS_BYTES >= 6148, "tds_sort_tile_wg: at least 3 * 512 + 1 words of LDS");
    constexpr uint32_t CAPW = (LDS_BYTES / 8 >= 4096) ? 4096u : (LDS_BYTES / 8 >= 2048) ? 2048u : (LDS_BYTES / 8 >= 1024) ? 1024u : 512u;
    const uint32_t t = threadIdx.x;
    if (n <= 1u) { __syncthreads(); return; }
    {
        // rank by counting inside depth buckets, up to 1024 entries (2 words per entry + 264 of the lent LDS; four entries per thread -- eight, for
        // 2048 entries, cost k_blend_fwd 71 VGPRs and its eighth wave per SIMD);
        // buckets == false: the all-pairs count up to 512 entries and the bitonic network above
        constexpr uint32_t BCAP = ((LDS_BYTES / 4 - 264) / 2 >= 1024) ? 1024u : 512u;
        // Lists up to 256 entries keep the all-pairs count: it has three barriers against seven, and at that length the prologue is its chain of
        // dependent loads and barriers, not its compares (measured: blend forward 0.2307 vs 0.2314 ms at mean 169, 0.1018 vs 0.0979 at mean 56 with
        // buckets for every length).  Above, the buckets replace n^2 compares / the bitonic network: 0.430 vs 0.448 ms at mean 337, 0.665 vs 0.723
        // at mean 562 (tools/ab_tile_rank.sh).
        if (buckets && n > 256u && n <= BCAP) {
            uint32_t* w = reinterpret_cast<uint32_t*>(lds);
            if (n <= 512u) tds_bucket_rank_wg<2>(w, list, n, depth_key);
            else if (n <= 768u) tds_bucket_rank_wg<(BCAP >= 1024u ? 3 : 2)>(w, list, n, depth_key);
            else tds_bucket_rank_wg<(BCAP >= 1024u ? 4 : 2)>(w, list, n, depth_key);
            return;
        }
    }
    if (n <= 512u) {
        if (n <= 256u) tds_rank_sort_wg<1>(reinterpret_cast<uint32_t*>(lds), list, n, depth_key);
        else tds_rank_sort_wg<2>(reinterpret_cast<uint32_t*>(lds), list, n, depth_key);
        return;
    }
    if (n <= CAPW) {
        unsigned long long* s = reinterpret_cast<unsigned long long*>(lds);
        uint32_t m = 1024u;
        while (m < n) m <<= 1;
        for (uint32_t e = t; e < m; e += 256u) {
            unsigned long long w = ~0ull;                    // padding sorts behind every real entry
            if (e < n) { const uint32_t id = list[e]; w = ((unsigned long long)depth_key[id] << 32) | id; }
            s[e] = w;
        }
        // Wave w owns the m/4 consecutive words [w m/4, (w+1) m/4): every stage with j <= m/8 exchanges inside those blocks, so it needs no
        // workgroup barrier -- the DS operations of one wave execute in order -- and only the three stages with j >= m/4 (k = m/2: j = m/4;
        // k = m: j = m/2, m/4) are bracketed by __syncthreads (3 + 3 barriers instead of one per stage: 55 stages at m = 1024, 66 at 2048).
        const uint32_t w = t >> 6, lane = t & 63u, per_wave = m >> 3;      // compare-exchanges per wave and stage
        bool prev_global = true;                                            // the fill above was not wave-local
        for (uint32_t k = 2u; k <= m; k <<= 1)
            for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                const bool global = j > per_wave;
                if (global || prev_global) __syncthreads();
                else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
                for (uint32_t q = w * per_wave + lane; q < (w + 1u) * per_wave; q += 64u) tds_cmpx(s, q, j, k);
                prev_global = global;
            }
        __syncthreads();
        for (uint32_t e = t; e < n; e += 256u) list[e] = (uint32_t)s[e];
        __syncthreads();
        return;
    }
    {
        // (keys, ids) = (sc.keys, list) <-> (sc.tile_keys, sc.ids); the tile's tile_keys are rewritten afterwards (constant = tile id)
        uint32_t* w32 = reinterpret_cast<uint32_t*>(lds);
        uint32_t *hist = w32, *cnt = w32 + 256, *lds17 = w32 + 256 + 1024;
        for (uint32_t e = t; e < n; e += 256u) sc.keys[first + e] = depth_key[list[e]];
        __threadfence_block();
        __syncthreads();
        tds_global_radix(list, sc.keys + first, sc.ids + first, sc.tile_keys + first, n, hist, cnt, lds17, any_order);
        for (uint32_t e = t; e < n; e += 256u) sc.tile_keys[first + e] = tile;
        __threadfence_block();
        __syncthreads();
    }
}
