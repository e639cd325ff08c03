// gsr_binning.hip -- integer stages between preprocess and blend (all HBM-bound, bit-exact vs the oracle).
//
// The reference (3DGS rasterizer_impl.cu:277-314) scans tiles_touched, emits one 64-bit (tile<<32 | depth) key per
// tile instance and runs a 45-bit stable LSD radix sort over all R instances (6 passes x 24 B/instance).
// Here the same ORDER -- by tile, then depth bits, ties by gaussian id -- is produced with far less traffic:
//   1. stable radix sort of the P gaussians by depth bits (4 passes over P << R elements),
//   2. prefix sum of tiles_touched in that order, duplicate-with-keys in that order (instances are now globally
//      depth-ordered and carry only a 32-bit tile id),
//   3. stable radix sort of the R instances by tile id only: ceil(log2 T) bits = 13 @1080p -> 2 passes x 16 B.
//   4. tile ranges from key boundaries (identifyTileRanges, rasterizer_impl.cu:116-138).
// Stability of every pass makes the result identical to the reference's single 64-bit sort.
#include "gsr_common.h"
#include "gsr_tile_sort.h"
#include "gsr_tile_cull.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive scan across a 64-lane wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}

// block-wide inclusive scan for up to 1024 threads; returns inclusive value, *total = block sum
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* lds /*>=17 words*/, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t s = wave_incl_scan(v);
    if (lane == 63) lds[wave] = s;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = (lane < nw) ? lds[lane] : 0;
        uint32_t ws = wave_incl_scan(w);
        if (lane < nw) lds[lane] = ws - w;        // exclusive prefix per wave
        if (lane == nw - 1) lds[16] = ws;
    }
    __syncthreads();
    s += lds[wave];
    *total = lds[16];
    __syncthreads();
    return s;
}

// ------------------------------------------------------------------------------------------------ generic scan
// in-place exclusive scan of n words by ONE block (n is small: histograms, block sums)
__global__ void __launch_bounds__(GSR_SCAN_BLOCK) k_scan_small(uint32_t* data, uint32_t n, uint32_t* total_out, uint32_t* host_word, int total_only = 0)
{
    __shared__ uint32_t lds[17];
    const uint32_t chunk = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t b = threadIdx.x * chunk, e = min(n, b + chunk);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += data[i];
    uint32_t tot;
    uint32_t incl = block_incl_scan(sum, lds, &tot);
    uint32_t run = incl - sum;
    if (!total_only)           // total_only: the block sums stay as they are (k_duplicate adds them up itself), only the total is published
        for (uint32_t i = b; i < e; i++) { uint32_t v = data[i]; data[i] = run; run += v; }
    if (threadIdx.x == 0) {
        if (total_out) *total_out = tot;
        if (host_word) __hip_atomic_store(host_word, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // mapped pinned word: no D2H copy command
    }
}

// exclusive scan of every row of a [rows x cols] matrix in place, one block per row (coalesced), row totals to tot[].
// Used on the digit-major histogram hist[d * nblk + blk] of LARGE sorts (more than GSR_SORT_MAX_GROUPS block groups), rows = digits.
__global__ void __launch_bounds__(256) k_scan_rows(uint32_t* __restrict__ data, uint32_t cols, uint32_t* __restrict__ tot)
{
    __shared__ uint32_t lds[17];
    uint32_t* row = data + (size_t)blockIdx.x * cols;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < cols; c0 += 256) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = (i < cols) ? row[i] : 0;
        uint32_t t;
        const uint32_t incl = block_incl_scan(v, lds, &t);
        if (i < cols) row[i] = carry + incl - v;
        carry += t;
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

// ------------------------------------------------------------------------------------------------ radix sort
// Digits of up to 11 bits: NB = histogram bins the kernel is built for (256 for <= 8-bit digits, 2048 for 9..11 bits; the wide form is
// an A/B option of the depth order only, see gsr_launch_depth_order).
// per-block digit histogram, block-major: H[blk * NB + d]; the same counts are added into the histogram of the block's group of
// GSR_SORT_GROUP blocks, GH[(blk / GROUP) * NB + d] (device atomics, zeroed beforehand), so that a scatter block finds the number of
// keys in front of its own with ~ groups + GROUP coalesced row reads instead of a separate scan kernel over the whole matrix.
template <int ITEMS, int NB>
__global__ void __launch_bounds__(GSR_SORT_THREADS) k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ n_dev,
                                                                 int shift, uint32_t mask, uint32_t* __restrict__ H, uint32_t* __restrict__ GH,
                                                                 uint32_t digit_major_nblk)
{
    __shared__ uint32_t h[NB];
    if (n_dev) n = min(n, *n_dev);       // device-side element count (speculative forward): n is then the capacity
    for (uint32_t d = threadIdx.x; d < NB; d += GSR_SORT_THREADS) h[d] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (GSR_SORT_THREADS * ITEMS);
#pragma unroll 4
    for (int it = 0; it < ITEMS; it++) {
        uint32_t i = base + it * GSR_SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (digit_major_nblk) {              // large sort: digit-major matrix for k_scan_rows
        for (uint32_t d = threadIdx.x; d <= mask; d += GSR_SORT_THREADS) H[(size_t)d * digit_major_nblk + blockIdx.x] = h[d];
        return;
    }
    uint32_t* grow = GH + (size_t)(blockIdx.x / GSR_SORT_GROUP) * NB;
    for (uint32_t d = threadIdx.x; d < NB; d += GSR_SORT_THREADS) {
        const uint32_t c = h[d];
        H[(size_t)blockIdx.x * NB + d] = c;
        if (c) atomicAdd(&grow[d], c);
    }
}

// exclusive prefix over (mask + 1) <= NB per-digit values held DPT per thread (digit d = DPT * tid + k); returns through arr[]
template <int NB>
__device__ __forceinline__ void digit_excl_scan(const uint32_t* v, uint32_t* out_excl, uint32_t* lds)
{
    constexpr int DPT = NB / GSR_SORT_THREADS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < DPT; k++) sum += v[k];
    uint32_t tot;
    const uint32_t incl = block_incl_scan(sum, lds, &tot);
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < DPT; k++) { out_excl[k] = run; run += v[k]; }
}

// stable scatter.  Order inside a block is (wave, item, lane): wave w owns 64*ITEMS consecutive keys.
template <int ITEMS, int NB>
__global__ void __launch_bounds__(GSR_SORT_THREADS) k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                    uint32_t n, const uint32_t* __restrict__ n_dev, int shift, int bits,
                                                                    const uint32_t* __restrict__ H, const uint32_t* __restrict__ GH, uint32_t ngroups,
                                                                    uint32_t* __restrict__ GH_next, uint32_t digit_major_nblk)
{
    constexpr int DPT = NB / GSR_SORT_THREADS;
    if (n_dev) n = min(n, *n_dev);
    __shared__ uint32_t cnt[4][NB];
    __shared__ uint32_t gbase[NB], lbase[NB];
    __shared__ uint32_t skey[(GSR_SORT_THREADS * ITEMS)], sval[(GSR_SORT_THREADS * ITEMS)];
    __shared__ uint32_t lds[17];
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // The block's keys (and, where the registers allow, values) are requested before the histogram rows below: the two sets of loads are
    // independent, and a block pays their round trips once instead of one after the other.
    constexpr bool EARLY_VALS = ITEMS <= 4;
    const uint32_t base = blockIdx.x * (GSR_SORT_THREADS * ITEMS) + wave * (GSR_WAVE * ITEMS);
    uint32_t key[ITEMS], rank[ITEMS], val[EARLY_VALS ? ITEMS : 1];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t i = base + it * GSR_WAVE + lane;
        key[it] = i < n ? keys_in[i] : 0xFFFFFFFFu;
        if (EARLY_VALS) val[it] = (vals_in && i < n) ? vals_in[i] : i;
    }
    for (int i = threadIdx.x; i < 4 * NB; i += GSR_SORT_THREADS) (&cnt[0][0])[i] = 0;
    {   // global base of digit d for this block = (#keys with a smaller digit) + (#keys with digit d in earlier blocks): the group
        // histograms of the earlier groups + the block histograms of the earlier blocks of this group (coalesced NB-word rows)
        const uint32_t grp = blockIdx.x / GSR_SORT_GROUP;
        uint32_t t[DPT], before[DPT], ex[DPT];
#pragma unroll
        for (int k = 0; k < DPT; k++) { t[k] = 0; before[k] = 0; }
        if (digit_major_nblk) {          // large sort: k_scan_rows left the exclusive row prefix in H and the digit totals in GH
#pragma unroll
            for (int k = 0; k < DPT; k++) {
                const uint32_t d = DPT * threadIdx.x + k;
                if (d <= mask) { t[k] = GH[d]; before[k] = H[(size_t)d * digit_major_nblk + blockIdx.x]; }
            }
            ngroups = 0;
        }
        // rows are requested GB at a time (independent loads in flight): a rolled loop would pay one L2 round trip per row.  With one digit per
        // thread the block histograms of this group's earlier blocks are requested first and summed last, under the group rows' round trips.
        constexpr int GB = DPT == 1 ? 16 : 8;
        uint32_t ch[DPT == 1 ? GSR_SORT_GROUP - 1 : 1];
        const uint32_t b0 = grp * GSR_SORT_GROUP;
        if (DPT == 1 && !digit_major_nblk) {
#pragma unroll
            for (int u = 0; u < GSR_SORT_GROUP - 1; u++) ch[u] = (threadIdx.x <= mask && b0 + u < blockIdx.x) ? H[(size_t)(b0 + u) * NB + threadIdx.x] : 0u;
        }
        for (uint32_t g0 = 0; g0 < ngroups; g0 += GB) {
#pragma unroll
            for (int k = 0; k < DPT; k++) {
                const uint32_t d = DPT * threadIdx.x + k;
                uint32_t c[GB];
#pragma unroll
                for (int u = 0; u < GB; u++) c[u] = (d <= mask && g0 + u < ngroups) ? GH[(size_t)(g0 + u) * NB + d] : 0u;
#pragma unroll
                for (int u = 0; u < GB; u++) { t[k] += c[u]; if (g0 + u < grp) before[k] += c[u]; }
            }
        }
        if (!digit_major_nblk) {
            if (DPT == 1) {
#pragma unroll
                for (int u = 0; u < GSR_SORT_GROUP - 1; u++) before[0] += ch[u];
            } else {
#pragma unroll
                for (int k = 0; k < DPT; k++) {
                    const uint32_t d = DPT * threadIdx.x + k;
                    uint32_t c[GSR_SORT_GROUP - 1];
#pragma unroll
                    for (int u = 0; u < GSR_SORT_GROUP - 1; u++) c[u] = (d <= mask && b0 + u < blockIdx.x) ? H[(size_t)(b0 + u) * NB + d] : 0u;
#pragma unroll
                    for (int u = 0; u < GSR_SORT_GROUP - 1; u++) before[k] += c[u];
                }
            }
        }
        digit_excl_scan<NB>(t, ex, lds);
#pragma unroll
        for (int k = 0; k < DPT; k++) { const uint32_t d = DPT * threadIdx.x + k; if (d <= mask) gbase[d] = ex[k] + before[k]; }
        // the other group-histogram buffer is the next pass's: clear it here (this pass only reads GH)
        if (GH_next && blockIdx.x < ngroups)
            for (uint32_t d = threadIdx.x; d < NB; d += GSR_SORT_THREADS) GH_next[(size_t)blockIdx.x * NB + d] = 0;
    }
    __syncthreads();

    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t i = base + it * GSR_WAVE + lane;
        const bool valid = i < n;
        const uint32_t d = (key[it] >> shift) & mask;
        // lanes of this wave holding the same digit (invalid lanes form their own group and are ignored)
        uint64_t peers = __ballot(valid);
        if (!valid) peers = ~peers;
        for (int b = 0; b < bits; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt);
        uint32_t old = 0;
        if (valid && before == 0) old = atomicAdd(&cnt[wave][d], (uint32_t)__popcll(peers));
        const int leader = __ffsll((unsigned long long)peers) - 1;
        old = __shfl(old, leader, 64);
        rank[it] = old + before;
    }
    __syncthreads();
    // per-digit totals of the block -> exclusive prefix over the digits = position of the digit's run inside the block
    {
        uint32_t tot_d[DPT], ex[DPT];
#pragma unroll
        for (int k = 0; k < DPT; k++) {
            const uint32_t d = DPT * threadIdx.x + k;
            tot_d[k] = 0;
            if (d <= mask) {
#pragma unroll
                for (int w = 0; w < 4; w++) { uint32_t t = cnt[w][d]; cnt[w][d] = tot_d[k]; tot_d[k] += t; }
            }
        }
        digit_excl_scan<NB>(tot_d, ex, lds);
#pragma unroll
        for (int k = 0; k < DPT; k++) { const uint32_t d = DPT * threadIdx.x + k; if (d <= mask) lbase[d] = ex[k]; }
    }
    __syncthreads();
    // stage the block's keys/values in LDS in digit order, then write each digit's run with consecutive lanes on consecutive
    // addresses (a direct scatter puts the 64 lanes of a store on 64 different cache lines)
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t i = base + it * GSR_WAVE + lane;
        if (i < n) {
            const uint32_t d = (key[it] >> shift) & mask;
            const uint32_t lp = lbase[d] + cnt[wave][d] + rank[it];
            skey[lp] = key[it];
            sval[lp] = EARLY_VALS ? val[it] : (vals_in ? vals_in[i] : i);
        }
    }
    __syncthreads();
    const uint32_t nb = min((uint32_t)(GSR_SORT_THREADS * ITEMS), n - min(n, blockIdx.x * (GSR_SORT_THREADS * ITEMS)));
    for (uint32_t q = threadIdx.x; q < nb; q += GSR_SORT_THREADS) {
        const uint32_t k = skey[q];
        const uint32_t d = (k >> shift) & mask;
        const uint32_t pos = gbase[d] + (q - lbase[d]);
        keys_out[pos] = k;
        vals_out[pos] = sval[q];
    }
}

// hist must hold gsr_sort_hist_words(nblk for the 1024-key geometry, 256) words; digits of at most 8 bits
int gsr_radix_sort_pairs(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, const uint32_t* n_dev,
                         int begin_bit, int end_bit, int bits_per_pass, bool identity_vals, uint32_t* hist,
                         bool* result_in_b, hipStream_t s, bool big_blocks, bool group0_zeroed)
{
    // keys per block: 1024 (many blocks: small inputs are latency-bound) or 4096 (longer digit runs -> full-line writes on big inputs).
    // The histogram area is always sized for the 1024-key geometry, the larger upper bound.
    const uint32_t nblk = gsr_sort_blocks(n, big_blocks);
    const uint32_t NB = 256u;
    if (bits_per_pass > 8) bits_per_pass = 8;
    const uint32_t ngroups = gsr_div_up(nblk, GSR_SORT_GROUP);
    uint32_t* GH[2] = { hist, hist + (size_t)NB * ngroups };
    uint32_t* H = hist + 2 * (size_t)NB * ngroups;
    // Every scatter block reads all group rows: O(nblk^2 / 16) words per pass.  Beyond GSR_SORT_MAX_GROUPS groups (~1.5 M keys at 1024 per block,
    // ~6 M at 4096) that costs more than the row-scan kernel it replaces (measured: 17 M instances, +0.23 ms per forward), so large sorts keep
    // the digit-major matrix + k_scan_rows (digit totals in GH[0]).
    const uint32_t dm = (ngroups > GSR_SORT_MAX_GROUPS) ? nblk : 0u;
    if (!dm && !group0_zeroed) if (gsr_memset_async(GH[0], 0, (size_t)NB * ngroups * sizeof(uint32_t), s)) { gsr_set_error("memset group histogram"); return 1; };
    // identity_vals: the first pass generates value i for element i instead of reading vals_a
    uint32_t *kin = keys_a, *vin = identity_vals ? nullptr : vals_a, *kout = keys_b, *vout = vals_b;
    bool in_b = false;
    int pass = 0;
    for (int shift = begin_bit; shift < end_bit; pass++) {
        // balance the digits over the remaining passes
        int remaining = end_bit - shift;
        int passes_left = (remaining + bits_per_pass - 1) / bits_per_pass;
        int bits = (remaining + passes_left - 1) / passes_left;
        uint32_t mask = (1u << bits) - 1u;
        const dim3 g(nblk), b(GSR_SORT_THREADS);
        uint32_t* gh = dm ? GH[0] : GH[pass & 1];
        uint32_t* gh_next = (!dm && passes_left > 1) ? GH[(pass + 1) & 1] : nullptr;
        if (big_blocks) hipLaunchKernelGGL((k_radix_hist<16, 256>), g, b, 0, s, kin, n, n_dev, shift, mask, H, gh, dm);
        else hipLaunchKernelGGL((k_radix_hist<GSR_SORT_ITEMS, 256>), g, b, 0, s, kin, n, n_dev, shift, mask, H, gh, dm);
        if (dm) hipLaunchKernelGGL(k_scan_rows, dim3(mask + 1), dim3(256), 0, s, H, nblk, gh);
        if (big_blocks) hipLaunchKernelGGL((k_radix_scatter<16, 256>), g, b, 0, s, kin, vin, kout, vout, n, n_dev, shift, bits, H, gh, ngroups, gh_next, dm);
        else hipLaunchKernelGGL((k_radix_scatter<GSR_SORT_ITEMS, 256>), g, b, 0, s, kin, vin, kout, vout, n, n_dev, shift, bits, H, gh, ngroups, gh_next, dm);
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        if (vin == nullptr) { vin = vout; vout = vals_a; }     // first pass generated identity values into vals_b
        else { t = vin; vin = vout; vout = t; }
        in_b = !in_b;
        shift += bits;
    }
    if (result_in_b) *result_in_b = in_b;
    return 0;
}


// (Round 6 built the one-sweep form of this sort -- all four digit histograms in one pass, then four scatter launches whose tiles find the equal digits in front of them
// by decoupled look-back over per-tile state words, tickets for the tile order -- and measured it on 3 M keys: 0.80 ms PER PASS against 0.062 for hist + row scan +
// scatter.  All 733 tiles of 4096 keys are resident at once on 256 CUs, none has an inclusive prefix to offer when the others look back, so the last tile walks ~700
// aggregates at one L2 round trip (1.09 us) each: the chain is as long as the grid is wide.  A kernel boundary is the cheap grid barrier on this part (EXPERIMENTS.md (32),
// (75)): the histogram launch in front of the scatter stays.)
// ------------------------------------------------------------------------------------------------ depth order
// offsets[i] = inclusive prefix of tiles_touched[sorted_idx[i]] (block-local), block_sums[blk] = block total
__global__ void __launch_bounds__(GSR_SCAN_BLOCK) k_offsets_local(const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ tiles_touched,
                                                                  uint32_t P, uint32_t* __restrict__ offsets, uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t lds[17];
    const uint32_t i = blockIdx.x * GSR_SCAN_BLOCK + threadIdx.x;
    uint32_t v = (i < P) ? tiles_touched[sorted_idx ? sorted_idx[i] : i] : 0;      // sorted_idx == nullptr: gaussians in id order
    uint32_t tot;
    uint32_t incl = block_incl_scan(v, lds, &tot);
    if (i < P) offsets[i] = incl;
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// Where the depth order of a tile's list comes from.
//   "tile" (default, round 3): no global depth sort.  Instances are emitted in ID order, the stable tile sort bins them, and every tile's list is put
//           in (depth bits, id) order in LDS -- in k_blend_fwd's prologue (gsr_tile_sort.h, GSR_TILE_SORT=fused) or by k_tile_depth_sort (=kernel) --
//           instead of the eight launches of a 4-pass radix sort over P keys, each of which costs its ~5-10 us latency floor whatever P is
//           (profiles/r03_timeline_surfel.json).  The block-local prefix of tiles_touched is written by the preprocess kernel, k_duplicate adds up
//           the block sums in front of its own and publishes num_rendered: no prefix launch at all in the single-call forwards.
//   "global" (GSR_DEPTH_ORDER=global, rounds 1-2): stable LSD sort of the P gaussians by depth bits first, instances emitted in that order.
// Both give the reference's order: by tile, then depth bits, then gaussian id (3DGS rasterizer_impl.cu:70-111, 300-308).
// Measured with the fused per-tile sort (MI355X, 1080p, surfel, tools/ab_depth_order.sh): depth_order + binning + blend_fwd, tile / global, in ms:
//   P = 600k (337 entries per tile) 0.548 / 0.582;  800k (450) 0.698 / 0.742;  1M (562) 0.878 / 0.889;  1.5M (845) 1.008 / 1.050;
//   2M (1125) 1.197 / 1.165;  3M (1688) 1.317 / 1.381 -- a tie above ~1.5M (rank by counting is O(n^2) up to 512 entries, a bitonic network above).
// "auto" (default) takes the per-tile path while P <= 192 tiles' worth of gaussians (1.57M at 1080p).
// The rule without feedback: GSR_DEPTH_ORDER=global|tile forces one (the returned flag says so), otherwise per tile while P <= 192 T.
bool gsr_depth_order_static_rule(int P, int T, bool* forced, int variant)
{
    static int mode = -1;                       // 0 auto, 1 global, 2 tile
    if (mode < 0) { const char* e = getenv("GSR_DEPTH_ORDER"); mode = !e ? 0 : (e[0] == 'g' ? 1 : (e[0] == 't' ? 2 : 0)); }
    if (forced) *forced = mode != 0;
    if (mode == 1) return true;
    if (mode == 2) return false;
    // the crossover sits at a mean tile list of ~900 entries; an EWA gaussian touches ~5.8 tiles of the SURVEY 8d scene, a PLANE one ~5.1, a surfel ~4.6
    // (EWA at P = 1.5 M, mean 1070: global 640 vs per-tile 609 it/s; surfel at 1.5 M, mean 845: 333 vs 343)
    const long long per_tile = variant == GSR_EWA ? 155ll : (variant == GSR_PLANE ? 176ll : 192ll);
    return (long long)P > per_tile * (long long)T;
}

int gsr_launch_depth_order(const gsr_cfg* cfg, GeomView g, uint32_t* host_word_dev, hipStream_t s, bool global_order, bool need_total)
{
    const uint32_t P = (uint32_t)cfg->P;
    if (!global_order) {
        // id order: the block-local prefix and the RAW block sums were written by the preprocess kernel; k_duplicate adds up the sums in front of each
        // workgroup and publishes num_rendered.  A two-stage forward needs the total NOW (the host sizes the binning arena from it): one single-block
        // kernel in its total-only form -- the sums stay raw, so a redo of the binning finds them as the first run did.
        if (need_total)
            hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(GSR_SCAN_BLOCK), 0, s, g.scan_tmp, gsr_div_up(P, 256u), g.counters, host_word_dev, 1);
        return gsr_check_launch("depth_order", s, cfg->debug);
    }
    bool in_b = false;
    // Four 8-bit passes: keys depth_key (A) <-> keys_b end in A, ids: identity -> vals_b -> vals_a -> vals_b -> vals_a (= sorted_idx).
    // the preprocess kernel cleared the first group-histogram buffer (gsr_preprocess.hip: PreParams::zero_ptr)
    if (gsr_radix_sort_pairs(g.depth_key, g.vals_a, g.keys_b, g.vals_b, P, nullptr, 0, 32, 8, true, g.hist, &in_b, s, false, true)) return 1;
    const uint32_t nblk = gsr_div_up(P, GSR_SCAN_BLOCK);
    hipLaunchKernelGGL(k_offsets_local, dim3(nblk), dim3(GSR_SCAN_BLOCK), 0, s, g.sorted_idx, g.tiles_touched, P, g.offsets, g.scan_tmp);
    // the single-block scan of the block sums also publishes num_rendered (device word + mapped pinned host word): k_duplicate adds the
    // block prefix itself, so the third prefix kernel of round 1 (k_offsets_add) is gone
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(GSR_SCAN_BLOCK), 0, s, g.scan_tmp, nblk, g.counters, host_word_dev);
    return gsr_check_launch("depth_order", s, cfg->debug);
}

// ------------------------------------------------------------------------------------------------ duplicate + ranges
// duplicateWithKeys (3DGS rasterizer_impl.cu:70-111) over gaussians in depth order; key = tile id only.
// Wave-cooperative: a wave owns 64 consecutive gaussians of the depth order and emits their instances 64 at a time, lane = instance
// (binary search over the wave's 64 prefix values with shuffles), so the work per lane is even -- one thread per gaussian left a few
// lanes looping over hundreds of tiles -- and the key/value stores are fully coalesced.  Order within a gaussian is (y outer, x inner),
// as in the reference's nested loop.
// CULLV (template argument): >= 0 = the variant whose region test applies, -1 = the reference's whole rect.  With culling, tiles_touched / offsets
// count only the tiles the gaussian's cull record can reach (gsr_preprocess.hip pre_tile_count); the wave walks the UNCULLED rects of its 64
// gaussians, repeats the region test per candidate tile (same function, same words: same answers) and writes the hits compacted, in walk order,
// behind the wave's culled base.
template <int CULLV>
__global__ void __launch_bounds__(256) k_duplicate(uint32_t P, const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ offsets,
                                                   const uint32_t* __restrict__ block_prefix, const uint32_t* __restrict__ tiles_touched,
                                                   const ushort4* __restrict__ rect, const float4* __restrict__ cull, int gx,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t cap,
                                                   uint2* __restrict__ ranges, uint32_t T, uint32_t* __restrict__ zero_ptr, uint32_t zero_n,
                                                   uint32_t* __restrict__ order_valid, uint32_t scan_block,
                                                   uint32_t self_nblk, uint32_t* __restrict__ total_out, uint32_t* __restrict__ host_word)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x);          // position in depth order
    if (i < T) ranges[i] = make_uint2(0xFFFFFFFFu, 0u);                  // the cudaMemset of rasterizer_impl.cu:310, folded in; x > y = no instance (the
                                                                         // last scatter pass merges range candidates with atomicMin / atomicMax)
    for (uint32_t z = i; z < zero_n; z += gridDim.x * blockDim.x) zero_ptr[z] = 0u;      // first group-histogram buffer of the tile sort that follows
    if (i == 0) *order_valid = 0u;                                       // ImgView::tile_order[T]: set again by k_tile_order if it runs for this forward
    const bool v = i < P;
    const uint32_t g = v ? (sorted_idx ? sorted_idx[i] : i) : 0u;
    const uint32_t cnt = v ? tiles_touched[g] : 0u;
    // inclusive prefix of tiles_touched in depth order = block-local prefix (k_offsets_local / the preprocess kernel) + exclusive prefix of the
    // block sums.  self_nblk > 0: block_prefix holds the RAW sums of self_nblk 256-gaussian blocks and every workgroup adds up the ones in front of
    // its own (a few KB of L2 reads) instead of a single-workgroup scan kernel doing it for all (k_scan_small, 4.5 us + a launch); the workgroup of
    // the last block publishes the total = num_rendered (device word for the kernels that follow, mapped pinned word for the host).
    uint32_t bp = 0u;
    if (self_nblk) {
        __shared__ uint32_t s_red[4];
        if (blockIdx.x < self_nblk) {                    // workgroup-uniform
            uint32_t part = 0u;
            for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256u) part += block_prefix[j];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
            if (lane == 0) s_red[threadIdx.x >> 6] = part;
            __syncthreads();
            bp = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
            if (blockIdx.x == self_nblk - 1u && threadIdx.x == 0) {
                const uint32_t tot = bp + block_prefix[blockIdx.x];
                *total_out = tot;
                if (host_word) __hip_atomic_store(host_word, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    } else if (v) bp = block_prefix[i / scan_block];
    const uint32_t incl = v ? offsets[i] + bp : 0u;
    ushort4 r = make_ushort4(0, 0, 1, 1);
    if (cnt) r = rect[g];
    const uint32_t wave_base = __shfl(incl - cnt, 0, 64);
    if constexpr (CULLV >= 0) {
        __shared__ float4 s_cull[2 * 256];
        __shared__ float2 s_slope[256];
        __shared__ ushort4 s_rect[256];
        const uint32_t wbase = threadIdx.x & ~63u;
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = make_float4(0.f, -1.f, 0.f, 0.f);
        if (cnt) { c0 = cull[2 * (size_t)g]; c1 = cull[2 * (size_t)g + 1]; }
        s_cull[2 * threadIdx.x] = c0; s_cull[2 * threadIdx.x + 1] = c1; s_slope[threadIdx.x] = tc_slopes(c0, c1); s_rect[threadIdx.x] = r;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the walk covers the rects of the gaussians that kept at least one tile (a gaussian whose every tile was dropped has nothing to emit)
        const uint32_t area = cnt ? ((uint32_t)r.z - (uint32_t)r.x) * ((uint32_t)r.w - (uint32_t)r.y) : 0u;
        const uint32_t ai = wave_incl_scan(area);
        const uint32_t total = (uint32_t)__shfl((int)ai, 63, 64), excl = ai - area;
        uint32_t run = wave_base;
        for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
            const TcCand c = tc_candidate<CULLV>(k0 + lane, total, excl, s_cull + 2 * wbase, s_slope + wbase, s_rect + wbase);
            const uint64_t hits = __ballot(c.hit);
            const uint32_t off = run + (uint32_t)__popcll(hits & lanemask_lt());
            const uint32_t sg = (uint32_t)__shfl((int)g, (int)c.s, 64);
            if (c.hit && off < cap) {                    // cap == R normally; smaller only when a speculative forward overflowed
                keys[off] = c.ty * (uint32_t)gx + c.tx;
                vals[off] = sg;
            }
            run += (uint32_t)__popcll(hits);
        }
        // count (k_preprocess_*) and emission (here) must agree instance for instance: one wave-level comparison, free
        const uint32_t counted = (uint32_t)__shfl((int)wave_incl_scan(cnt), 63, 64);
        if (lane == 0 && run - wave_base != counted) total_out[GSR_CNT_CULL_MISMATCH] = 1u;
        return;
    }
    // lanes past P carry incl = 0: give them the wave's running total so the prefix stays monotone
    uint32_t last = __shfl(incl, 63, 64);
    {
        const uint64_t vm = __ballot(v);
        if (vm != ~0ull) last = vm ? __shfl(incl, 63 - __builtin_clzll(vm), 64) : wave_base;
    }
    const uint32_t total = last - wave_base;
    const uint32_t excl = v ? (incl - cnt) - wave_base : total;
    const uint32_t w = (uint32_t)r.z - (uint32_t)r.x;
    const float rw = 1.0f / (float)(w ? w : 1u);
    for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
        const uint32_t k = k0 + lane;
        uint32_t s = 0;
#pragma unroll
        for (uint32_t step = 32; step >= 1; step >>= 1) {
            const uint32_t t = s + step;
            const uint32_t e = __shfl(excl, t & 63u, 64);
            if (t < 64u && e <= k) s = t;
        }
        const uint32_t j = k - __shfl(excl, s, 64);
        const uint32_t sw = __shfl(w, s, 64), sx0 = __shfl((uint32_t)r.x, s, 64), sy0 = __shfl((uint32_t)r.y, s, 64), sg = __shfl(g, s, 64);
        const float srw = __shfl(rw, s, 64);
        const uint32_t q = (uint32_t)(((float)j + 0.5f) * srw);           // j / sw: exact for j < 2^20 (margin 0.5/sw >> fp32 error)
        const uint32_t tx = sx0 + (j - q * sw), ty = sy0 + q;
        const uint32_t off = wave_base + k;
        if (k < total && off < cap) {                    // cap == R normally; smaller only when a speculative forward overflowed
            keys[off] = ty * (uint32_t)gx + tx;
            vals[off] = sg;
        }
    }
}

// identifyTileRanges (3DGS rasterizer_impl.cu:116-138)
__global__ void __launch_bounds__(256) k_tile_ranges(uint32_t R, const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ tile_keys, uint2* __restrict__ ranges)
{
    if (n_dev) R = min(R, *n_dev);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const uint32_t cur = tile_keys[i];
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = tile_keys[i - 1];
        if (cur != prev) { ranges[prev].y = i; ranges[cur].x = i; }
    }
    if (i == R - 1) ranges[cur].y = R;
}

// ------------------------------------------------------------------------------------------------ one-pass bucket sort on the tile id
// With the per-tile depth order the instances only have to be GROUPED by tile -- the sort by (depth, id) that follows (gsr_tile_sort.h) takes a
// tile's list in any order -- so the two stable 8-bit radix passes and k_tile_ranges (45 us at the headline size, each kernel near its launch
// floor) are replaced by one counting sort over all tile ids, three kernels:
//   k_tb_hist     per chunk of 4096..16384 instances a histogram over the T tiles in LDS (16-bit counters, two tiles per word: a chunk holds
//                 < 65536 instances) -> one row of the table of per-(chunk, tile) counts; and, per GROUP of 64 tiles, the chunk's instances in the
//                 groups in front of it (a second, small table; atomics on a total per group were tried first: 109 workgroups adding to the same
//                 128 words took 20 us, ~0.2 us per same-address atomic);
//   k_tb_columns  workgroup = one group of 64 tiles: first slot of the group = that column of the small table summed over the chunks, first slot
//                 of a tile = +
//                 the wave scan of the group's column totals (written out as the tiles' ranges: no k_tile_ranges); every cell of the table becomes
//                 the ABSOLUTE first slot of its (chunk, tile) cell;
//   k_tb_scatter  every instance goes to its cell's first slot + its rank inside the cell, taken from the same LDS counters as in k_tb_hist.
// No global atomic per instance (two of them per instance cost more than the radix sort they would replace: DESIGN Appendix A (69)).
// TH threads take a chunk of 16 TH instances, 16 per thread.  Dynamic LDS: Tp / 2 words rounded up to a multiple of 32.
template <int TH>
__global__ void __launch_bounds__(TH) k_tb_hist(const uint32_t* __restrict__ keys, uint32_t R, const uint32_t* __restrict__ n_dev, uint32_t Tp,
                                                uint32_t* __restrict__ tab, uint32_t* __restrict__ group_prefix)
{
    extern __shared__ uint32_t tb_lds[];
    __shared__ uint32_t gs[GSR_TB_GROUPS_MAX];
    uint32_t* h = tb_lds;
    if (n_dev) R = min(R, *n_dev);
    const uint32_t TB = Tp >> 1, TBp = (TB + 31u) & ~31u;
    const uint32_t base = blockIdx.x * (16u * TH) + threadIdx.x;
    uint32_t k[16];
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++) k[j] = (base + j * TH < R) ? keys[base + j * TH] : 0xFFFFFFFFu;      // all loads in flight
    for (uint32_t w = threadIdx.x; w < TBp; w += TH) h[w] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++)
        if (k[j] != 0xFFFFFFFFu) atomicAdd(&h[k[j] >> 1], (k[j] & 1u) ? 65536u : 1u);
    __syncthreads();
    uint2* row = reinterpret_cast<uint2*>(tab + (size_t)blockIdx.x * Tp);
    for (uint32_t w = threadIdx.x; w < TB; w += TH) { const uint32_t v = h[w]; row[w] = make_uint2(v & 0xFFFFu, v >> 16); }
    // group sums: a group of 64 tiles = 32 LDS words; a wave takes two groups per pass (lanes 0-31 / 32-63: consecutive words, no bank conflict)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, ngroups = TBp >> 5;
    for (uint32_t g0 = 2u * wave; g0 < ngroups; g0 += 2u * (TH / 64)) {
        const uint32_t g = g0 + (lane >> 5);
        uint32_t v = 0u;
        if (g < ngroups) { const uint32_t x = h[32u * g + (lane & 31u)]; v = (x & 0xFFFFu) + (x >> 16); }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        if ((lane & 31u) == 0u && g < ngroups) gs[g] = v;
    }
    __syncthreads();
    if (wave == 0u) {                                    // exclusive prefix over the (at most 256) groups: four per lane
        uint32_t c[4], sum = 0u;
#pragma unroll
        for (int q = 0; q < 4; q++) { c[q] = (4u * lane + q < ngroups) ? gs[4u * lane + q] : 0u; sum += c[q]; }
        uint32_t run = wave_incl_scan(sum) - sum;
        uint32_t* out = group_prefix + (size_t)blockIdx.x * GSR_TB_GROUPS_MAX;
#pragma unroll
        for (int q = 0; q < 4; q++) { if (4u * lane + q < ngroups) out[4u * lane + q] = run; run += c[q]; }
    }
}
// Workgroup = one group of 64 tiles x 4 row segments (wave = segment, lane = tile: every load is a coalesced 256-byte piece of a row); the segments'
// sums meet in LDS.  In: the per-(chunk, tile) counts and the chunks' prefixes over the groups.  Out: every cell = the absolute first slot of its (chunk, tile) cell;
// ranges[t] = the tile's slots.
__global__ void __launch_bounds__(256) k_tb_columns(uint32_t* __restrict__ tab, uint32_t rows, uint32_t Tp, uint32_t T, const uint32_t* __restrict__ group_prefix,
                                                    uint2* __restrict__ ranges)
{
    __shared__ uint32_t seg[4][64];
    __shared__ uint32_t s_part[4];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * 64u + lane;
    const uint32_t per = (rows + 3u) >> 2, b0 = min(rows, w * per), b1 = min(rows, b0 + per);
    constexpr int MAXSEG = (GSR_TB_ROWS_MAX + 3) / 4;          // 64 rows per segment at most
    uint32_t c[MAXSEG];
    uint32_t sum = 0u;
    if (t < T) {
#pragma unroll
        for (int q = 0; q < MAXSEG; q++) { c[q] = (b0 + q < b1) ? tab[(size_t)(b0 + q) * Tp + t] : 0u; }
#pragma unroll
        for (int q = 0; q < MAXSEG; q++) sum += c[q];
    }
    // first slot of the group: over the chunks (at most GSR_TB_ROWS_MAX = 256, one per thread), the chunk's instances in the groups in front of this one
    uint32_t part = (threadIdx.x < rows) ? group_prefix[(size_t)threadIdx.x * GSR_TB_GROUPS_MAX + blockIdx.x] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
    seg[w][lane] = sum;
    if (lane == 0u) s_part[w] = part;
    __syncthreads();
    const uint32_t group_base = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const uint32_t total = (seg[0][lane] + seg[1][lane]) + (seg[2][lane] + seg[3][lane]);      // the tile's instances (0 for lanes past T)
    const uint32_t first = group_base + wave_incl_scan(total) - total;
    if (t >= T) return;
    if (w == 0u) ranges[t] = make_uint2(first, first + total);
    uint32_t run = first;
    for (uint32_t q = 0; q < w; q++) run += seg[q][lane];
#pragma unroll
    for (int q = 0; q < MAXSEG; q++)
        if (b0 + q < b1) { tab[(size_t)(b0 + q) * Tp + t] = run; run += c[q]; }
}
// Dynamic LDS: Tp / 2 words.
template <int TH>
__global__ void __launch_bounds__(TH) k_tb_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t R,
                                                   const uint32_t* __restrict__ n_dev, uint32_t Tp, const uint32_t* __restrict__ tab,
                                                   uint32_t* __restrict__ point_list)
{
    extern __shared__ uint32_t tb_lds[];
    uint32_t* h = tb_lds;
    if (n_dev) R = min(R, *n_dev);
    const uint32_t TB = Tp >> 1;
    const uint32_t base = blockIdx.x * (16u * TH) + threadIdx.x;
    const uint32_t* row = tab + (size_t)blockIdx.x * Tp;
    uint32_t k[16], v[16], cell[16];
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++) {
        const bool in = base + j * TH < R;
        k[j] = in ? keys[base + j * TH] : 0xFFFFFFFFu; v[j] = in ? vals[base + j * TH] : 0u;
    }
    for (uint32_t w = threadIdx.x; w < TB; w += TH) h[w] = 0u;
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++) cell[j] = (k[j] != 0xFFFFFFFFu) ? row[k[j]] : 0u;          // gathers in flight
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 16u; j++)
        if (k[j] != 0xFFFFFFFFu) {
            const uint32_t old = atomicAdd(&h[k[j] >> 1], (k[j] & 1u) ? 65536u : 1u);
            point_list[cell[j] + ((k[j] & 1u) ? (old >> 16) : (old & 0xFFFFu))] = v[j];
        }
}

// Launch order of the blend kernels: tiles by DESCENDING list length (eight power-of-two classes, raster order inside a class so that
// neighbouring tiles -- which share most of their splats -- still run close together).  The blend kernels run one workgroup per tile and a
// tile's time is proportional to its list; in raster order the launch ends on whichever long tiles happen to come late, with this order the
// long ones start first and the short ones fill the tail.  Stable counting sort by one workgroup (T is 8160 at 1080p).
__global__ void __launch_bounds__(1024) k_tile_order(const uint2* __restrict__ ranges, uint32_t T, uint32_t* __restrict__ order)
{
    constexpr int NB = 8;
    __shared__ uint32_t cnt[NB * 1024];
    __shared__ uint32_t lds[17];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (T + 1023u) / 1024u, b0 = min(T, tid * per), e0 = min(T, b0 + per);
    auto cls = [](uint2 r) -> int {
        const uint32_t len = r.y > r.x ? r.y - r.x : 0u;
        if (len == 0) return NB - 1;
        const int l2 = 31 - __builtin_clz(len);
        return (NB - 2) - min(max(l2 - 4, 0), NB - 2);          // >= 1024 -> 0, 512.. -> 1, ... 32..63 -> 5, 1..31 -> 6
    };
    uint32_t c[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) c[k] = 0;
    for (uint32_t t = b0; t < e0; t++) {
        const int k = cls(ranges[t]);
#pragma unroll
        for (int q = 0; q < NB; q++) c[q] += (q == k) ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < NB; k++) cnt[k * 1024 + tid] = c[k];
    __syncthreads();
    // exclusive prefix over the class-major array (class 0 of every thread, then class 1, ...): thread t owns entries [8t, 8t + 8)
    uint32_t v[NB], sum = 0;
#pragma unroll
    for (int k = 0; k < NB; k++) { v[k] = cnt[NB * tid + k]; sum += v[k]; }
    uint32_t tot;
    const uint32_t incl = block_incl_scan(sum, lds, &tot);
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < NB; k++) { cnt[NB * tid + k] = run; run += v[k]; }
    __syncthreads();
    uint32_t pos[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) pos[k] = cnt[k * 1024 + tid];
    for (uint32_t t = b0; t < e0; t++) {
        const int k = cls(ranges[t]);
        uint32_t dst = 0;
#pragma unroll
        for (int q = 0; q < NB; q++) if (q == k) { dst = pos[q]; pos[q]++; }
        order[dst] = t;
    }
    if (tid == 0) order[T] = 1u;                  // the order of this forward is in place (k_duplicate cleared the word)
}

// ------------------------------------------------------------------------------------------------ per-tile depth order
// One wave per tile: the tile's list (ids in id order after the stable tile sort) is loaded together with the gaussians' depth bits, sorted as
// 64-bit (depth << 32 | id) words by a bitonic network in the wave's private LDS slice (no barriers: DS operations of a wave execute in order),
// and the ids are written back.  Lists longer than TDS_WAVE_CAP are left to the end of the workgroup's life, where its four waves sort them
// together (up to TDS_WG_CAP, block-level network with barriers); anything longer goes through a stable 4-pass LSD radix sort in global
// memory by the workgroup (scratch: the free ping-pong half of the binning arena) -- slow, correct, and only reached by tiles with > 4096 entries.
#define TDS_RANK_CAP 256u
#define TDS_WAVE_CAP 1024u
#define TDS_WG_CAP 4096u
// rank-by-counting sort of a list of n <= 64 NQ entries by one wave (see k_tile_depth_sort).  sl: the wave's LDS slice (>= 3 * 64 NQ + 8 dwords).
// Main loop on the 32-BIT depth keys only (v_cmp_lt_u32 + add: a 64-bit compare issues at a quarter of that rate and made the first version
// of this kernel compute-bound at 27 us); entries whose keys are equal then collide on their rank, which a per-rank counter in LDS detects,
// and only then (wave-uniform, rare: two gaussians with bit-identical view depth in one tile) the ranks are recomputed with the id as tie-break.
template <int NQ>
__device__ __forceinline__ void tds_rank_sort(unsigned long long* sl64, uint32_t* __restrict__ list, const uint32_t* __restrict__ depth_key, uint32_t n, uint32_t lane)
{
    uint32_t* kk = reinterpret_cast<uint32_t*>(sl64);       // [64 NQ + 8] keys, padded with 0xFFFFFFFF (real keys are positive float bits)
    uint32_t* ii = kk + 64 * NQ + 8;                        // [64 NQ] ids
    uint32_t* fl = ii + 64 * NQ;                            // [64 NQ] how many entries took each rank
    uint32_t key[NQ], id[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t e = lane + 64u * q;
        key[q] = 0xFFFFFFFFu; id[q] = 0u;
        if (e < n) { id[q] = list[e]; key[q] = depth_key[id[q]]; }
        kk[e] = key[q]; ii[e] = id[q]; fl[e] = 0u;
    }
    if (lane < 8u) kk[64u * NQ + lane] = 0xFFFFFFFFu;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t rank[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) rank[q] = 0u;
    for (uint32_t e = 0; e < n; e += 8u) {                   // eight keys per iteration: two independent 16-byte broadcast reads
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = kk[e + u];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) c += (v[u] < key[q]) ? 1u : 0u;
            rank[q] += c;
        }
    }
    bool dup = false;
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (lane + 64u * q < n) dup |= atomicAdd(&fl[rank[q]], 1u) != 0u;
    if (__ballot(dup) != 0ull) {                             // some keys are equal: ties go by id (the list arrives in id order, the ids are distinct)
#pragma unroll
        for (int q = 0; q < NQ; q++) rank[q] = 0u;
        for (uint32_t e = 0; e < n; e++) {
            const uint32_t k2 = kk[e], i2 = ii[e];
#pragma unroll
            for (int q = 0; q < NQ; q++) rank[q] += ((k2 < key[q]) | ((k2 == key[q]) & (i2 < id[q]))) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
        if (lane + 64u * q < n) list[rank[q]] = id[q];
}
__global__ void __launch_bounds__(256) k_tile_depth_sort(const uint2* __restrict__ ranges, uint32_t T, uint32_t cap, const uint32_t* __restrict__ depth_key,
                                                         uint32_t* __restrict__ point_list, uint32_t* __restrict__ tile_keys,
                                                         uint32_t* __restrict__ scratch_keys, uint32_t* __restrict__ scratch_ids, int any_order)
{
    __shared__ unsigned long long s_all[TDS_WG_CAP];        // four wave slices of TDS_WAVE_CAP words, or one block-level buffer
    __shared__ uint32_t s_hist[256], s_cnt[4 * 256], s_lds[17];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + wave;
    uint2 r = make_uint2(0u, 0u);
    if (tile < T) r = ranges[tile];
    r.y = min(r.y, cap);                                     // a speculative forward that overflowed its arena is redone by the caller; stay in bounds
    const uint32_t n = r.y > r.x ? r.y - r.x : 0u;
    if (n > 1u && n <= TDS_RANK_CAP) {
        // short lists (the common case: ~170 entries at 300k gaussians / 1080p): rank by counting.  Every lane holds up to four entries and counts,
        // over ALL entries of the list (broadcast LDS reads, no dependency between iterations), how many sort in front of each -- the words are
        // distinct (the id is part of them), so the ranks are a permutation and each id is written straight to its place.  A bitonic network of the
        // same size is a chain of 36 dependent LDS round trips per wave; this is ~n/2 independent ones.
        unsigned long long* sl = s_all + wave * TDS_WAVE_CAP;
        const uint32_t nq = (n + 63u) >> 6;                  // entries per lane actually in use (wave-uniform): 3 for the typical 170-entry list
        if (nq == 1u) tds_rank_sort<1>(sl, point_list + r.x, depth_key, n, lane);
        else if (nq == 2u) tds_rank_sort<2>(sl, point_list + r.x, depth_key, n, lane);
        else if (nq == 3u) tds_rank_sort<3>(sl, point_list + r.x, depth_key, n, lane);
        else tds_rank_sort<4>(sl, point_list + r.x, depth_key, n, lane);
    } else if (n > TDS_RANK_CAP && n <= TDS_WAVE_CAP) {
        unsigned long long* sl = s_all + wave * TDS_WAVE_CAP;
        uint32_t m = 2u;
        while (m < n) m <<= 1;
        for (uint32_t e = lane; e < m; e += 64u) {
            unsigned long long w = ~0ull;                    // padding sorts behind every real entry
            if (e < n) { const uint32_t id = point_list[r.x + e]; w = ((unsigned long long)depth_key[id] << 32) | id; }
            sl[e] = w;
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = 2u; k <= m; k <<= 1)
            for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                for (uint32_t t = lane; t < (m >> 1); t += 64u) tds_cmpx(sl, t, j, k);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        for (uint32_t e = lane; e < n; e += 64u) point_list[r.x + e] = (uint32_t)sl[e];
    }
    // lists too long for one wave: the workgroup's four waves take them one after the other (block-uniform loop over the four tiles)
    __shared__ uint32_t s_big[4];
    if (lane == 0) s_big[wave] = (n > TDS_WAVE_CAP) ? 1u : 0u;
    __syncthreads();
    for (uint32_t w4 = 0; w4 < 4u; w4++) {
        if (!s_big[w4]) continue;                            // block-uniform
        const uint32_t t4 = blockIdx.x * 4u + w4;
        uint2 rr = ranges[t4];
        rr.y = min(rr.y, cap);
        const uint32_t nn = rr.y - rr.x;
        __syncthreads();
        if (nn <= TDS_WG_CAP) {
            uint32_t m = 2u;
            while (m < nn) m <<= 1;
            for (uint32_t e = threadIdx.x; e < m; e += 256u) {
                unsigned long long w = ~0ull;
                if (e < nn) { const uint32_t id = point_list[rr.x + e]; w = ((unsigned long long)depth_key[id] << 32) | id; }
                s_all[e] = w;
            }
            __syncthreads();
            for (uint32_t k = 2u; k <= m; k <<= 1)
                for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                    for (uint32_t t = threadIdx.x; t < (m >> 1); t += 256u) tds_cmpx(s_all, t, j, k);
                    __syncthreads();
                }
            for (uint32_t e = threadIdx.x; e < nn; e += 256u) point_list[rr.x + e] = (uint32_t)s_all[e];
            __syncthreads();
        } else {
            // (keys, ids) = (scratch_keys, point_list) <-> (tile_keys, scratch_ids); tile_keys of this tile is rewritten afterwards (constant = tile id)
            for (uint32_t e = threadIdx.x; e < nn; e += 256u) scratch_keys[rr.x + e] = depth_key[point_list[rr.x + e]];
            __threadfence_block();
            __syncthreads();
            tds_global_radix(point_list + rr.x, scratch_keys + rr.x, scratch_ids + rr.x, tile_keys + rr.x, nn, s_hist, s_cnt, s_lds, any_order != 0);
            for (uint32_t e = threadIdx.x; e < nn; e += 256u) tile_keys[rr.x + e] = t4;
            __syncthreads();
        }
    }
}

// GSR_TILE_CULL=0: tiles_touched and the instance list cover the whole tile rect of every gaussian, like the reference's (gsr_tile_cull.h)
bool gsr_tile_cull_enabled()
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("GSR_TILE_CULL"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}

// GSR_TILE_BUCKET=0: the two-pass radix sort on the tile id even where the one-pass bucket sort applies (A/B, and the test that both leave the same
// bytes behind the per-tile depth sort).
uint32_t gsr_tile_bucket_chunk(bool global_order, int T, uint32_t cap)
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("GSR_TILE_BUCKET"); on = e ? (atoi(e) != 0) : 1; }
    if (!on || global_order || T > GSR_TB_TILES_MAX) return 0u;
    for (uint32_t chunk = 4096u; chunk <= 16384u; chunk <<= 1)
        if (gsr_div_up(cap > 0u ? cap : 1u, chunk) <= GSR_TB_ROWS_MAX) return chunk;
    return 0u;
}

// GSR_TILE_SORT=fused (default): k_blend_fwd orders its tile's list in its prologue (gsr_tile_sort.h); =kernel: the separate k_tile_depth_sort launch.
bool gsr_tile_sort_is_fused()
{
    static int fused = -1;
    if (fused < 0) { const char* e = getenv("GSR_TILE_SORT"); fused = (e && e[0] == 'k') ? 0 : 1; }
    return fused != 0;
}

static int tile_bits(int T)
{
    int b = 1;
    while ((1 << b) < T) b++;
    return b;
}
int gsr_tile_sort_passes(int T) { return (tile_bits(T) + 7) / 8; }

// R is the exact instance count, or -- when n_dev != nullptr -- the CAPACITY of the binning arena while the exact count
// is read on the device from *n_dev (speculative forward: the host has not seen it yet).
int gsr_launch_binning(const gsr_cfg* cfg, GeomView g, BinView b, ImgView im, uint32_t R, const uint32_t* n_dev, hipStream_t s,
                       bool global_order, uint32_t* host_word_dev)
{
    const int gx = (cfg->W + GSR_TILE - 1) / GSR_TILE, gy = (cfg->H + GSR_TILE - 1) / GSR_TILE;
    const int T = gx * gy;
    if (R == 0) {
        // ranges [T] and the tile order [T + 1] are contiguous in the image arena: all empty, order word "not in place"
        const size_t span = (size_t)(reinterpret_cast<char*>(im.tile_order + T + 1) - reinterpret_cast<char*>(im.ranges));      // incl. the arena's alignment gap
        if (gsr_memset_async(im.ranges, 0, (span + 3) & ~(size_t)3, s)) { gsr_set_error("memset ranges"); return 1; }
        return 0;
    }
    // unsorted instances go to the buffer from which an integral number of passes lands in (tile_keys, point_list)
    const int passes = gsr_tile_sort_passes(T);
    uint32_t *k0 = (passes & 1) ? b.keys_b : b.tile_keys, *v0 = (passes & 1) ? b.vals_b : b.point_list;
    uint32_t *k1 = (passes & 1) ? b.tile_keys : b.keys_b, *v1 = (passes & 1) ? b.point_list : b.vals_b;
    // 4096-key blocks for the tile sort from this many instances on, 1024-key blocks below (the threshold was an environment switch while it was being measured).  Re-measured at the end of
    // round 3, binning stage small / big blocks in ms: R = 0.92 M 0.057 / 0.069; 1.38 M 0.069 / 0.076; 1.60 M 0.0757 / 0.0771;
    // 1.83 M 0.0798 / 0.0785; 2.06 M 0.087 / 0.080; 4.6 M 0.157 / 0.144; 17.5 M 0.587 / 0.495 -- the round-2 threshold of 2^19 was half a size class early.
    const uint32_t big_from0 = 1700000u;
    // R is the CAPACITY of the arena when the count is read on the device (speculative / sync-free forwards: 1.25 x the last count + 16384)
    const uint32_t big_from = n_dev ? big_from0 + big_from0 / 4 : big_from0;
    const uint32_t chunk = gsr_tile_bucket_chunk(global_order, T, b.cap);         // decided on the arena's capacity: the blend forward decides the same way
    const bool bucket = chunk != 0u;
    if (bucket) { k0 = b.keys_b; v0 = b.vals_b; }                               // emission order -> (keys_b, vals_b); BinView::tile_keys is not written
    {
        const dim3 dg(gsr_div_up((uint32_t)max(cfg->P, T), 256)), db(256);
        const uint32_t* sidx = global_order ? (const uint32_t*)g.sorted_idx : (const uint32_t*)nullptr;
        const uint32_t zn = bucket ? 0u : gsr_sort_group_words(R, R >= big_from, 256);
        uint32_t* zp = b.hist;                           // first group histogram of the radix sort
        // per-tile order: 256-gaussian blocks, RAW block sums (every workgroup adds up the ones in front of its own, the last one publishes the
        // total); global order: GSR_SCAN_BLOCK-gaussian blocks, block sums already scanned by k_scan_small
        const uint32_t sblk = global_order ? (uint32_t)GSR_SCAN_BLOCK : 256u;
        const uint32_t snb = global_order ? 0u : gsr_div_up((uint32_t)cfg->P, 256u);
#define GSR_DUP(CV) hipLaunchKernelGGL(k_duplicate<CV>, dg, db, 0, s, (uint32_t)cfg->P, sidx, g.offsets, g.scan_tmp, g.tiles_touched, g.rect, g.cull, gx, k0, v0, R, \
                                       im.ranges, (uint32_t)T, zp, zn, im.tile_order + T, sblk, snb, g.counters, host_word_dev)
        if (!gsr_tile_cull_enabled()) GSR_DUP(-1);
        else if (cfg->variant == GSR_SURFEL) GSR_DUP(GSR_SURFEL);
        else GSR_DUP(GSR_EWA);
#undef GSR_DUP
    }
    if (bucket) {
        const uint32_t rows = gsr_div_up(R, chunk), Tp = ((uint32_t)T + 1u) & ~1u;
        const size_t lds_h = (size_t)(((Tp >> 1) + 31u) & ~31u) * 4u;
        uint32_t* group_prefix = b.tile_tab + gsr_tile_bucket_words(b.cap, (size_t)T);     // [rows][GSR_TB_GROUPS_MAX]
#define GSR_TB(TH) do { \
        hipLaunchKernelGGL((k_tb_hist<TH>), dim3(rows), dim3(TH), lds_h, s, k0, R, n_dev, Tp, b.tile_tab, group_prefix); \
        hipLaunchKernelGGL(k_tb_columns, dim3(gsr_div_up((uint32_t)T, 64u)), dim3(256), 0, s, b.tile_tab, rows, Tp, (uint32_t)T, group_prefix, im.ranges); \
        hipLaunchKernelGGL((k_tb_scatter<TH>), dim3(rows), dim3(TH), lds_h, s, k0, v0, R, n_dev, Tp, b.tile_tab, b.point_list); } while (0)
        if (chunk == 4096u) GSR_TB(256); else if (chunk == 8192u) GSR_TB(512); else GSR_TB(1024);
#undef GSR_TB
        if (!gsr_tile_sort_is_fused())
            hipLaunchKernelGGL(k_tile_depth_sort, dim3(gsr_div_up((uint32_t)T, 4u)), dim3(256), 0, s, im.ranges, (uint32_t)T, R, g.depth_key, b.point_list, b.tile_keys,
                               b.keys_b, b.vals_b, 1);
        if (gsr_tile_order_wanted()) hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, im.ranges, (uint32_t)T, im.tile_order);
        return gsr_check_launch("binning", s, cfg->debug);
    }
    bool in_b = false;
    // (tile ranges written by the last scatter pass instead of k_tile_ranges were measured in round 3 and lost: binning 0.0847 vs 0.0748 ms, DESIGN Appendix A (42))
    if (gsr_radix_sort_pairs(k0, v0, k1, v1, R, n_dev, 0, tile_bits(T), 8, false, b.hist, &in_b, s, R >= big_from, true)) return 1;
    hipLaunchKernelGGL(k_tile_ranges, dim3(gsr_div_up(R, 256)), dim3(256), 0, s, R, n_dev, b.tile_keys, im.ranges);
    if (!global_order && !gsr_tile_sort_is_fused())
        hipLaunchKernelGGL(k_tile_depth_sort, dim3(gsr_div_up((uint32_t)T, 4u)), dim3(256), 0, s, im.ranges, (uint32_t)T, R, g.depth_key, b.point_list, b.tile_keys,
                           b.keys_b, b.vals_b, 0);
    if (gsr_tile_order_wanted()) hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, im.ranges, (uint32_t)T, im.tile_order);
    return gsr_check_launch("binning", s, cfg->debug);
}
