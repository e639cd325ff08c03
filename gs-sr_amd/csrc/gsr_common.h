// gsr_common.h -- private declarations shared by the HIP translation units of libgsrast_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gsrast.h"

#define GSR_TILE 16            // tile edge in pixels (BLOCK_X/BLOCK_Y of the reference, config.h:16-17)
#define GSR_WAVE 64
#define GSR_SUB 8              // one wavefront owns an 8x8 sub-tile: 4 waves per 16x16 tile

// record stride (in float4) per variant, see DESIGN.md "data layout in HBM"
#define GSR_REC_EWA 3
#define GSR_REC_PLANE 4
#define GSR_REC_SURFEL 5
// backward accumulator: floats per gaussian row (STRIDE) and the part the kernels write (USED: EWA 9, PLANE 16, SURFEL 18 components, rounded up to
// float4s).  The rows are packed.  Round 4 measured power-of-two strides (EWA 16, SURFEL 32 floats), with which the 16-lane float atomic of the blend
// backward's flush stays inside one 64-byte line instead of straddling two for 50-84 % of the instructions (TCC_ATOMIC 1.85 M / 1.64 M line operations
// for 1.25 M / 0.89 M instructions): blend_bwd 0.4007 vs 0.4014 ms (SURFEL), 0.2793 vs 0.2791 (EWA), iterations 1.5 % / 0.4 % SLOWER (the preprocess
// backward reads and clears wider rows) -- the atomics are not what the kernel waits for.  -DGSR_ACC_SURFEL=32 -DGSR_ACC_EWA=16 rebuilds that variant.
#ifndef GSR_ACC_EWA
#define GSR_ACC_EWA 12
#endif
#define GSR_ACC_PLANE 16
#ifndef GSR_ACC_SURFEL
#define GSR_ACC_SURFEL 20
#endif
#define GSR_ACC_USED_EWA 12
#define GSR_ACC_USED_PLANE 16
#define GSR_ACC_USED_SURFEL 20

// radix sort geometry
#define GSR_SORT_THREADS 256
#define GSR_SORT_ITEMS 4
#define GSR_SORT_BLOCK (GSR_SORT_THREADS * GSR_SORT_ITEMS)
#define GSR_SCAN_BLOCK 1024

static inline __host__ __device__ int gsr_rec_stride(int variant)
{
    return variant == GSR_EWA ? GSR_REC_EWA : (variant == GSR_PLANE ? GSR_REC_PLANE : GSR_REC_SURFEL);
}
static inline __host__ __device__ int gsr_acc_stride(int variant)
{
    return variant == GSR_EWA ? GSR_ACC_EWA : (variant == GSR_PLANE ? GSR_ACC_PLANE : GSR_ACC_SURFEL);
}

static inline __host__ __device__ size_t gsr_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline __host__ __device__ uint32_t gsr_div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---- arena views -------------------------------------------------------------------------------------------
struct GeomView {
    uint32_t* depth_key;      // [P]  bit pattern of view-space depth; 0xFFFFFFFF for culled gaussians
    uint32_t* tiles_touched;  // [P]
    ushort4* rect;            // [P]  tile rect {x0,y0,x1,y1}
    float4* cull;             // [2P] region of alpha >= 1/255.  EWA/PLANE {x,y,A,B},{C,2tau,-,-}; SURFEL {cx,cy,A,B},{C,r2,pix,piy}: ellipse form <= 1 or disc
    float4* rec;              // [P * stride] packed blend record
    uint32_t* clamped;        // [P]  3 bits: SH colour clamped to 0 per channel
    uint32_t* sorted_idx;     // [P]  gaussian ids in (depth, id) order
    uint32_t* offsets;        // [P]  BLOCK-LOCAL inclusive prefix of tiles_touched in depth order (k_offsets_local); the global prefix is offsets[i] + scan_tmp[i / GSR_SCAN_BLOCK]
    uint32_t* keys_b;         // [P]  sort ping-pong
    uint32_t* vals_a;         // [P]
    uint32_t* vals_b;         // [P]
    uint32_t* hist;           // [256 * nblk(P)]
    uint32_t* scan_tmp;       // [>= nblk]
    uint32_t* counters;       // [64] misc device scalars; [0] = num_rendered
    size_t bytes;
};
struct BinView {
    uint32_t* point_list;     // [R] final: gaussian id per instance, sorted by (tile, depth, id)
    uint32_t* tile_keys;      // [R] final: tile id per instance
    uint32_t* keys_b;         // [R]
    uint32_t* vals_b;         // [R]
    uint32_t* hist;           // [128 * nblk(R)]
    uint32_t* scan_tmp;
    uint32_t* tile_tab;       // [(rows + 1) * Tp] one-pass bucket sort on the tile id (gsr_binning.hip): instances per (16384-key chunk, tile), then the tiles' totals
    uint32_t cap;             // the R this view was carved for
    // per (tile, 64-entry batch of its list, 8x8 quadrant): the forward's sub-tile cull ballot, reused by the splat-parallel backward instead of
    // re-testing every entry against the four quadrants.  Word index ((range.x >> 6) + tile + batch) * 4 + quadrant (unique per tile and batch).
    unsigned long long* qmask;
    size_t bytes;
};
struct ImgView {
    float* final_T;           // [N] (SURFEL [3N]: T, M1, M2)
    uint32_t* n_contrib;      // [N] (SURFEL [2N]: last, median)
    uint2* ranges;            // [T]
    uint32_t* tile_order;     // [T + 1] launch order of the blend kernels: tiles by descending list length (gsr_binning.hip k_tile_order); word T: 1 = the
                              // order of THIS forward is in place (k_tile_order ran), 0 = blockIdx -> tile directly
    size_t bytes;
};

GeomView gsr_carve_geom(int variant, int P, void* base);
BinView gsr_carve_bin(int variant, uint32_t R, int W, int H, void* base);
ImgView gsr_carve_img(int variant, int W, int H, void* base);

// ---- error plumbing ----------------------------------------------------------------------------------------
void gsr_set_error(const char* fmt, ...);
int gsr_check_launch(const char* what, hipStream_t s, bool debug);
#define GSR_CHECK(call, what)                                                                \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) { gsr_set_error("%s: %s", what, hipGetErrorString(e_)); return 1; } \
    } while (0)

// ---- stage launchers (each in its own .hip) ------------------------------------------------------------------
// Depth order of a forward (global stable sort of the gaussians by depth, or every tile's list sorted in the blend forward's prologue): decided ONCE
// per forward by its entry point (gsr_decide_depth_order, gsr_api.hip) and handed to every stage launcher as `global_order`.  It also fixes the layout of
// offsets / scan_tmp / sorted_idx in the geom arena, so the preprocess kernel records it IN the arena (GeomView::counters[GSR_CNT_MODE]); a later call
// on that arena (gsr_forward_stage2, also the redo after an overflowed gsr_forward) reads it back and refuses an arena that carries none.
#define GSR_CNT_MODE 1
// counters[GSR_CNT_CULL_MISMATCH]: set by k_duplicate when a wave emitted a different number of instances than its gaussians counted in the preprocess kernel
// (both run gsr_tile_cull.h's test on the same words; a disagreement would shift every later instance of the list).  Cleared by the preprocess kernel, looked
// at by the forward when cfg->debug is set (it synchronises after every stage then; gsr_debug_read does not expose the word).
#define GSR_CNT_CULL_MISMATCH 2
#define GSR_MODE_TILE 0x47530001u
#define GSR_MODE_GLOBAL 0x47530002u
bool gsr_decide_depth_order(const gsr_cfg* cfg);          // static rule (GSR_DEPTH_ORDER, P <= ~192 T) + the long-list feedback; polls the feedback word: call once per forward
int gsr_launch_preprocess(const gsr_cfg* cfg, const gsr_inputs* in, GeomView g, int32_t* radii, hipStream_t s, bool global_order, uint32_t* prefiltered_err = nullptr);
// global order: sorted_idx, offsets, the scanned block sums and counters[0]; per-tile order: nothing unless `need_total` (two-stage forward: the host
// sizes the binning arena from num_rendered) -- otherwise k_duplicate adds up the raw block sums and publishes the total itself
int gsr_launch_depth_order(const gsr_cfg* cfg, GeomView g, uint32_t* host_word_dev, hipStream_t s, bool global_order, bool need_total);
int gsr_launch_binning(const gsr_cfg* cfg, GeomView g, BinView b, ImgView im, uint32_t R, const uint32_t* n_dev, hipStream_t s,
                       bool global_order, uint32_t* host_word_dev = nullptr);
int gsr_launch_blend_fwd(const gsr_cfg* cfg, const gsr_inputs* in, GeomView g, BinView b, ImgView im,
                         const gsr_outputs* out, hipStream_t s, bool global_order, uint32_t* status_dev = nullptr, uint32_t status_cap = 0u);
int gsr_launch_blend_bwd(const gsr_cfg* cfg, const gsr_inputs* in, GeomView g, BinView b, ImgView im,
                         const gsr_out_grads* og, float* acc, hipStream_t s);
bool gsr_blend_bwd_is_sp();            // GSR_BWD=sp (default) | px (gsr_blend.hip)
void gsr_blend_bwd_attach_events(hipEvent_t start, hipEvent_t stop);     // splat-parallel backward only: the next launch of this thread carries them
// hipMemsetAsync that is safe to record into a HIP graph: on ROCm 7.2 a memset NODE replays with a corrupted fill value from the second replay
// on (measured round 3: vis_idx filled with 0x5A5A5A5A instead of 0xFF...), so while `s` is being captured the fill is a kernel; eagerly it is
// the runtime's memset.  nbytes must be a multiple of 4.
bool gsr_depth_order_static_rule(int P, int T, bool* forced, int variant = GSR_SURFEL);   // GSR_DEPTH_ORDER=tile|global|auto and the P <= 192 T rule (gsr_binning.hip)
bool gsr_tile_cull_enabled();         // GSR_TILE_CULL=0|1 (default 1): tile instances culled at emission (gsr_tile_cull.h, gsr_binning.hip)
// One-pass bucket sort on the tile id (per-tile depth order only: it leaves a tile's list in no particular order, which the sort by (depth, id) that
// follows does not mind).  Chunks of 4096 / 8192 / 16384 instances (the smallest that keeps the arena's capacity within GSR_TB_ROWS_MAX chunks: more
// workgroups for the scattered stores), a table of per-(chunk, tile) counts; applies up to GSR_TB_TILES_MAX tiles (the 16-bit counters of two tiles
// share an LDS word) and GSR_TB_ROWS_MAX chunks of 16384, else the two-pass radix sort.  GSR_TILE_BUCKET=0: never.
#define GSR_TB_ROWS_MAX 256u
#define GSR_TB_TILES_MAX 16384
// -> instances per chunk (0: the bucket sort does not apply).  Decided on the ARENA's capacity, so that binning and blend forward agree.
uint32_t gsr_tile_bucket_chunk(bool global_order, int T, uint32_t cap);
static inline size_t gsr_tile_bucket_words(uint32_t cap, size_t T)
{
    if (T > (size_t)GSR_TB_TILES_MAX) return 0;
    const size_t Tp = (T + 1) & ~(size_t)1, rows = ((size_t)cap + 4095) / 4096;      // upper envelope over the three chunk sizes, monotone in cap:
    return ((rows < GSR_TB_ROWS_MAX ? rows : (size_t)GSR_TB_ROWS_MAX) + 1) * Tp;      // gsr_binning_capacity inverts gsr_binning_bytes by bisection
}
#define GSR_TB_GROUPS_MAX (GSR_TB_TILES_MAX / 64)       // behind the table (BinView::tile_tab + gsr_tile_bucket_words(cap, T)): per chunk, the exclusive prefix of its
                                                        // instances over the groups of 64 tiles, [GSR_TB_ROWS_MAX][GSR_TB_GROUPS_MAX]
bool gsr_tile_sort_is_fused();        // GSR_TILE_SORT=fused|kernel: who orders a tile's list by depth when the depth order is per tile (gsr_binning.hip)
bool gsr_tile_order_wanted();         // on while recent forwards reported long tile lists (gsr_api.hip); once per forward
const uint32_t* gsr_static_tile_map(int gx, int gy, hipStream_t s);     // device [gx*gy] blockIdx -> tile, block-cyclic over the XCDs; cached per device and grid; nullptr if unavailable (gsr_api.hip)
uint32_t* gsr_long_list_word();       // device pointer of the per-device feedback word the blend forward reports long lists into, or nullptr (gsr_api.hip)
int gsr_memset_async(void* p, int byte_value, size_t nbytes, hipStream_t s);
int gsr_launch_preprocess_bwd(const gsr_cfg* cfg, const gsr_inputs* in, const int32_t* radii, GeomView g,
                              float* acc, const gsr_in_grads* ig, bool leave_zero, hipStream_t s);

// generic device-wide primitives (gsr_binning.hip)
// Histogram scratch of the sort: [2 group-histogram buffers of NB * groups words][block histograms NB * nblk], groups = ceil(nblk / 16).
// `group0_zeroed`: the caller's preceding kernel already zeroed the first NB * groups words (gsr_sort_group_words) -- otherwise a memset does.
#define GSR_SORT_GROUP 16
#ifndef GSR_SORT_MAX_GROUPS
#define GSR_SORT_MAX_GROUPS 96        // re-measured with 1024-key blocks (tools/ab_sort_groups.sh): binning 0.0678 / 0.0688 / 0.071 ms at 96 / 48 / 160 groups
#endif
static inline uint32_t gsr_sort_blocks(uint32_t n, bool big_blocks) { return gsr_div_up(n, GSR_SORT_THREADS * (big_blocks ? 16u : (uint32_t)GSR_SORT_ITEMS)); }
static inline uint32_t gsr_sort_group_words(uint32_t n, bool big_blocks, uint32_t NB) { return NB * gsr_div_up(gsr_sort_blocks(n, big_blocks), GSR_SORT_GROUP); }
static inline size_t gsr_sort_hist_words(uint32_t nblk_1024, uint32_t NB) { return (size_t)NB * nblk_1024 + 2 * (size_t)NB * (nblk_1024 / GSR_SORT_GROUP + 1); }
int gsr_radix_sort_pairs(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n, const uint32_t* n_dev,
                         int begin_bit, int end_bit, int bits_per_pass, bool identity_vals, uint32_t* hist,
                         bool* result_in_b, hipStream_t s, bool big_blocks = false, bool group0_zeroed = false);
