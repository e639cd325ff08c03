// gsr_tsdf_sparse.hip -- block-sparse TSDF volume: the role of Open3D's ScalableTSDFVolume in GS-SR's mesh extraction
// (gssr/utils/mesh_utils.py:154-178 `extract_mesh_bounded`, extract_mesh_split.py:91-119: voxel_length = depth_trunc / 1024,
// sdf_trunc = 5 voxels, RGB8 colours, depth_scale 1, depth_trunc) without a dense >= 1024^3 grid.
//
// Open3D 0.18 is a pip dependency of the reference (requirements.txt:6), not part of /root/reference: PARITY UNPINNED.  What is
// restated here is its published algorithm (cpp/open3d/pipelines/integration/ScalableTSDFVolume.cpp `Integrate`,
// UniformTSDFVolume.cpp `IntegrateWithDepthToCameraDistanceMultiplier`):
//   * space is cut into volume units of 16^3 voxels, allocated on demand in a hash map keyed by the unit's integer coordinate;
//   * per frame, every `stride`-th depth pixel (default 4) with d > 0 is back-projected to a world point p, and every unit
//     overlapping the box [p - sdf_trunc, p + sdf_trunc] is opened (allocated if new) and integrated ONCE for this frame;
//   * integrating a unit = the uniform-volume rule for each of its 4096 voxels: project the voxel centre, nearest pixel,
//     sdf = (d - z_cam) * |ray|, update where sdf > -trunc with min(1, sdf / trunc), running averages of tsdf and colour, weight += 1.
// The voxel rule is the one k_tsdf_dense (gsr_extra.hip) uses, instruction for instruction, so a sparse volume equals the dense
// volume on every allocated unit bit for bit (tested); the CPU checker used by the tests restates the same algorithm in plain C.
//
// MI355X shape: four launches per frame -- (0) depth + colour planes -> one texel per pixel, colours put on the 0..255 scale on the way (k_ts_texels: the
// voxel pass then needs ONE gather per voxel; 8 bytes -- depth + three colour bytes -- with the reference's uint8 colours, 16 otherwise),
// (1) one thread per sampled pixel inserts <= 8 unit keys into an open-addressing table (64-bit atomicCAS; the winner takes the next pool slot),
// (2) the same pixels stamp their units for this frame and append the newly stamped ones to a work list (a unit whose stamp was still 0 has never
// been written: it is listed as FRESH),
// (3) a persistent grid walks the list (its length is read on the device: the launch does not wait for the host): a workgroup takes a unit, thread =
// four groups of four consecutive z (a wave = 64 consecutive 16-byte groups = 1 KB per load / store instruction in each of the five planes tsdf, weight,
// r, g, b), two groups per pass: 8 projections and texel gathers first, then -- only for a group with an update -- its five 16-byte loads (only if the group
// has ever been written: the unit's 1024-bit `mask`), the running averages and five 16-byte stores.
// ABI 8 (round 6): a unit plane is stored as 4x4x4 bricks of 2x2x4-voxel sectors instead of x-major columns, and nothing is zero-filled any more, not even a
// unit's first frame: the mask says which groups exist (rounds 2-4 zero-filled the pools at creation, round 5 wrote a fresh unit's 80 KB in full).
// Earlier forms, measured and gone: thread = a z-column (64 lines per wave instruction, 2.11 ms on the config-5 frame), thread = every 256th voxel behind a
// host read of the list length (round 4: 1.73 ms; ABI <= 6 entry point gsr_tsdf_sparse_integrate, removed with ABI 8).
// HBM-bound: 40 B per updated voxel; the pool is sized for 288 GB parts.
#include "gsr_common.h"
#include <algorithm>

#define TS_RES 16
#define TS_VOX (TS_RES * TS_RES * TS_RES)
static constexpr unsigned long long TS_EMPTY = ~0ull;

#define TS_UNIT_FLOATS (5 * TS_VOX)      // a unit's record: tsdf plane, weight plane, three colour planes (80 KB)
struct SparseTsdf {
    unsigned long long* keys;      // [cap_hash] packed unit coordinate, TS_EMPTY when free
    int32_t* slot;                 // [cap_hash] pool index of the unit
    int32_t* coord;                // [cap_blocks][3]
    uint32_t* stamp;               // [cap_blocks] last frame that touched the unit
    int32_t* list;                 // [cap_blocks] units touched by the current frame
    int32_t* counters;             // [0] = units allocated, [1] = units in `list`, [2] = pool/hash overflow flag
    unsigned long long* mask;      // [cap_blocks][16] written-group bits (ABI 8)
    float* chunk[GSR_TSDF_MAX_CHUNKS];      // unit records in chunks of doubling size (ABI 8): growing a volume allocates a chunk, it never copies a voxel
    uint32_t chunk0_log2, cap_hash_log2, cap_blocks;
};
// unit b's record: chunk 0 holds units [0, 2^chunk0_log2), chunk c >= 1 the units [2^(chunk0_log2 + c - 1), 2^(chunk0_log2 + c))
__device__ __forceinline__ float* ts_unit(const SparseTsdf& v, int b)
{
    const uint32_t hi = (uint32_t)b >> v.chunk0_log2;
    const int c = hi ? 32 - __clz((int)hi) : 0;
    const uint32_t base = c ? (1u << (v.chunk0_log2 + c - 1)) : 0u;
    return v.chunk[c] + (size_t)((uint32_t)b - base) * TS_UNIT_FLOATS;
}

__host__ __device__ __forceinline__ unsigned long long ts_pack(int x, int y, int z)
{
    return ((unsigned long long)(uint32_t)(x + (1 << 20)) << 42) | ((unsigned long long)(uint32_t)(y + (1 << 20)) << 21) | (unsigned long long)(uint32_t)(z + (1 << 20));
}
__device__ __forceinline__ uint32_t ts_hash(unsigned long long k, uint32_t log2cap) { return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64 - log2cap)); }

// find-or-insert; returns the hash position (slot[] may still be unpublished inside the inserting kernel) or -1 on overflow
__device__ __forceinline__ int ts_insert(const SparseTsdf& v, int x, int y, int z)
{
    const unsigned long long key = ts_pack(x, y, z);
    const uint32_t mask = (1u << v.cap_hash_log2) - 1u;
    uint32_t h = ts_hash(key, v.cap_hash_log2);
    for (uint32_t probe = 0; probe <= mask; probe++, h = (h + 1) & mask) {
        unsigned long long cur = __hip_atomic_load(&v.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return (int)h;
        if (cur == TS_EMPTY) {
            unsigned long long expect = TS_EMPTY;
            if (__hip_atomic_compare_exchange_strong(&v.keys[h], &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const int idx = atomicAdd(&v.counters[0], 1);
                if ((uint32_t)idx >= v.cap_blocks) { v.counters[2] = 1; v.slot[h] = -1; return -1; }
                v.coord[3 * idx] = x; v.coord[3 * idx + 1] = y; v.coord[3 * idx + 2] = z;
                v.slot[h] = idx;
                return (int)h;
            }
            if (expect == key) return (int)h;
        }
    }
    v.counters[2] = 1;
    return -1;
}
__device__ __forceinline__ int ts_find(const SparseTsdf& v, int x, int y, int z)
{
    const unsigned long long key = ts_pack(x, y, z);
    const uint32_t mask = (1u << v.cap_hash_log2) - 1u;
    uint32_t h = ts_hash(key, v.cap_hash_log2);
    for (uint32_t probe = 0; probe <= mask; probe++, h = (h + 1) & mask) {
        const unsigned long long cur = v.keys[h];
        if (cur == key) return v.slot[h];
        if (cur == TS_EMPTY) return -1;
    }
    return -1;
}

struct TouchParams {
    int W, H, stride;
    float fx, fy, cx, cy, rfx, rfy, trunc, dtrunc, unit_len, inv_unit;
    float P[12];       // camera -> world (inverse extrinsic), row-major 3x4
};
__device__ __forceinline__ bool ts_pixel_range(const TouchParams& t, const float* __restrict__ depth, int i, int* lo, int* hi, int* oor)
{
    const int nu = (t.W + t.stride - 1) / t.stride;
    const int u = (i % nu) * t.stride, v = (i / nu) * t.stride;
    if (v >= t.H) return false;
    const float d = depth[(size_t)v * t.W + u];
    if (!(d > 0.f) || d > t.dtrunc) return false;
    const float xc = ((float)u - t.cx) * d * t.rfx, yc = ((float)v - t.cy) * d * t.rfy;
    const float p[3] = { t.P[0] * xc + t.P[1] * yc + t.P[2] * d + t.P[3], t.P[4] * xc + t.P[5] * yc + t.P[6] * d + t.P[7],
                         t.P[8] * xc + t.P[9] * yc + t.P[10] * d + t.P[11] };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = (int)floorf((p[a] - t.trunc) * t.inv_unit);
        hi[a] = (int)floorf((p[a] + t.trunc) * t.inv_unit);
        // outside the 21-bit key range (or a box wider than the 4 units check_vol admits): the sample cannot be stored -> raise the volume's
        // "out of range" flag instead of dropping it silently (the host turns it into an error after the launch)
        if (lo[a] < -(1 << 20) + 1 || hi[a] > (1 << 20) - 2 || hi[a] - lo[a] > 3) { *oor = 1; return false; }
    }
    return true;
}
// ---- the two touch kernels.  A unit of 16 voxels is ~12 sampled pixels wide, so the 64 consecutive samples of a wave name the same few units over and over:
// a lane looks a unit up only if its left neighbour's sample did not name the same one at the same corner of its box (corner c = lo or hi per axis; the
// boxes of neighbouring samples are translates of each other).  That cuts the hash lookups ~10 x and, more important, leaves ~12 stampers per unit
// instead of ~150: the same-address exchanges of round 4's stamp kernel (83 us for 3569 units) are gone without relying on the timing of a plain load,
// so a lane can issue the independent loads of its (<= 8) units together.  Boxes wider than two units per axis (sdf_trunc > one unit) take the plain loops.
__device__ __forceinline__ bool ts_corner(const int* lo, const int* hi, int c, int* x)
{
    x[0] = (c & 1) ? hi[0] : lo[0]; x[1] = (c & 2) ? hi[1] : lo[1]; x[2] = (c & 4) ? hi[2] : lo[2];
    return !(((c & 1) && hi[0] == lo[0]) || ((c & 2) && hi[1] == lo[1]) || ((c & 4) && hi[2] == lo[2]));
}
__device__ __forceinline__ unsigned long long shfl_up1_u64(unsigned long long v)
{
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, 1, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// -> small: every lane of the wave with a sample has a box of <= 2 units per axis (wave-uniform); lead[c]: this lane looks corner c up
__device__ __forceinline__ bool ts_leaders(bool valid, const int* lo, const int* hi, unsigned long long* key, bool* lead)
{
    const bool small = !valid || (hi[0] - lo[0] <= 1 && hi[1] - lo[1] <= 1 && hi[2] - lo[2] <= 1);
    if (__ballot(!small) != 0ull) return false;
    const bool first = (threadIdx.x & 63) == 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        int x[3] = { 0, 0, 0 };
        const bool use = valid && ts_corner(lo, hi, c, x);
        key[c] = use ? ts_pack(x[0], x[1], x[2]) : TS_EMPTY;
        const unsigned long long prev = shfl_up1_u64(key[c]);
        lead[c] = use && (first || prev != key[c]);
    }
    return true;
}
__global__ void __launch_bounds__(256) k_ts_touch_insert(SparseTsdf v, TouchParams t, const float* __restrict__ depth, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    const bool valid = i < n && ts_pixel_range(t, depth, i, lo, hi, &v.counters[3]);
    unsigned long long key[8];
    bool lead[8];
    if (ts_leaders(valid, lo, hi, key, lead)) {
        unsigned long long cur[8];
#pragma unroll
        for (int c = 0; c < 8; c++)
            cur[c] = lead[c] ? __hip_atomic_load(&v.keys[ts_hash(key[c], v.cap_hash_log2)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : key[c];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (!lead[c] || cur[c] == key[c]) continue;      // present at its first probe position: the common case after the first frames
            int x[3];
            (void)ts_corner(lo, hi, c, x);
            (void)ts_insert(v, x[0], x[1], x[2]);
        }
        return;
    }
    if (!valid) return;
    for (int x = lo[0]; x <= hi[0]; x++)
        for (int y = lo[1]; y <= hi[1]; y++)
            for (int z = lo[2]; z <= hi[2]; z++) (void)ts_insert(v, x, y, z);
}
// One same-address atomic per listed unit (`list[atomicAdd(&counters[1], 1)] = ...`) is what round 4's stamp kernel spent its time on once the exchanges were
// out of the way: device-scope atomics on one word retire every 2-10 ns (36 us for 3569 units, 122 us for the 53 580 of a config-5 frame).  A workgroup now
// reserves the list slots of all its winners with ONE global atomic: winners take a rank from an LDS counter, thread 0 adds the total, everybody writes.
__global__ void __launch_bounds__(256) k_ts_touch_stamp(SparseTsdf v, TouchParams t, const float* __restrict__ depth, int n, uint32_t frame)
{
    __shared__ int s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    int scratch_flag = 0;
    const bool valid = i < n && ts_pixel_range(t, depth, i, lo, hi, &scratch_flag);
    unsigned long long key[8];
    bool lead[8];
    int won[8];            // list entries (idx, or ~idx for a unit that has never been written) this thread's exchanges won
    int nwon = 0, rank0 = 0;
    const bool small = ts_leaders(valid, lo, hi, key, lead);      // wave-uniform
    if (small) {
        uint32_t h[8];
        unsigned long long cur[8];
        int idx[8];
        uint32_t st[8];
#pragma unroll
        for (int c = 0; c < 8; c++) { h[c] = ts_hash(key[c], v.cap_hash_log2); cur[c] = lead[c] ? v.keys[h[c]] : 0ull; }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            idx[c] = -1;
            if (!lead[c]) continue;
            if (cur[c] == key[c]) idx[c] = v.slot[h[c]];
            else { int x[3]; (void)ts_corner(lo, hi, c, x); idx[c] = ts_find(v, x[0], x[1], x[2]); }      // displaced by a collision: walk the probe sequence
            if (idx[c] < 0) v.counters[2] = 1;      // a key without a pool slot (left by an earlier refused frame): the unit cannot be integrated -> "capacity exhausted", never a silent drop
        }
#pragma unroll
        for (int c = 0; c < 8; c++) st[c] = idx[c] >= 0 ? __hip_atomic_load(&v.stamp[idx[c]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : frame;
        uint32_t old[8];
#pragma unroll
        for (int c = 0; c < 8; c++) old[c] = (st[c] != frame) ? atomicExch(&v.stamp[idx[c]], frame) : frame;      // all of a lane's exchanges in flight together
#pragma unroll
        for (int c = 0; c < 8; c++) {
            won[c] = 0;
            if (old[c] != frame) { won[nwon] = old[c] ? idx[c] : ~idx[c]; nwon++; }      // the exchange decides who lists the unit.  Old stamp 0 = never written: fresh (~idx)
        }
        if (nwon) rank0 = atomicAdd(&s_cnt, nwon);
    } else if (valid) {      // boxes wider than two units per axis somewhere in the wave (sdf_trunc > one unit): plain loops, one global atomic per listed unit
        for (int x = lo[0]; x <= hi[0]; x++)
            for (int y = lo[1]; y <= hi[1]; y++)
                for (int z = lo[2]; z <= hi[2]; z++) {
                    const int idx = ts_find(v, x, y, z);
                    if (idx < 0) v.counters[2] = 1;
                    if (idx < 0 || __hip_atomic_load(&v.stamp[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == frame) continue;
                    const uint32_t old = atomicExch(&v.stamp[idx], frame);
                    if (old != frame) v.list[atomicAdd(&v.counters[1], 1)] = old ? idx : ~idx;
                }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(&v.counters[1], s_cnt) : 0;
    __syncthreads();
    for (int k = 0; k < nwon; k++) v.list[s_base + rank0 + k] = won[k];
}

struct IntParams {
    int W, H;
    float vl, trunc, dtrunc, fx, fy, cx, cy, rfx, rfy, rtrunc, unit_len;
    float E[12];
};

// ---- texels.  quant 0: colours as given; 1: clamp to [0,1], x 255 (the scale the volume stores); 2: additionally truncated to an integer, the
// uint8 round trip of mesh_utils.py:170 (`(rgb * 255).astype(np.uint8)`, here on the clamped value like the wrapper's torch chain of rounds 2-4).
// quant 0 / 1: one (r, g, b, depth) float4 per pixel.  quant 2 (the reference's setting): the truncated colours ARE bytes, so the texel is 8 bytes --
// (depth, r | g << 8 | b << 16) -- half the gather traffic of the voxel pass and half its texel registers (ABI 8).
template <bool PK>
__global__ void __launch_bounds__(256) k_ts_texels(const float* __restrict__ depth, const float* __restrict__ rgb, void* __restrict__ tex, int N, int quant)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float c[3] = { rgb[i], rgb[(size_t)N + i], rgb[2 * (size_t)N + i] };
    if (quant) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = fminf(fmaxf(c[k], 0.0f), 1.0f) * 255.0f;
            if (quant == 2) c[k] = floorf(c[k]);
        }
    }
    if (PK) reinterpret_cast<float2*>(tex)[i] = make_float2(depth[i], __uint_as_float((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16)));
    else reinterpret_cast<float4*>(tex)[i] = make_float4(c[0], c[1], c[2], depth[i]);
}

// ---- the voxel pass.  The voxel rule of k_tsdf_dense (gsr_extra.hip), the same operations in the same order per voxel.
// Storage of a unit plane (ABI 8): 4 x 4 x 4 bricks of 256 B; inside a brick 64-byte sectors of 2 x 2 x 4 voxels, 128-byte lines of 2 x 4 x 4 -- the shell of voxels a
// frame updates is a surface through the unit, and compact sectors are cut by it far less often than the 1 x 1 x 16 columns of ABI 7 (ts_group below).
// A unit's 4096 voxels are 1024 groups of four consecutive z; thread t takes groups t, t + 256, t + 512, t + 768 (brick x = round): the 64 lanes of a
// wave read and write 64 consecutive 16-byte groups = 4 bricks = 1 KB per instruction in each of the five planes (tsdf, weight, three colour planes).
// mask[unit][16] (64-bit words, bit = group): the groups that have ever been written.  A group whose bit is clear is never read (its first update starts
// from zeros and sets the bit) and a fresh unit writes only the groups the frame observed: nothing zero-fills a unit's 80 KB any more, at creation or later.
__host__ __device__ __forceinline__ int ts_group(int x, int y, int z)
{
    return ((x >> 2) << 8) | ((y >> 2) << 6) | ((z >> 2) << 4) | (((x >> 1) & 1) << 3) | (((y >> 1) & 1) << 2) | ((x & 1) << 1) | (y & 1);
}
// the inverse for thread t of a 256-thread workgroup in round r (group t + 256 r): voxel (x, y, z0 .. z0 + 3)
struct TsLane { int lx, iy, iz0; };
__device__ __forceinline__ TsLane ts_lane(int t)
{
    TsLane l;
    l.lx = ((t >> 3) & 1) * 2 + ((t >> 1) & 1);
    l.iy = ((t >> 6) & 3) * 4 + ((t >> 2) & 1) * 2 + (t & 1);
    l.iz0 = ((t >> 4) & 3) * 4;
    return l;
}
__device__ __forceinline__ unsigned long long ts_uniform64(unsigned long long x)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}
__device__ __forceinline__ void ts_unpack4(const float4 a, float* o) { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; }

#ifndef TS_GRID
#define TS_GRID 2048
#endif
#ifndef TS_WGRAN
#define TS_WGRAN 8      // groups per completed run: 1 (none), 2 (32 B), 4 (64 B), 8 (a 128-byte line)
#endif
__device__ __forceinline__ unsigned long long ts_complete_runs(unsigned long long m)
{
    if (TS_WGRAN >= 2) m |= m >> 1;
    if (TS_WGRAN >= 4) m |= m >> 2;
    if (TS_WGRAN >= 8) m |= m >> 4;
    if (TS_WGRAN == 2) return (m & 0x5555555555555555ull) * 3ull;
    if (TS_WGRAN == 4) return (m & 0x1111111111111111ull) * 15ull;
    if (TS_WGRAN == 8) return (m & 0x0101010101010101ull) * 255ull;
    return m;
}
#define TS_GPT 2      // groups per pass: 8 voxels' projections and texel gathers are in flight before anything of the volume is read
#ifndef TS_WPE
#define TS_WPE 4
#endif
template <bool PK> struct TsTexel;
template <> struct TsTexel<true> { typedef float2 type; static __device__ __forceinline__ float2 zero() { return make_float2(0.f, 0.f); } };
template <> struct TsTexel<false> { typedef float4 type; static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); } };
template <bool PK>
__device__ __forceinline__ void ts_integrate_col(const SparseTsdf& v, const IntParams& p, const void* __restrict__ tex_, int n_fixed)
{
    typedef typename TsTexel<PK>::type texel_t;
    const texel_t* __restrict__ tex = reinterpret_cast<const texel_t*>(tex_);
    const int n = n_fixed >= 0 ? n_fixed : v.counters[1];
    if (v.counters[2] | v.counters[3]) {
        // the touch pass ran out of pool slots / key range: nothing of this frame is integrated (the host grows the pool and runs the frame again, or raises).
        // Units stamped as fresh by this frame go back to "never written".
        for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) { const int e = v.list[k]; if (e < 0) v.stamp[~e] = 0u; }
        return;
    }
    const TsLane L = ts_lane((int)threadIdx.x);
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int e = v.list[k];
        const bool fresh = e < 0;
        const int b = __builtin_amdgcn_readfirstlane(fresh ? ~e : e);
        const float ox = (float)v.coord[3 * b] * p.unit_len, oy = (float)v.coord[3 * b + 1] * p.unit_len, oz = (float)v.coord[3 * b + 2] * p.unit_len;
        const float y = oy + p.vl * ((float)L.iy + 0.5f);
        float* rec = ts_unit(v, b);
        float4* S4 = reinterpret_cast<float4*>(rec);
        float4* W4 = reinterpret_cast<float4*>(rec + TS_VOX);
        float4* C4 = reinterpret_cast<float4*>(rec + 2 * TS_VOX);      // three planes of 4096 floats: [channel][voxel]
        unsigned long long* M = v.mask + (size_t)b * 16;
        unsigned long long had[4];      // wave-uniform: the written-group words of this wave's four rounds (a fresh unit's words hold whatever the pool held)
#pragma unroll
        for (int r = 0; r < 4; r++) had[r] = fresh ? 0ull : ts_uniform64(M[wv + 4 * r]);
#pragma unroll
        for (int pass = 0; pass < 4 / TS_GPT; pass++) {
            texel_t t[4 * TS_GPT];
            float zcs[4 * TS_GPT], rl[4 * TS_GPT];
            uint32_t cand = 0;
#pragma unroll
            for (int q = 0; q < TS_GPT; q++) {
                const int ix = L.lx + 4 * (pass * TS_GPT + q);
                const float x = ox + p.vl * ((float)ix + 0.5f);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int s_ = 4 * q + j;
                    const float z = oz + p.vl * ((float)(L.iz0 + j) + 0.5f);
                    const float xc = p.E[0] * x + p.E[1] * y + p.E[2] * z + p.E[3];
                    const float yc = p.E[4] * x + p.E[5] * y + p.E[6] * z + p.E[7];
                    const float zc = p.E[8] * x + p.E[9] * y + p.E[10] * z + p.E[11];
                    zcs[s_] = zc; rl[s_] = 0.f;
                    t[s_] = TsTexel<PK>::zero();
                    if (!(zc > 0.f)) continue;
                    const float rz = __builtin_amdgcn_rcpf(zc);
                    const float uf = xc * p.fx * rz + p.cx + 0.5f, vf = yc * p.fy * rz + p.cy + 0.5f;
                    if (!(uf >= 0.f && uf < (float)p.W && vf >= 0.f && vf < (float)p.H)) continue;
                    const int u = (int)uf, vv = (int)vf;
                    const float rx = ((float)u - p.cx) * p.rfx, ry = ((float)vv - p.cy) * p.rfy;
                    rl[s_] = __builtin_amdgcn_sqrtf(rx * rx + ry * ry + 1.0f);
                    cand |= 1u << s_;
                    t[s_] = tex[(size_t)vv * p.W + u];
                }
            }
            uint32_t m = 0;
            float tv[4 * TS_GPT];
#pragma unroll
            for (int s_ = 0; s_ < 4 * TS_GPT; s_++) {
                tv[s_] = 0.f;
                if (!((cand >> s_) & 1u)) continue;
                const float d = PK ? ((const float*)&t[s_])[0] : ((const float*)&t[s_])[3];
                if (!(d > 0.f) || d > p.dtrunc) continue;
                const float sdf = (d - zcs[s_]) * rl[s_];
                if (!(sdf > -p.trunc)) continue;
                tv[s_] = fminf(1.0f, sdf * p.rtrunc);
                m |= 1u << s_;
            }
#pragma unroll
            for (int q = 0; q < TS_GPT; q++) {
                const int r = pass * TS_GPT + q;
                const uint32_t mq = (m >> (4 * q)) & 0xFu;
                // the groups of this wave's 64 that the frame updates -- completed to whole TS_WGRAN-group runs (8 = a 128-byte line per plane): the lanes of a
                // run without an update of their own load their group (or start from zeros) and store it back unchanged, so that the L2 evicts FULL lines.
                // A partially written line costs the memory side a read-modify-write: 52-54 ps per line against 38 ps for a line read and written in full
                // (tools/microbench/hbm_granule, profiles/r06_hbm_granule.json), and the line has been fetched for the updated group anyway.
                const unsigned long long now = ts_complete_runs(__ballot(mq != 0));
                if (lane == 0 && (fresh || (now & ~had[r]) != 0ull)) M[wv + 4 * r] = had[r] | now;      // (a fresh unit's words are written in full: they held garbage)
                if (!((now >> lane) & 1ull)) continue;
                const int g = (int)threadIdx.x + 256 * r;      // group index inside the unit
                float w[4] = { 0.f, 0.f, 0.f, 0.f }, sd[4] = { 0.f, 0.f, 0.f, 0.f }, c[3][4] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
                if ((had[r] >> lane) & 1ull) {
                    const float4 a = W4[g], bq = S4[g], c0 = C4[g], c1 = C4[1024 + g], c2 = C4[2048 + g];
                    ts_unpack4(a, w); ts_unpack4(bq, sd); ts_unpack4(c0, c[0]); ts_unpack4(c1, c[1]); ts_unpack4(c2, c[2]);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (!((mq >> j) & 1u)) continue;
                    const int s_ = 4 * q + j;
                    float cr, cg, cb;
                    if (PK) {
                        const uint32_t pk = __float_as_uint(((const float*)&t[s_])[1]);
                        cr = (float)(pk & 255u); cg = (float)((pk >> 8) & 255u); cb = (float)((pk >> 16) & 255u);
                    } else { cr = ((const float*)&t[s_])[0]; cg = ((const float*)&t[s_])[1]; cb = ((const float*)&t[s_])[2]; }
                    const float wo = w[j], wp = wo + 1.0f, rwp = __builtin_amdgcn_rcpf(wp);
                    sd[j] = (sd[j] * wo + tv[s_]) * rwp;
                    c[0][j] = (c[0][j] * wo + cr) * rwp; c[1][j] = (c[1][j] * wo + cg) * rwp; c[2][j] = (c[2][j] * wo + cb) * rwp;
                    w[j] = wp;
                }
                W4[g] = make_float4(w[0], w[1], w[2], w[3]); S4[g] = make_float4(sd[0], sd[1], sd[2], sd[3]);
                C4[g] = make_float4(c[0][0], c[0][1], c[0][2], c[0][3]); C4[1024 + g] = make_float4(c[1][0], c[1][1], c[1][2], c[1][3]);
                C4[2048 + g] = make_float4(c[2][0], c[2][1], c[2][2], c[2][3]);
            }
        }
    }
}

// 8-byte texels: 124 VGPRs, 4 waves per SIMD, no spills (forcing 5: 31 spilled registers and 1.6 x the time; 3: the same time as 4).  16-byte texels: 3 waves.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TS_WPE, 8))) k_ts_integrate_col(SparseTsdf v, IntParams p, const void* __restrict__ tex, int n_fixed)
{
    ts_integrate_col<true>(v, p, tex, n_fixed);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) k_ts_integrate_col_f32(SparseTsdf v, IntParams p, const void* __restrict__ tex, int n_fixed)
{
    ts_integrate_col<false>(v, p, tex, n_fixed);
}

// every group of units [0, n) becomes readable: what has never been written is zero-filled and marked written (a unit nobody ever wrote -- stamp 0, e.g.
// allocated by a frame that then ran out of capacity -- in full).  The pools are plain arrays afterwards: units() / export.
__global__ void __launch_bounds__(256) k_ts_materialize(SparseTsdf v, int n)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        const bool fresh = v.stamp[b] == 0u;
        unsigned long long* M = v.mask + (size_t)b * 16;
        float* rec = ts_unit(v, b);
        float4* S4 = reinterpret_cast<float4*>(rec);
        float4* W4 = reinterpret_cast<float4*>(rec + TS_VOX);
        float4* C4 = reinterpret_cast<float4*>(rec + 2 * TS_VOX);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned long long had = fresh ? 0ull : ts_uniform64(M[wv + 4 * r]);
            if (had == ~0ull) continue;
            const int g = (int)threadIdx.x + 256 * r;
            if (!((had >> lane) & 1ull)) { W4[g] = z4; S4[g] = z4; C4[g] = z4; C4[1024 + g] = z4; C4[2048 + g] = z4; }
            if (lane == 0) M[wv + 4 * r] = ~0ull;
        }
        __syncthreads();
        if (fresh && threadIdx.x == 0) v.stamp[b] = 0xFFFFFFFFu;      // written, by no frame
    }
}

// re-keys a table after its pools were re-allocated (growth): unit i keeps slot i.  keys must be all-ones, coord[0..n) valid.
__global__ void __launch_bounds__(256) k_ts_rehash(SparseTsdf v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = ts_pack(v.coord[3 * i], v.coord[3 * i + 1], v.coord[3 * i + 2]);
    const uint32_t mask = (1u << v.cap_hash_log2) - 1u;
    uint32_t h = ts_hash(key, v.cap_hash_log2);
    for (uint32_t probe = 0; probe <= mask; probe++, h = (h + 1) & mask) {
        unsigned long long expect = TS_EMPTY;
        if (__hip_atomic_compare_exchange_strong(&v.keys[h], &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { v.slot[h] = i; return; }
    }
    v.counters[2] = 1;
}

// ---- bulk insertion of a list of unit coordinates (merging volumes / loading a fused volume): keys first, data in a second launch
__global__ void __launch_bounds__(256) k_ts_insert_list(SparseTsdf v, const int32_t* __restrict__ coords, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) (void)ts_insert(v, coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]);
}
// self <- weighted merge of self and the given units (tsdf, weight, colour per unit): the running averages are associative in
// (sum w*tsdf, sum w), so fusing per-tile volumes equals integrating all frames into one volume up to fp32 rounding.
// One workgroup per incoming unit, thread = four 16-byte groups as in the voxel pass; only groups the incoming unit has data in are touched, a group of the
// target that has never been written becomes a copy (zeros where the incoming weight is 0) without being read.
// SRC 0: the incoming units are plain arrays in LOGICAL order (x-major, z fastest; colour [voxel][3]) -- the lists merge_() gathers from other ranks;
// SRC 1: they are the pools of another volume (storage order, colour planes, written-group words, stamps): merge_from() on one device, nothing is copied or
// re-ordered in between and what the source never wrote is never read.
template <int SRC>
__global__ void __launch_bounds__(256) k_ts_merge(SparseTsdf v, SparseTsdf o, const int32_t* __restrict__ coords, const float* __restrict__ o_tsdf, const float* __restrict__ o_weight,
                                                  const float* __restrict__ o_color, int n)
{
    const unsigned long long* __restrict__ o_mask = o.mask;
    const uint32_t* __restrict__ o_stamp = o.stamp;
    const TsLane L = ts_lane((int)threadIdx.x);
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        if (SRC == 1 && o_stamp[k] == 0u) continue;      // allocated by the source, never written
        const int b = ts_find(v, coords[3 * k], coords[3 * k + 1], coords[3 * k + 2]);
        if (b < 0) continue;
        const bool fresh = v.stamp[b] == 0u;      // a unit this call (or an aborted frame) allocated: its pool memory and mask words are uninitialised
        unsigned long long* M = v.mask + (size_t)b * 16;
        float* rec = ts_unit(v, b);
        float4* S4 = reinterpret_cast<float4*>(rec);
        float4* W4 = reinterpret_cast<float4*>(rec + TS_VOX);
        float4* C4 = reinterpret_cast<float4*>(rec + 2 * TS_VOX);
        const float* orec = SRC == 1 ? ts_unit(o, k) : nullptr;
        const float* ot = SRC == 1 ? orec : o_tsdf + (size_t)k * TS_VOX;
        const float* ow = SRC == 1 ? orec + TS_VOX : o_weight + (size_t)k * TS_VOX;
        const float* oc = SRC == 1 ? orec + 2 * TS_VOX : o_color + (size_t)k * TS_VOX * 3;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned long long had = fresh ? 0ull : ts_uniform64(M[wv + 4 * r]);
            const int g = (int)threadIdx.x + 256 * r;
            float w1[4] = { 0.f, 0.f, 0.f, 0.f }, t1[4] = { 0.f, 0.f, 0.f, 0.f }, c1[3][4] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
            if (SRC == 1) {
                const unsigned long long src_has = ts_uniform64(o_mask[(size_t)k * 16 + wv + 4 * r]);
                if ((src_has >> lane) & 1ull) {
                    ts_unpack4(reinterpret_cast<const float4*>(ow)[g], w1); ts_unpack4(reinterpret_cast<const float4*>(ot)[g], t1);
#pragma unroll
                    for (int c = 0; c < 3; c++) ts_unpack4(reinterpret_cast<const float4*>(oc)[1024 * c + g], c1[c]);
                }
            } else {
                const int i0 = ((L.lx + 4 * r) * TS_RES + L.iy) * TS_RES + L.iz0;      // logical index of the group's first voxel
                ts_unpack4(*reinterpret_cast<const float4*>(ow + i0), w1); ts_unpack4(*reinterpret_cast<const float4*>(ot + i0), t1);
                float a[12];
#pragma unroll
                for (int q = 0; q < 3; q++) ts_unpack4(*reinterpret_cast<const float4*>(oc + 3 * i0 + 4 * q), a + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; j++) { c1[0][j] = a[3 * j]; c1[1][j] = a[3 * j + 1]; c1[2][j] = a[3 * j + 2]; }
            }
            const bool any = w1[0] > 0.f || w1[1] > 0.f || w1[2] > 0.f || w1[3] > 0.f;
            const unsigned long long now = ts_complete_runs(__ballot(any));      // whole 128-byte lines are written back, as in the voxel pass
            if (lane == 0 && (fresh || (now & ~had) != 0ull)) M[wv + 4 * r] = had | now;
            if (!((now >> lane) & 1ull)) continue;
            float w0[4] = { 0.f, 0.f, 0.f, 0.f }, t0[4] = { 0.f, 0.f, 0.f, 0.f }, c0[3][4] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
            if ((had >> lane) & 1ull) {
                ts_unpack4(W4[g], w0); ts_unpack4(S4[g], t0);
#pragma unroll
                for (int c = 0; c < 3; c++) ts_unpack4(C4[1024 * c + g], c0[c]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!(w1[j] > 0.f)) continue;
                if (!(w0[j] > 0.f)) {      // nothing there yet: the incoming voxel as it is (no (t * w) / w round trip)
                    t0[j] = t1[j]; w0[j] = w1[j];
#pragma unroll
                    for (int c = 0; c < 3; c++) c0[c][j] = c1[c][j];
                    continue;
                }
                const float ws = w0[j] + w1[j], rs = 1.0f / ws;
                t0[j] = (t0[j] * w0[j] + t1[j] * w1[j]) * rs;
#pragma unroll
                for (int c = 0; c < 3; c++) c0[c][j] = (c0[c][j] * w0[j] + c1[c][j] * w1[j]) * rs;
                w0[j] = ws;
            }
            W4[g] = make_float4(w0[0], w0[1], w0[2], w0[3]); S4[g] = make_float4(t0[0], t0[1], t0[2], t0[3]);
#pragma unroll
            for (int c = 0; c < 3; c++) C4[1024 * c + g] = make_float4(c0[c][0], c0[c][1], c0[c][2], c0[c][3]);
        }
        __syncthreads();
        if (fresh && threadIdx.x == 0) v.stamp[b] = 0xFFFFFFFFu;      // written, by no frame
    }
}

// ------------------------------------------------------------------------------------------------ C ABI (include/gsrast.h)
static SparseTsdf make_view(const gsr_tsdf_sparse* s)
{
    SparseTsdf v;
    v.keys = (unsigned long long*)s->keys; v.slot = s->slot; v.coord = s->coord; v.stamp = s->stamp; v.list = s->list; v.counters = s->counters;
    v.mask = (unsigned long long*)s->mask; v.cap_hash_log2 = s->cap_hash_log2; v.cap_blocks = s->cap_blocks; v.chunk0_log2 = s->chunk0_log2;
    for (int c = 0; c < GSR_TSDF_MAX_CHUNKS; c++) v.chunk[c] = c < (int)s->n_chunks ? s->chunk[c] : nullptr;
    return v;
}
static int check_vol(const gsr_tsdf_sparse* s)
{
    if (!s || !s->keys || !s->slot || !s->coord || !s->stamp || !s->list || !s->counters || !s->mask) {
        gsr_set_error("tsdf_sparse: null volume buffers"); return 1;
    }
    if (s->n_chunks < 1 || s->n_chunks > GSR_TSDF_MAX_CHUNKS || s->chunk0_log2 > 27 || s->cap_blocks != (1u << (s->chunk0_log2 + s->n_chunks - 1))) {
        gsr_set_error("tsdf_sparse: %u chunks of first size 2^%u do not make a pool of %u units (chunk c >= 1 holds 2^(chunk0_log2 + c - 1) units)", s->n_chunks, s->chunk0_log2,
                      s->cap_blocks); return 1;
    }
    for (uint32_t c = 0; c < s->n_chunks; c++)
        if (!s->chunk[c]) { gsr_set_error("tsdf_sparse: null chunk %u", c); return 1; }
    if (s->cap_hash_log2 < 4 || s->cap_hash_log2 > 30 || s->cap_blocks == 0 || (1ull << s->cap_hash_log2) < 2ull * s->cap_blocks) {
        gsr_set_error("tsdf_sparse: hash table must hold at least twice the unit capacity"); return 1;
    }
    if (!(s->voxel_length > 0.f) || !(s->sdf_trunc > 0.f)) { gsr_set_error("tsdf_sparse: voxel_length / sdf_trunc must be positive"); return 1; }
    // a depth sample opens the units its +-sdf_trunc box overlaps, at most 4 per axis: a wider band would be dropped sample by sample
    if (2.0f * s->sdf_trunc > 3.0f * TS_RES * s->voxel_length) {
        gsr_set_error("tsdf_sparse: sdf_trunc %g exceeds 1.5 units (%g = 24 voxels): the truncation band must fit in 4 units per axis", s->sdf_trunc,
                      1.5f * TS_RES * s->voxel_length); return 1;
    }
    return 0;
}

static void fill_params(const gsr_tsdf_sparse* s, int32_t W, int32_t H, float fx, float fy, float cx, float cy, const float* extrinsic, const float* pose,
                        float depth_trunc, int32_t stride, TouchParams& t, IntParams& p)
{
    t.W = W; t.H = H; t.stride = stride; t.fx = fx; t.fy = fy; t.cx = cx; t.cy = cy; t.rfx = 1.0f / fx; t.rfy = 1.0f / fy;
    t.trunc = s->sdf_trunc; t.dtrunc = depth_trunc; t.unit_len = s->voxel_length * TS_RES; t.inv_unit = 1.0f / t.unit_len;
    for (int k = 0; k < 12; k++) t.P[k] = pose[k];
    p.W = W; p.H = H; p.vl = s->voxel_length; p.trunc = s->sdf_trunc; p.dtrunc = depth_trunc; p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy;
    p.rfx = 1.0f / fx; p.rfy = 1.0f / fy; p.rtrunc = 1.0f / s->sdf_trunc; p.unit_len = t.unit_len;
    for (int k = 0; k < 12; k++) p.E[k] = extrinsic[k];
}
// what the counters of a finished frame mean for the caller (shared by the synchronous entry points and gsr_tsdf_sparse_status)
static int frame_errors(const gsr_tsdf_sparse* s, const int32_t* c, hipStream_t st)
{
    if (c[2]) { gsr_set_error("tsdf_sparse: capacity exhausted (%u units); allocate a larger volume", s->cap_blocks); return 1; }
    if (c[3]) {
        (void)gsr_memset_async((int32_t*)s->counters + 3, 0, sizeof(int32_t), st);
        gsr_set_error("tsdf_sparse: a depth sample lies outside the addressable volume (|unit coordinate| >= 2^20, i.e. %g scene units from the origin)",
                      (double)(1 << 20) * s->voxel_length * TS_RES); return 1;
    }
    return 0;
}

extern "C" int gsr_tsdf_sparse_integrate2(const gsr_tsdf_sparse* s, int32_t W, int32_t H, const float* depth, const float* rgb, int32_t quant, float fx,
                                          float fy, float cx, float cy, const float* extrinsic, const float* pose, float depth_trunc, int32_t stride,
                                          uint32_t frame, float* texels, int32_t* status_host, uint32_t flags, void* stream)
{
    if (check_vol(s)) return 1;
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535 || stride <= 0 || !depth || !rgb || !extrinsic || !pose || !texels || quant < 0 || quant > 2) {
        gsr_set_error("tsdf_sparse_integrate2: bad arguments"); return 1;
    }
    if ((flags & GSR_TSDF_NO_SYNC) && !status_host) { gsr_set_error("tsdf_sparse_integrate2: GSR_TSDF_NO_SYNC needs a status buffer"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    SparseTsdf v = make_view(s);
    TouchParams t; IntParams p;
    fill_params(s, W, H, fx, fy, cx, cy, extrinsic, pose, depth_trunc, stride, t, p);
    const int n = ((W + stride - 1) / stride) * ((H + stride - 1) / stride), N = W * H;
    if (gsr_memset_async(v.counters + 1, 0, sizeof(int32_t), st)) { gsr_set_error("tsdf_sparse: reset list"); return 1; };
    const bool pk = quant == 2;      // byte colours: 8-byte texels
    if (pk) hipLaunchKernelGGL(k_ts_texels<true>, dim3((N + 255) / 256), dim3(256), 0, st, depth, rgb, (void*)texels, N, (int)quant);
    else hipLaunchKernelGGL(k_ts_texels<false>, dim3((N + 255) / 256), dim3(256), 0, st, depth, rgb, (void*)texels, N, (int)quant);
    hipLaunchKernelGGL(k_ts_touch_insert, dim3((n + 255) / 256), dim3(256), 0, st, v, t, depth, n);
    hipLaunchKernelGGL(k_ts_touch_stamp, dim3((n + 255) / 256), dim3(256), 0, st, v, t, depth, n, frame);
    // the voxel pass reads the list length on the device: it is enqueued without waiting for the host (a persistent grid: 8 workgroups per CU)
    if (pk) hipLaunchKernelGGL(k_ts_integrate_col, dim3(TS_GRID), dim3(256), 0, st, v, p, (const void*)texels, -1);
    else hipLaunchKernelGGL(k_ts_integrate_col_f32, dim3(TS_GRID), dim3(256), 0, st, v, p, (const void*)texels, -1);
    int32_t local[4] = { 0, 0, 0, 0 };
    int32_t* c = status_host ? status_host : local;
    GSR_CHECK(hipMemcpyAsync(c, v.counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st), "tsdf_sparse: read counters");
    if (flags & GSR_TSDF_NO_SYNC) return gsr_check_launch("tsdf_sparse_integrate2", st, false);      // the caller waits for its own event and calls gsr_tsdf_sparse_status
    GSR_CHECK(hipStreamSynchronize(st), "tsdf_sparse: sync");
    if (frame_errors(s, c, st)) return 1;
    return gsr_check_launch("tsdf_sparse_integrate2", st, false);
}

// the status words a GSR_TSDF_NO_SYNC frame copied to the host, once the caller knows the copy has landed: 0, or 1 + gsr_last_error() exactly like the synchronous call
extern "C" int gsr_tsdf_sparse_status(const gsr_tsdf_sparse* s, const int32_t* status_host, void* stream)
{
    if (check_vol(s) || !status_host) return 1;
    return frame_errors(s, status_host, (hipStream_t)stream);
}

// after the caller re-allocated the arrays of a volume (growth: a further chunk, coord / stamp / mask of units [0, n) copied into larger arrays, keys all-ones,
// counters[0] = n): every unit gets its key back with the slot it had.  No voxel is touched.
extern "C" int gsr_tsdf_sparse_rehash(const gsr_tsdf_sparse* s, int32_t n_units, void* stream)
{
    if (check_vol(s)) return 1;
    if (n_units < 0 || (uint32_t)n_units > s->cap_blocks) { gsr_set_error("tsdf_sparse_rehash: %d units do not fit the volume", n_units); return 1; }
    if (n_units == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ts_rehash, dim3((n_units + 255) / 256), dim3(256), 0, st, make_view(s), n_units);
    return gsr_check_launch("tsdf_sparse_rehash", st, false);
}

// the capacity check both merges share: keys of the incoming units first, then one read of the counters
static int merge_insert(const gsr_tsdf_sparse* s, SparseTsdf& v, const int32_t* coords, int32_t n_units, hipStream_t st)
{
    hipLaunchKernelGGL(k_ts_insert_list, dim3((n_units + 255) / 256), dim3(256), 0, st, v, coords, n_units);
    int32_t c[3] = { 0, 0, 0 };
    GSR_CHECK(hipMemcpyAsync(c, v.counters, sizeof(c), hipMemcpyDeviceToHost, st), "tsdf_sparse: read counters");
    GSR_CHECK(hipStreamSynchronize(st), "tsdf_sparse: sync");
    if (c[2]) { gsr_set_error("tsdf_sparse: capacity exhausted (%u units) while merging", s->cap_blocks); return 1; }
    return 0;
}
extern "C" int gsr_tsdf_sparse_merge(const gsr_tsdf_sparse* s, int32_t n_units, const int32_t* coords, const float* tsdf, const float* weight,
                                     const float* color, void* stream)
{
    if (check_vol(s)) return 1;
    if (n_units <= 0) return 0;
    if (!coords || !tsdf || !weight || !color) { gsr_set_error("tsdf_sparse_merge: null unit list"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    SparseTsdf v = make_view(s);
    if (merge_insert(s, v, coords, n_units, st)) return 1;
    hipLaunchKernelGGL(k_ts_merge<0>, dim3((uint32_t)std::min(n_units, 8 * TS_GRID)), dim3(256), 0, st, v, v, coords, tsdf, weight, color, (int)n_units);
    return gsr_check_launch("tsdf_sparse_merge", st, false);
}
// ABI 8.  vol <- weighted merge with the first n_units units of another volume ON THE SAME DEVICE, read where they lie (pools in storage order, written-group
// words, stamps): nothing is exported, re-ordered or copied in between, and what `other` never wrote is never read.
extern "C" int gsr_tsdf_sparse_merge_volume(const gsr_tsdf_sparse* s, const gsr_tsdf_sparse* other, int32_t n_units, void* stream)
{
    if (check_vol(s) || check_vol(other)) return 1;
    if (n_units <= 0) return 0;
    if ((uint32_t)n_units > other->cap_blocks) { gsr_set_error("tsdf_sparse_merge_volume: %d units exceed the source's capacity", n_units); return 1; }
    if (s->voxel_length != other->voxel_length || s->sdf_trunc != other->sdf_trunc) { gsr_set_error("tsdf_sparse_merge_volume: volumes must share voxel_length and sdf_trunc"); return 1; }
    if (s->chunk[0] == other->chunk[0]) { gsr_set_error("tsdf_sparse_merge_volume: a volume cannot be merged into itself"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    SparseTsdf v = make_view(s);
    if (merge_insert(s, v, other->coord, n_units, st)) return 1;
    hipLaunchKernelGGL(k_ts_merge<1>, dim3((uint32_t)std::min(n_units, 8 * TS_GRID)), dim3(256), 0, st, v, make_view(other), (const int32_t*)other->coord, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (int)n_units);
    return gsr_check_launch("tsdf_sparse_merge_volume", st, false);
}
// ABI 8.  Makes units [0, n_units) plain arrays: groups that were never written are zero-filled and marked written (see k_ts_materialize).  What a reader of
// the pools (export, marching cubes, units()) calls first; the kernels of this file never need it.
extern "C" int gsr_tsdf_sparse_materialize(const gsr_tsdf_sparse* s, int32_t n_units, void* stream)
{
    if (check_vol(s)) return 1;
    if (n_units < 0 || (uint32_t)n_units > s->cap_blocks) { gsr_set_error("tsdf_sparse_materialize: %d units do not fit the volume", n_units); return 1; }
    if (n_units == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ts_materialize, dim3((uint32_t)std::min(n_units, 8 * TS_GRID)), dim3(256), 0, st, make_view(s), (int)n_units);
    return gsr_check_launch("tsdf_sparse_materialize", st, false);
}
