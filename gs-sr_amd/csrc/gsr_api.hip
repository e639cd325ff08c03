// gsr_api.hip -- C ABI of libgsrast_hip.so (see include/gsrast.h): arena carving, stage orchestration, errors.
#include "gsr_common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

int gsr_tile_sort_passes(int T);

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void gsr_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gsr_last_error(void) { return g_err; }
extern "C" int32_t gsr_abi_version(void) { return GSR_ABI_VERSION; }

int gsr_check_launch(const char* what, hipStream_t s, bool debug)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gsr_set_error("%s: launch failed: %s", what, hipGetErrorString(e)); return 1; }
    if (debug) {   // CHECK_CUDA semantics of the reference (auxiliary.h:166-173)
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) { gsr_set_error("%s: %s", what, hipGetErrorString(e)); return 1; }
    }
    return 0;
}

__global__ void __launch_bounds__(256) k_fill_words(uint32_t* __restrict__ p, uint32_t v, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int gsr_memset_async(void* p, int byte_value, size_t nbytes, hipStream_t s)
{
    if (nbytes == 0) return 0;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive && (nbytes & 3) == 0 && (((uintptr_t)p) & 3) == 0) {
        const uint32_t b = (uint32_t)(byte_value & 0xFF), w = b | (b << 8) | (b << 16) | (b << 24);
        const size_t n = nbytes / 4;
        const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_fill_words, dim3(blocks), dim3(256), 0, s, (uint32_t*)p, w, n);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    return hipMemsetAsync(p, byte_value, nbytes, s) == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------ stage profiler
// Optional: HIP events recorded on the launch stream around every stage; read back with gsr_profile_read.
// Used by bench.py for the live kernel durations behind the roofline figure.  One mutex guards the record lists, so concurrent
// forwards from several host threads / streams only serialise on the bookkeeping (a few hundred ns), never on the GPU work.
#include <chrono>
#include <vector>
#include <algorithm>
struct ProfRec { int label; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static std::atomic<uint32_t> g_prof_on{0};      // bit i: stage i is timed
static std::vector<ProfRec> g_prof;
static std::vector<ProfRec> g_prof_free;
static double g_prof_ms[GSR_PROF_LABELS];
static uint64_t g_prof_n[GSR_PROF_LABELS];

static void prof_drain()      // caller holds g_prof_mu
{
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_prof_ms[r.label] += ms; g_prof_n[r.label] += 1;
        }
        g_prof_free.push_back(r);
    }
    g_prof.clear();
}
struct ProfScope {
    ProfRec r; bool on; hipStream_t s;
    ProfScope(int label, hipStream_t s_) : on((g_prof_on.load(std::memory_order_relaxed) >> label) & 1u), s(s_)
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof.size() >= 8192) prof_drain();
        if (!g_prof_free.empty()) { r = g_prof_free.back(); g_prof_free.pop_back(); }
        else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
        r.label = label;
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};
extern "C" int gsr_profile_enable(int32_t enable)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    // enable: 0 = off, 1 = every stage (the original meaning), otherwise bit (8 + i) selects stage i alone -- an event pair costs ~10 us of
    // stream idle time per stage boundary (profiles/r03_timeline.json), so a caller that wants ONE kernel's duration inside a timed
    // region enables only that stage
    // any other non-zero value whose stage mask is empty (2, -1 & 0xFF, ...) keeps the pre-round-3 meaning "on" = every stage
    const uint32_t mask = ((uint32_t)enable >> 8) & 0xFFu;
    g_prof_on = enable == 0 ? 0u : ((enable == 1 || mask == 0u) ? 0xFFu : mask);
    for (int i = 0; i < GSR_PROF_LABELS; i++) { g_prof_ms[i] = 0; g_prof_n[i] = 0; }
    return 0;
}
extern "C" int gsr_profile_read(double* ms_total, uint64_t* counts)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    for (int i = 0; i < GSR_PROF_LABELS; i++) { ms_total[i] = g_prof_ms[i]; counts[i] = g_prof_n[i]; }
    return 0;
}

// ------------------------------------------------------------------------------------------------ arenas
template <typename T>
static T* take(char*& p, size_t n)
{
    T* r = reinterpret_cast<T*>(p);
    p += gsr_align(n * sizeof(T));
    return r;
}

GeomView gsr_carve_geom(int variant, int P, void* base)
{
    GeomView g;
    char* p = reinterpret_cast<char*>(base);
    const size_t n = (size_t)(P > 0 ? P : 1);
    const uint32_t nblk = gsr_div_up((uint32_t)n, GSR_SORT_BLOCK);
    g.depth_key = take<uint32_t>(p, n);
    g.tiles_touched = take<uint32_t>(p, n);
    g.rect = take<ushort4>(p, n);
    g.cull = take<float4>(p, n * 2);
    g.rec = take<float4>(p, n * gsr_rec_stride(variant));
    g.clamped = take<uint32_t>(p, n);
    g.vals_a = take<uint32_t>(p, n);
    g.offsets = take<uint32_t>(p, n);
    g.keys_b = take<uint32_t>(p, n);
    g.vals_b = take<uint32_t>(p, n);
    g.sorted_idx = g.vals_a;                 // four sort passes: identity -> vals_b -> vals_a -> vals_b -> vals_a (gsr_launch_depth_order)
    g.hist = take<uint32_t>(p, gsr_sort_hist_words(nblk, 256));   // 256-bin digits
    g.scan_tmp = take<uint32_t>(p, gsr_div_up((uint32_t)n, 256) + 64);       // block sums of the prefix: 256-gaussian blocks when the preprocess writes them
    g.counters = take<uint32_t>(p, 64);
    g.bytes = (size_t)(p - reinterpret_cast<char*>(base));
    return g;
}

BinView gsr_carve_bin(int variant, uint32_t R, int W, int H, void* base)
{
    (void)variant;
    BinView b;
    char* p = reinterpret_cast<char*>(base);
    const size_t n = R > 0 ? R : 1;
    const uint32_t nblk = gsr_div_up((uint32_t)n, GSR_SORT_BLOCK);
    b.point_list = take<uint32_t>(p, n);
    b.tile_keys = take<uint32_t>(p, n);
    b.keys_b = take<uint32_t>(p, n);
    b.vals_b = take<uint32_t>(p, n);
    b.hist = take<uint32_t>(p, gsr_sort_hist_words(nblk, 256));
    b.scan_tmp = take<uint32_t>(p, 64);
    {
        const size_t T = (size_t)((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE);
        b.qmask = take<unsigned long long>(p, ((n >> 6) + T + 2) * 4);
        b.tile_tab = take<uint32_t>(p, gsr_tile_bucket_words((uint32_t)n, T) + (T <= (size_t)GSR_TB_TILES_MAX ? (size_t)GSR_TB_ROWS_MAX * GSR_TB_GROUPS_MAX : (size_t)2));
    }
    b.cap = R;
    b.bytes = (size_t)(p - reinterpret_cast<char*>(base));
    return b;
}

ImgView gsr_carve_img(int variant, int W, int H, void* base)
{
    ImgView im;
    char* p = reinterpret_cast<char*>(base);
    const size_t N = (size_t)W * H;
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    im.final_T = take<float>(p, N * (variant == GSR_SURFEL ? 3 : 1));
    im.n_contrib = take<uint32_t>(p, N * (variant == GSR_SURFEL ? 2 : 1));
    im.ranges = take<uint2>(p, (size_t)gx * gy);
    im.tile_order = take<uint32_t>(p, (size_t)gx * gy + 1);
    im.bytes = (size_t)(p - reinterpret_cast<char*>(base));
    return im;
}

extern "C" size_t gsr_geom_bytes(int32_t variant, int32_t P) { return gsr_carve_geom(variant, P, nullptr).bytes; }
extern "C" size_t gsr_img_bytes(int32_t variant, int32_t W, int32_t H) { return gsr_carve_img(variant, W, H, nullptr).bytes; }
extern "C" size_t gsr_binning_bytes(int32_t variant, uint32_t R, int32_t W, int32_t H) { return gsr_carve_bin(variant, R, W, H, nullptr).bytes; }
extern "C" size_t gsr_backward_scratch_bytes(int32_t variant, int32_t P)
{
    return gsr_align((size_t)(P > 0 ? P : 1) * gsr_acc_stride(variant) * sizeof(float));
}

extern "C" uint32_t gsr_binning_capacity(int32_t variant, size_t bytes, int32_t W, int32_t H);

static int check_cfg(const gsr_cfg* cfg, const gsr_inputs* in)
{
    if (!cfg || !in) { gsr_set_error("null cfg/inputs"); return 1; }
    if (cfg->variant < GSR_EWA || cfg->variant > GSR_PLANE) { gsr_set_error("bad variant %d", cfg->variant); return 1; }
    if (cfg->P < 0 || cfg->W <= 0 || cfg->H <= 0) { gsr_set_error("bad sizes P=%d W=%d H=%d", cfg->P, cfg->W, cfg->H); return 1; }
    if (cfg->P == 0) return 0;
    if (!in->means3D || !in->opacities) { gsr_set_error("means3D/opacities must be provided"); return 1; }
    if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) {
        gsr_set_error("Please provide excatly one of either SHs or precomputed colors!"); return 1;
    }
    const bool has_sr = in->scales && in->rotations;
    if (has_sr == (in->cov3D_precomp != nullptr) || (!has_sr && (in->scales || in->rotations))) {
        gsr_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"); return 1;
    }
    if (in->shs && (cfg->M < (cfg->D + 1) * (cfg->D + 1) || cfg->D < 0 || cfg->D > 3)) {
        gsr_set_error("sh_degree %d needs %d coefficients, got M=%d", cfg->D, (cfg->D + 1) * (cfg->D + 1), cfg->M); return 1;
    }
    if (!cfg->bg || !cfg->viewmatrix || !cfg->projmatrix || !cfg->campos) { gsr_set_error("camera tensors missing"); return 1; }
    return 0;
}

// ------------------------------------------------------------------------------------------------ pinned mailbox
// num_rendered travels through one mapped, pinned host word per device that the last scan kernel writes directly
// (system-scope store): the forward's single sync is then a bare hipStreamSynchronize, no D2H copy command.
// Concurrency: a forward owns its slot from `take_slot` until it has read the word back.  Slots are handed out by an atomic ticket
// and guarded by a per-slot busy flag, so forwards running concurrently on several host threads / streams of one device never share
// a word (round 1 used an unsynchronised `next++`); with more than GSR_MAIL_SLOTS forwards in flight the caller spins for a free slot.
#define GSR_MAIL_SLOTS 64
struct Mailbox {
    uint32_t* host = nullptr; uint32_t* dev = nullptr;
    std::atomic<unsigned> next{0};
    std::atomic<int> busy[GSR_MAIL_SLOTS];
    std::atomic<int> order_ttl{0};          // forwards for which the longest-first launch order stays on after the last long-list report
    std::atomic<int> global_ttl{0};         // forwards for which the global depth order stays on after the last report of a list > 6000 entries
};
static Mailbox g_mail[64];
static std::mutex g_mail_mu;
static unsigned take_slot(Mailbox* m)
{
    for (;;) {
        const unsigned s = m->next.fetch_add(1u, std::memory_order_relaxed) % GSR_MAIL_SLOTS;
        int expect = 0;
        if (m->busy[s].compare_exchange_strong(expect, 1, std::memory_order_acquire)) return s;
    }
}
static void release_slot(Mailbox* m, unsigned s) { m->busy[s].store(0, std::memory_order_release); }
static Mailbox* mailbox()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
    Mailbox& m = g_mail[d];
    std::lock_guard<std::mutex> lk(g_mail_mu);
    if (!m.host) {
        void* h = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&h, (GSR_MAIL_SLOTS + 1) * 64, hipHostMallocMapped) != hipSuccess) return nullptr;     // one cache line per slot + the long-list feedback word
        ((uint32_t*)h)[16 * GSR_MAIL_SLOTS] = 0u;
        if (hipHostGetDevicePointer(&dp, h, 0) != hipSuccess) { (void)hipHostFree(h); return nullptr; }
        for (int i = 0; i < GSR_MAIL_SLOTS; i++) m.busy[i].store(0);
        m.host = (uint32_t*)h; m.dev = (uint32_t*)dp;
    }
    return &m;
}

// Launch order of the blend workgroups, decided per forward.  Raster order (workgroup b = tile b, XCD b % 8) spreads every image region over the
// eight XCDs; on object-centric scenes a few hundred tiles carry lists 5-50x the mean and the launch ends on whichever of them start late, which
// k_tile_order (longest lists first, 11 us) removes: 986 -> 1209 it/s with lists up to 3800 entries, but -1.7 % on the uniform BASELINE scene
// (tools/ab_tile_order.sh, profiles/r03_launch_order.txt).  So the order is used when it pays: every blend forward stores its tile's list length into a
// mapped per-device word when it exceeds max(1024, 4 x mean) (BlendParams::long_word), and the forwards that follow a report -- the next 64 -- run
// k_tile_order.  The decision costs the host one read of pinned memory; a stale decision is only a slower or faster launch order, never a wrong one
// (the order's validity travels in the image arena, ImgView::tile_order[T]).  (GSR_TILE_ORDER=0|1 forced it for the round-3/4 A/B runs: removed in round 6.)
bool gsr_tile_order_wanted()
{
    Mailbox* mb = mailbox();
    return mb && mb->order_ttl.load(std::memory_order_relaxed) > 0;      // set by gsr_decide_depth_order, once per forward
}

// Once per forward: read the long-list word the previous forwards' blend kernels stored into, refresh the two counters it drives -- longest-first
// launch order for 64 forwards after a list beyond max(1024, 4 x mean); GLOBAL depth order for 64 forwards after a list beyond 6000 entries, where
// the prologue's global-memory radix path loses to it (816 vs 780 it/s at 12 633 entries, a tie at 3800: profiles/r03_skewed_density.txt) --
// and return the depth order of THIS forward.  The caller hands it to every stage launcher; the preprocess kernel records it in the geom arena
// (GeomView::counters[GSR_CNT_MODE]) for later calls on that arena (gsr_forward_stage2).
bool gsr_decide_depth_order(const gsr_cfg* cfg)
{
    const int T = ((cfg->W + GSR_TILE - 1) / GSR_TILE) * ((cfg->H + GSR_TILE - 1) / GSR_TILE);
    bool forced = false;
    bool global = gsr_depth_order_static_rule(cfg->P, T, &forced, cfg->variant);
    Mailbox* mb = mailbox();
    if (mb) {
        volatile uint32_t* w = mb->host + 16 * GSR_MAIL_SLOTS;
        const uint32_t longest = *w;
        if (longest != 0u) {
            *w = 0u;
            mb->order_ttl.store(64, std::memory_order_relaxed);
            if (longest > 6000u) mb->global_ttl.store(64, std::memory_order_relaxed);
            else { const int t = mb->global_ttl.load(std::memory_order_relaxed); if (t > 0) mb->global_ttl.store(t - 1, std::memory_order_relaxed); }
        } else {
            int t = mb->order_ttl.load(std::memory_order_relaxed);
            if (t > 0) mb->order_ttl.store(t - 1, std::memory_order_relaxed);
            t = mb->global_ttl.load(std::memory_order_relaxed);
            if (t > 0) mb->global_ttl.store(t - 1, std::memory_order_relaxed);
        }
        if (!forced && mb->global_ttl.load(std::memory_order_relaxed) > 0) global = true;
    }
    return global;
}
uint32_t* gsr_long_list_word()
{
    Mailbox* mb = mailbox();
    return mb ? mb->dev + 16 * GSR_MAIL_SLOTS : nullptr;
}

// blockIdx -> tile with 4x4-tile blocks dealt out to the eight XCDs cyclically (workgroup b runs on XCD b % 8): XCD x owns the blocks with
// (bx + 3 by) % 8 == x and walks them in raster order, 16 tiles each; the eight lists are interleaved so that b % 8 selects the list.  Where the
// lists differ in length the tail is filled from whichever list still has tiles (a permutation in any case).  Built on the host once per
// (device, grid), kept for the life of the process; not built while a stream is being captured (raster order until an eager call has built it).
const uint32_t* gsr_static_tile_map(int gx, int gy, hipStream_t s)
{
    struct Ent { int dev, gx, gy; uint32_t* map; };
    static std::mutex mu;
    static std::vector<Ent> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    for (const Ent& e : cache) if (e.dev == dev && e.gx == gx && e.gy == gy) return e.map;
    {   // hipMalloc / hipMemcpy are illegal while the launch stream is being captured: raster order until an eager call has built the map
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    }
    const int T = gx * gy, B = 4, bxn = (gx + B - 1) / B, byn = (gy + B - 1) / B;
    std::vector<std::vector<uint32_t>> lists(8);
    for (int by = 0; by < byn; by++)
        for (int bx = 0; bx < bxn; bx++) {
            std::vector<uint32_t>& L = lists[(bx + 3 * by) & 7];
            for (int ty = by * B; ty < std::min(gy, by * B + B); ty++)
                for (int tx = bx * B; tx < std::min(gx, bx * B + B); tx++) L.push_back((uint32_t)(ty * gx + tx));
        }
    std::vector<uint32_t> map((size_t)T);
    size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < T; b++) {
        int x = b & 7;
        for (int k = 0; k < 8 && pos[x] >= lists[x].size(); k++) x = (x + 1) & 7;      // this XCD's list is used up: take from the next that is not
        map[(size_t)b] = lists[x][pos[x]++];
    }
    uint32_t* d = nullptr;
    if (hipMalloc(&d, (size_t)T * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemcpy(d, map.data(), (size_t)T * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return nullptr; }
    cache.push_back({dev, gx, gy, d});
    return d;
}

// the reference's message (auxiliary.h:157); it then traps the device, this library fails the forward call
#define GSR_PREFILTERED_MSG "Point is filtered although prefiltered is set. This shouldn't happen!"

// ------------------------------------------------------------------------------------------------ forward
// cfg->debug (the stream has just been synchronised by gsr_check_launch): did k_duplicate emit what the preprocess kernel counted?
static int check_cull_agreement(const gsr_cfg* cfg, const GeomView& g, hipStream_t s)
{
    if (!cfg->debug) return 0;
    uint32_t w = 0u;
    GSR_CHECK(hipMemcpyAsync(&w, g.counters + GSR_CNT_CULL_MISMATCH, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read cull agreement word");
    GSR_CHECK(hipStreamSynchronize(s), "debug sync");
    if (w) { gsr_set_error("tile culling: k_duplicate emitted a different number of instances than the preprocess kernel counted (gsr_tile_cull.h evaluated differently in the two units)"); return 1; }
    return 0;
}

extern "C" int gsr_forward_stage1_ex(const gsr_cfg* cfg, const gsr_inputs* in, void* geom, size_t geom_bytes,
                                     int32_t* radii, uint32_t* num_rendered_host, uint32_t* depth_order_host, void* stream)
{
    if (check_cfg(cfg, in)) return 1;
    hipStream_t s = (hipStream_t)stream;
    *num_rendered_host = 0;
    if (depth_order_host) *depth_order_host = 0u;
    if (cfg->P == 0) return 0;
    GeomView g = gsr_carve_geom(cfg->variant, cfg->P, geom);
    if (g.bytes > geom_bytes) { gsr_set_error("geom buffer too small: %zu < %zu", geom_bytes, g.bytes); return 1; }
    Mailbox* mb = mailbox();
    const unsigned slot = mb ? take_slot(mb) : 0u;
    struct SlotGuard { Mailbox* m; unsigned s; ~SlotGuard() { if (m) release_slot(m, s); } } guard{mb, slot};
    if (mb) mb->host[16 * slot + 1] = 0u;                      // "a gaussian failed the frustum test although prefiltered is set"
    const bool global_order = gsr_decide_depth_order(cfg);      // recorded in the geom arena by the preprocess kernel, and handed to the caller for stage 2
    if (depth_order_host) *depth_order_host = global_order ? GSR_MODE_GLOBAL : GSR_MODE_TILE;
    { ProfScope ps(GSR_PROF_PREPROCESS, s); if (gsr_launch_preprocess(cfg, in, g, radii, s, global_order, mb ? mb->dev + 16 * slot + 1 : nullptr)) return 1; }
    { ProfScope ps(GSR_PROF_DEPTH_ORDER, s); if (gsr_launch_depth_order(cfg, g, mb ? mb->dev + 16 * slot : nullptr, s, global_order, true)) return 1; }
    // the one host<->device sync of the forward (reference: cudaMemcpy of point_offsets[P-1], rasterizer_impl.cu:281)
    if (mb) {
        GSR_CHECK(hipStreamSynchronize(s), "stage1 sync");
        *num_rendered_host = *(volatile uint32_t*)(mb->host + 16 * slot);
        if (cfg->prefiltered && *(volatile uint32_t*)(mb->host + 16 * slot + 1)) { gsr_set_error("%s", GSR_PREFILTERED_MSG); return 1; }
    } else {
        GSR_CHECK(hipMemcpyAsync(num_rendered_host, g.counters, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read num_rendered");
        GSR_CHECK(hipStreamSynchronize(s), "stage1 sync");
    }
    return 0;
}
extern "C" int gsr_forward_stage1(const gsr_cfg* cfg, const gsr_inputs* in, void* geom, size_t geom_bytes,
                                  int32_t* radii, uint32_t* num_rendered_host, void* stream)
{
    return gsr_forward_stage1_ex(cfg, in, geom, geom_bytes, radii, num_rendered_host, nullptr, stream);
}

extern "C" int gsr_forward_stage2_ex(const gsr_cfg* cfg, const gsr_inputs* in, void* geom, size_t geom_bytes,
                                     void* binning, size_t binning_bytes, void* img, size_t img_bytes,
                                     uint32_t num_rendered, uint32_t depth_order, const gsr_outputs* out, void* stream)
{
    if (check_cfg(cfg, in)) return 1;
    hipStream_t s = (hipStream_t)stream;
    (void)geom_bytes;
    GeomView g = gsr_carve_geom(cfg->variant, cfg->P, geom);
    // the arena layout is always derived from the capacity implied by binning_bytes (forward, backward and debug agree)
    const uint32_t cap = gsr_binning_capacity(cfg->variant, binning_bytes, cfg->W, cfg->H);
    if (cap < num_rendered) { gsr_set_error("binning buffer too small: holds %u instances, need %u", cap, num_rendered); return 1; }
    BinView b = gsr_carve_bin(cfg->variant, cap, cfg->W, cfg->H, binning);
    ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, img);
    if (im.bytes > img_bytes) { gsr_set_error("img buffer too small: %zu < %zu", img_bytes, im.bytes); return 1; }
    if (cfg->P == 0) {
        // the reference returns the zero-initialised outputs untouched when P == 0 (rasterize_points.cu:79-113)
        return 0;
    }
    // The depth order the geom arena was laid out for (which of sorted_idx / offsets / scan_tmp mean what): recorded there by the preprocess kernel of
    // the stage-1 call -- or of the gsr_forward whose overflow this call repairs -- and returned to the caller by gsr_forward_stage1_ex / gsr_forward_ex.
    // Handed back in `depth_order`, stage 2 is a pure enqueue again (the read-back below drained the stream on every two-stage forward); with
    // depth_order = 0 (the ABI <= 6 entry point) the record is read back from the arena; an arena without one is refused instead of being binned on a guess.
    uint32_t mode = depth_order;
    if (mode == 0u) {
        GSR_CHECK(hipMemcpyAsync(&mode, g.counters + GSR_CNT_MODE, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read depth-order record");
        GSR_CHECK(hipStreamSynchronize(s), "stage2 sync");
    }
    else if (cfg->debug) {
        // debug only (it synchronises anyway): the caller's word must be the record the preprocess kernel left -- a stale or mixed-up word would bin the
        // arena in the wrong layout (which of sorted_idx / offsets / scan_tmp mean what) without any error
        uint32_t rec = 0u;
        GSR_CHECK(hipMemcpyAsync(&rec, g.counters + GSR_CNT_MODE, sizeof(uint32_t), hipMemcpyDeviceToHost, s), "read depth-order record");
        GSR_CHECK(hipStreamSynchronize(s), "stage2 debug sync");
        if (rec != mode) { gsr_set_error("gsr_forward_stage2_ex: depth_order 0x%08x is not the record stage 1 left in the geom buffer (0x%08x)", mode, rec); return 1; }
    }
    if (mode != GSR_MODE_TILE && mode != GSR_MODE_GLOBAL) {
        gsr_set_error("geom buffer carries no depth-order record (0x%08x): it must come from gsr_forward_stage1 / gsr_forward of this library, unmodified", mode);
        return 1;
    }
    const bool global_order = mode == GSR_MODE_GLOBAL;
    { ProfScope ps(GSR_PROF_BINNING, s); if (gsr_launch_binning(cfg, g, b, im, num_rendered, nullptr, s, global_order)) return 1; }
    if (check_cull_agreement(cfg, g, s)) return 1;
    { ProfScope ps(GSR_PROF_BLEND_FWD, s); if (gsr_launch_blend_fwd(cfg, in, g, b, im, out, s, global_order)) return 1; }
    return 0;
}
extern "C" int gsr_forward_stage2(const gsr_cfg* cfg, const gsr_inputs* in, void* geom, size_t geom_bytes,
                                  void* binning, size_t binning_bytes, void* img, size_t img_bytes,
                                  uint32_t num_rendered, const gsr_outputs* out, void* stream)
{
    return gsr_forward_stage2_ex(cfg, in, geom, geom_bytes, binning, binning_bytes, img, img_bytes, num_rendered, 0u, out, stream);
}

// Largest instance count whose binning arena fits in `bytes` (inverse of gsr_binning_bytes).
extern "C" uint32_t gsr_binning_capacity(int32_t variant, size_t bytes, int32_t W, int32_t H)
{
    uint64_t lo = 0, hi = 0xFFFFFFFFull;
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (gsr_binning_bytes(variant, (uint32_t)mid, W, H) <= bytes) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}

// Single-call forward without a GPU idle gap: stage 2 is enqueued right behind stage 1 against a caller-provided
// binning arena of some CAPACITY (the kernels read the exact instance count from device memory), and the host only
// waits on an event recorded after stage 1 to learn num_rendered.  *overflow_host = 1 when num_rendered exceeds the
// capacity: the outputs are then incomplete and the caller must re-run gsr_forward_stage2 with a large-enough arena
// (geom is intact; PLANE callers must re-zero out_observe first).  Exact same results as stage1 + stage2 otherwise.
extern "C" int gsr_forward(const gsr_cfg* cfg, const gsr_inputs* in, void* geom, size_t geom_bytes,
                           void* binning, size_t binning_bytes, void* img, size_t img_bytes, int32_t* radii,
                           const gsr_outputs* out, uint32_t* num_rendered_host, int32_t* overflow_host, void* stream)
{
    if (check_cfg(cfg, in)) return 1;
    hipStream_t s = (hipStream_t)stream;
    *num_rendered_host = 0; *overflow_host = 0;
    if (cfg->P == 0) return 0;
    Mailbox* mb = mailbox();
    if (!mb) { gsr_set_error("gsr_forward: pinned mailbox unavailable, use stage1/stage2"); return 1; }
    GeomView g = gsr_carve_geom(cfg->variant, cfg->P, geom);
    if (g.bytes > geom_bytes) { gsr_set_error("geom buffer too small: %zu < %zu", geom_bytes, g.bytes); return 1; }
    const uint32_t cap = gsr_binning_capacity(cfg->variant, binning_bytes, cfg->W, cfg->H);
    if (cap == 0) { gsr_set_error("binning buffer too small"); return 1; }
    BinView b = gsr_carve_bin(cfg->variant, cap, cfg->W, cfg->H, binning);
    ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, img);
    if (im.bytes > img_bytes) { gsr_set_error("img buffer too small: %zu < %zu", img_bytes, im.bytes); return 1; }
    const unsigned slot = take_slot(mb);
    struct SlotGuard { Mailbox* m; unsigned s; ~SlotGuard() { release_slot(m, s); } } guard{mb, slot};
    // How the host learns num_rendered while stage 2 is already queued: k_duplicate (per-tile order) or the scan kernel (global order) stores it into
    // the slot's mapped pinned word (system-scope store) and the host POLLS that word -- preset to a sentinel -- instead of waiting on an event recorded
    // behind stage 1: the event record is a packet of its own and left ~6 us of stream idle time in every forward (profiles/r03_timeline_surfel.json).
    // The poll gives up after 2 s and synchronises the stream instead.
    volatile uint32_t* word = mb->host + 16 * slot;
    mb->host[16 * slot + 1] = 0u;                              // "a gaussian failed the frustum test although prefiltered is set"
    *word = 0xFFFFFFFFu; std::atomic_thread_fence(std::memory_order_seq_cst);
    const bool global_order = gsr_decide_depth_order(cfg);
    { ProfScope ps(GSR_PROF_PREPROCESS, s); if (gsr_launch_preprocess(cfg, in, g, radii, s, global_order, mb->dev + 16 * slot + 1)) return 1; }
    { ProfScope ps(GSR_PROF_DEPTH_ORDER, s); if (gsr_launch_depth_order(cfg, g, mb->dev + 16 * slot, s, global_order, false)) return 1; }
    { ProfScope ps(GSR_PROF_BINNING, s); if (gsr_launch_binning(cfg, g, b, im, cap, g.counters, s, global_order, mb->dev + 16 * slot)) return 1; }
    if (check_cull_agreement(cfg, g, s)) return 1;
    { ProfScope ps(GSR_PROF_BLEND_FWD, s); if (gsr_launch_blend_fwd(cfg, in, g, b, im, out, s, global_order)) return 1; }
    uint32_t R;
    {
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t spins = 0;
        while ((R = *word) == 0xFFFFFFFFu) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                GSR_CHECK(hipStreamSynchronize(s), "stage1 sync (mailbox poll timed out)");
                R = *word;
                if (R == 0xFFFFFFFFu) { gsr_set_error("gsr_forward: num_rendered never arrived"); return 1; }
                break;
            }
        }
    }
    *num_rendered_host = R;
    *overflow_host = (R > cap) ? 1 : 0;
    // the preprocess has finished by now (num_rendered comes from a kernel behind it), so its word is final
    if (cfg->prefiltered && *(volatile uint32_t*)(mb->host + 16 * slot + 1)) { gsr_set_error("%s", GSR_PREFILTERED_MSG); return 1; }
    return 0;
}

// No host synchronisation, no host-visible state: every call is a fixed sequence of launches on `stream` whose shapes depend only on cfg and the
// arena capacities -- the form a HIP graph can record and replay.
extern "C" int gsr_forward_async(const gsr_cfg* cfg, const gsr_inputs* in, void* geom, size_t geom_bytes,
                                 void* binning, size_t binning_bytes, void* img, size_t img_bytes, int32_t* radii,
                                 const gsr_outputs* out, uint32_t* status_dev, void* stream)
{
    if (check_cfg(cfg, in)) return 1;
    if (!status_dev) { gsr_set_error("gsr_forward_async: status_dev must be provided"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    if (cfg->P == 0) { if (gsr_memset_async(status_dev, 0, sizeof(uint32_t), s)) { gsr_set_error("status"); return 1; }; return 0; }
    // cfg->prefiltered: a gaussian that fails the frustum test sets status_dev[2] (sticky, like the overflow word): the reference traps the device
    // (auxiliary.h:156-160), the synchronous forwards fail the call, this one reports it where the caller looks when it synchronises
    GeomView g = gsr_carve_geom(cfg->variant, cfg->P, geom);
    if (g.bytes > geom_bytes) { gsr_set_error("geom buffer too small: %zu < %zu", geom_bytes, g.bytes); return 1; }
    const uint32_t cap = gsr_binning_capacity(cfg->variant, binning_bytes, cfg->W, cfg->H);
    if (cap == 0) { gsr_set_error("binning buffer too small"); return 1; }
    BinView b = gsr_carve_bin(cfg->variant, cap, cfg->W, cfg->H, binning);
    ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, img);
    if (im.bytes > img_bytes) { gsr_set_error("img buffer too small: %zu < %zu", img_bytes, im.bytes); return 1; }
    const bool global_order = gsr_decide_depth_order(cfg);      // a recorded graph replays the decision of its capture
    if (gsr_launch_preprocess(cfg, in, g, radii, s, global_order, status_dev + 2)) return 1;
    if (gsr_launch_depth_order(cfg, g, nullptr, s, global_order, false)) return 1;
    if (gsr_launch_binning(cfg, g, b, im, cap, g.counters, s, global_order, nullptr)) return 1;
    // status_dev[0] <- num_rendered, status_dev[1] <- 1 when it exceeds the capacity (sticky: never cleared here): written by the blend forward's
    // first workgroup -- after the binning, whose k_duplicate publishes the total
    if (gsr_launch_blend_fwd(cfg, in, g, b, im, out, s, global_order, status_dev, cap)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------ backward
extern "C" int gsr_backward_ex(const gsr_cfg* cfg, const gsr_inputs* in, const int32_t* radii,
                               const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes,
                               const void* img, size_t img_bytes, uint32_t num_rendered,
                               void* scratch, size_t scratch_bytes,
                               const gsr_out_grads* og, const gsr_in_grads* ig, uint32_t flags, void* stream)
{
    if (check_cfg(cfg, in)) return 1;
    (void)geom_bytes; (void)img_bytes;
    hipStream_t s = (hipStream_t)stream;
    if (cfg->P == 0) return 0;
    const size_t need = gsr_backward_scratch_bytes(cfg->variant, cfg->P);
    if (scratch_bytes < need) { gsr_set_error("backward scratch too small: %zu < %zu", scratch_bytes, need); return 1; }
    if (cfg->variant == GSR_PLANE && cfg->render_geo && !og->all_map_pixels) { gsr_set_error("PLANE backward needs all_map_pixels"); return 1; }
    GeomView g = gsr_carve_geom(cfg->variant, cfg->P, const_cast<void*>(geom));
    BinView b = gsr_carve_bin(cfg->variant, gsr_binning_capacity(cfg->variant, binning_bytes, cfg->W, cfg->H), cfg->W, cfg->H,
                              const_cast<void*>(binning));
    ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, const_cast<void*>(img));
    float* acc = reinterpret_cast<float*>(scratch);
    if (!(flags & GSR_BWD_SCRATCH_IS_ZERO)) { ProfScope ps(GSR_PROF_BWD_MEMSET, s); if (gsr_memset_async(acc, 0, need, s)) { gsr_set_error("memset acc"); return 1; }; }
    if (num_rendered > 0) {
        if (gsr_blend_bwd_is_sp() && ((g_prof_on.load(std::memory_order_relaxed) >> GSR_PROF_BLEND_BWD) & 1u)) {
            // one kernel: its own dispatch carries the two events (no event-record packets in front of and behind the dominant kernel)
            ProfRec r;
            {
                std::lock_guard<std::mutex> lk(g_prof_mu);
                if (g_prof.size() >= 8192) prof_drain();
                if (!g_prof_free.empty()) { r = g_prof_free.back(); g_prof_free.pop_back(); }
                else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
            }
            r.label = GSR_PROF_BLEND_BWD;
            gsr_blend_bwd_attach_events(r.a, r.b);
            const int rc = gsr_launch_blend_bwd(cfg, in, g, b, im, og, acc, s);
            { std::lock_guard<std::mutex> lk(g_prof_mu); g_prof.push_back(r); }
            if (rc) return 1;
        } else {
            ProfScope ps(GSR_PROF_BLEND_BWD, s);
            if (gsr_launch_blend_bwd(cfg, in, g, b, im, og, acc, s)) return 1;
        }
    }
    { ProfScope ps(GSR_PROF_PREPROCESS_BWD, s); if (gsr_launch_preprocess_bwd(cfg, in, radii, g, acc, ig, (flags & GSR_BWD_LEAVE_ZERO) != 0, s)) return 1; }
    return 0;
}

extern "C" int gsr_backward(const gsr_cfg* cfg, const gsr_inputs* in, const int32_t* radii,
                            const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes,
                            const void* img, size_t img_bytes, uint32_t num_rendered,
                            void* scratch, size_t scratch_bytes,
                            const gsr_out_grads* og, const gsr_in_grads* ig, void* stream)
{
    return gsr_backward_ex(cfg, in, radii, geom, geom_bytes, binning, binning_bytes, img, img_bytes, num_rendered, scratch, scratch_bytes, og, ig, 0u, stream);
}

// ------------------------------------------------------------------------------------------------ debug reads
__global__ void k_debug_ranges_view(uint2* r, int T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T && r[i].x >= r[i].y) r[i] = make_uint2(0u, 0u);
}
// the tile id of every instance, rebuilt from the ranges: BinView::tile_keys itself is only written by the two-pass radix path (gsr_binning.hip)
__global__ void __launch_bounds__(256) k_debug_tile_keys(const uint2* __restrict__ ranges, int T, uint32_t R, uint32_t* __restrict__ dst)
{
    const int t = blockIdx.x;
    if (t >= T) return;
    const uint2 r = ranges[t];
    for (uint32_t e = r.x + threadIdx.x; e < r.y && e < R; e += 256u) dst[e] = (uint32_t)t;
}

extern "C" int gsr_debug_read(const gsr_cfg* cfg, int32_t field, const void* geom, const void* binning, size_t binning_bytes,
                              const void* img, uint32_t num_rendered, void* dst, void* stream)
{
    const uint32_t bcap = gsr_binning_capacity(cfg->variant, binning_bytes, cfg->W, cfg->H);
    hipStream_t s = (hipStream_t)stream;
    const size_t N = (size_t)cfg->W * cfg->H;
    const int gx = (cfg->W + GSR_TILE - 1) / GSR_TILE, gy = (cfg->H + GSR_TILE - 1) / GSR_TILE;
    const void* src = nullptr; size_t bytes = 0;
    switch (field) {
    case GSR_DBG_TILES_TOUCHED: { GeomView g = gsr_carve_geom(cfg->variant, cfg->P, const_cast<void*>(geom)); src = g.tiles_touched; bytes = (size_t)cfg->P * 4; break; }
    case GSR_DBG_POINT_LIST: { BinView b = gsr_carve_bin(cfg->variant, bcap, cfg->W, cfg->H, const_cast<void*>(binning)); src = b.point_list; bytes = (size_t)num_rendered * 4; break; }
    case GSR_DBG_TILE_KEYS: {
        ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, const_cast<void*>(img));
        if (num_rendered) hipLaunchKernelGGL(k_debug_tile_keys, dim3(gx * gy), dim3(256), 0, s, im.ranges, gx * gy, num_rendered, (uint32_t*)dst);
        return gsr_check_launch("debug tile keys", s, cfg->debug);
    }
    case GSR_DBG_RANGES: { ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, const_cast<void*>(img)); src = im.ranges; bytes = (size_t)gx * gy * 8; break; }
    case GSR_DBG_FINAL_T: { ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, const_cast<void*>(img)); src = im.final_T; bytes = N * 4 * (cfg->variant == GSR_SURFEL ? 3 : 1); break; }
    case GSR_DBG_N_CONTRIB: { ImgView im = gsr_carve_img(cfg->variant, cfg->W, cfg->H, const_cast<void*>(img)); src = im.n_contrib; bytes = N * 4 * (cfg->variant == GSR_SURFEL ? 2 : 1); break; }
    default: gsr_set_error("unknown debug field %d", field); return 1;
    }
    if (bytes) GSR_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s), "debug copy");
    // a tile without instances is held as (0xFFFFFFFF, 0) (two-pass radix path of gsr_binning.hip) or as (start, start) (one-pass bucket sort); the
    // reference's view of it is (0, 0) (cudaMemset, rasterizer_impl.cu:310)
    if (field == GSR_DBG_RANGES && bytes) hipLaunchKernelGGL(k_debug_ranges_view, dim3((gx * gy + 255) / 256), dim3(256), 0, s, (uint2*)dst, gx * gy);
    return 0;
}
