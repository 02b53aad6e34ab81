// f3dg_api.hip -- the C ABI (include/f3dg.h): workspace carving and launch orchestration.
//
// Host-side counterpart of CudaRasterizer::Rasterizer::forward (reference
// RAST/cuda_rasterizer/rasterizer_impl.cu:247-405) and of the chunk carving in :188-243 / rasterizer_impl.h:23-89,
// re-designed so that a call never blocks: the instance count stays on the device, the workspace is sized once
// for a capacity, and all views of a batch go through one launch sequence.
#include "f3dg_common.h"

#include <stdio.h>
#include <string.h>
#include <time.h>
#include <mutex>
#include <vector>

namespace {

thread_local char g_last_error[512] = "";

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__global__ void init_header_kernel(F3dgHeader* hdr, unsigned capacity, unsigned alpha_fast = 0, unsigned save_aux = 0)
{
    // (calls without Gaussians; otherwise the projection kernel's first workgroup initialises the header: F3dgHeaderInit)
    if (threadIdx.x < 64) {
        unsigned* w = reinterpret_cast<unsigned*>(hdr);
        w[threadIdx.x] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) { hdr->capacity = capacity; hdr->alpha_fast = alpha_fast; hdr->save_aux = save_aux; }
}

__global__ void fill_background_kernel(int V, size_t HW, const float* __restrict__ bg, int bg_per_view,
                                       float* __restrict__ out)
{
    // P == 0: the reference skips the rasterizer and returns the zero-initialised tensor (rasterize_points.cu:72,85)
    const size_t n = (size_t)V * F3DG_OUT_CHANNELS * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = 0.0f;
    (void)bg; (void)bg_per_view;
}

// debug export: the final list's Gaussian ids (the quadrant masks of F3DG_ID_BITS stripped)
__global__ void export_ids_kernel(const unsigned* __restrict__ src, size_t n, unsigned* __restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i] & F3DG_ID_MASK;
}

// ---- optional per-stage timing with HIP events on the caller's stream (bench.py's live roofline figure) ----
enum { ST_PREPROCESS = 0, ST_BINNING, ST_RENDER, ST_COUNT };
enum { BW_RENDER = 0, BW_GAUSSIAN, BW_COUNT };       // backward stages, recorded by f3dg_backward
struct ProfCall { hipEvent_t ev[ST_COUNT + 1]; };
struct ProfBwd { hipEvent_t ev[BW_COUNT + 1]; };
struct Prof {
    bool enabled = false;
    std::vector<ProfCall> calls;     // events recorded since the last collect
    std::vector<ProfCall> pool;      // recycled events
    std::vector<ProfBwd> bwd;        // backward calls recorded since the last collect
};
thread_local Prof g_prof;

ProfCall* prof_begin(hipStream_t s)
{
    if (!g_prof.enabled) return nullptr;
    ProfCall c;
    if (!g_prof.pool.empty()) { c = g_prof.pool.back(); g_prof.pool.pop_back(); }
    else for (int i = 0; i <= ST_COUNT; i++) if (hipEventCreate(&c.ev[i]) != hipSuccess) return nullptr;
    g_prof.calls.push_back(c);
    (void)hipEventRecord(c.ev[0], s);
    return &g_prof.calls.back();
}
inline void prof_mark(ProfCall* c, int stage_done, hipStream_t s) { if (c) (void)hipEventRecord(c->ev[stage_done + 1], s); }

} // namespace

// backward stage events (f3dg_backward.hip): begin returns a slot or -1; mark(slot, k) closes stage k
int f3dg_prof_bwd_begin(hipStream_t s)
{
    if (!g_prof.enabled) return -1;
    ProfBwd b;
    for (int i = 0; i <= BW_COUNT; i++) if (hipEventCreate(&b.ev[i]) != hipSuccess) return -1;
    g_prof.bwd.push_back(b);
    (void)hipEventRecord(b.ev[0], s);
    return (int)g_prof.bwd.size() - 1;
}
void f3dg_prof_bwd_mark(int slot, int stage_done, hipStream_t s)
{
    if (slot >= 0 && slot < (int)g_prof.bwd.size()) (void)hipEventRecord(g_prof.bwd[slot].ev[stage_done + 1], s);
}

int g_f3dg_small_path = 1;
int g_f3dg_small_path_aux = 1;       // forwards that keep the auxiliary planes (a backward follows) may take the small-call path too
int g_f3dg_small_debug = 0;
namespace {
// shapes (P, n_views, W, H) whose small-call path overflowed a tile list: they take the general path from then on
// (the library is called from several host threads -- ctypes releases the GIL -- so the process-global tables take a mutex)
struct SmallShape { unsigned v[4]; };
std::vector<SmallShape> g_small_disabled;
std::mutex g_small_mutex, g_sites_mutex;
bool small_disabled(unsigned P, unsigned V, unsigned W, unsigned H)
{
    std::lock_guard<std::mutex> lock(g_small_mutex);
    for (const SmallShape& d : g_small_disabled)
        if (d.v[0] == P && d.v[1] == V && d.v[2] == W && d.v[3] == H) return true;
    return false;
}
void small_disable(unsigned P, unsigned V, unsigned W, unsigned H)
{
    std::lock_guard<std::mutex> lock(g_small_mutex);
    for (const SmallShape& d : g_small_disabled)
        if (d.v[0] == P && d.v[1] == V && d.v[2] == W && d.v[3] == H) return;
    g_small_disabled.push_back({{P, V, W, H}});
}
} // namespace
std::atomic<unsigned long long> g_f3dg_kernel_launches{0};      // host-side counter of F3DG_KLAUNCH
int g_f3dg_render_pretest = 1;
int g_f3dg_render_cull = 1;
int g_f3dg_render_queue = 1;
int g_f3dg_render_fast = 1;
int g_f3dg_render_kernel = 3;
int g_f3dg_render_pack = -1;
int g_f3dg_render_dma = 1;
int g_f3dg_render_replay = 0;
int g_f3dg_render_wpb = 1;
int g_f3dg_render_count = 0;
#define F3DG_RENDER_TAIL_DEFAULT 0
int g_f3dg_render_tail = F3DG_RENDER_TAIL_DEFAULT;
int g_f3dg_render_slide = 1;
int g_f3dg_render_lowocc = 1;
#define F3DG_RENDER_UNROLL_DEFAULT -1
#define F3DG_RENDER_SPLIT_DEFAULT -1
int g_f3dg_render_split = F3DG_RENDER_SPLIT_DEFAULT;
int g_f3dg_render_unroll = F3DG_RENDER_UNROLL_DEFAULT;
int g_f3dg_render_lds_pad = 0;
int g_f3dg_bwd_occ = 5;
int g_f3dg_render_round = 192;
int g_f3dg_sort_wide_groups = 0;
int g_f3dg_sort_fused_rects = 0;
int g_f3dg_pre_hoist = 0;
int g_f3dg_tile_cull = 1;            // instantiate a Gaussian only in the tiles its conservative ellipse reaches (0: the reference's tile lists)
int g_f3dg_pre_order = 0;             // projection grid: 0 view-major (a view's chunks follow each other), 1 chunk-major (a chunk's views do)
int g_f3dg_debug_skip_all = 0;       // experiment switch: pre-test threshold = +inf (measures the loop skeleton)

extern "C" int f3dg_set_option(const char* name, int value)
{
    if (!name) return F3DG_ERR_BAD_ARG;
    // ---- the options of the library (process-wide DEFAULTS: what a call's own flags do not say; include/f3dg.h)
    if (strcmp(name, "render_fast") == 0) { g_f3dg_render_fast = value < 0 ? 0 : value > 2 ? 2 : value; return F3DG_OK; }
    if (strcmp(name, "tile_cull") == 0) { g_f3dg_tile_cull = value != 0; return F3DG_OK; }
    if (strcmp(name, "small_path") == 0) { g_f3dg_small_path = value != 0; if (value == 2) { std::lock_guard<std::mutex> lock(g_small_mutex); g_small_disabled.clear(); } return F3DG_OK; }
    if (strcmp(name, "small_path_aux") == 0) { g_f3dg_small_path_aux = value != 0; return F3DG_OK; }
    if (strcmp(name, "render_pack") == 0) { g_f3dg_render_pack = value < 0 ? -1 : value != 0; return F3DG_OK; }
    if (strcmp(name, "render_pack_th") == 0) { g_f3dg_render_pack_th = value < 0 ? 0 : value > 64 ? 64 : value; return F3DG_OK; }
    if (strcmp(name, "render_scan") == 0) { g_f3dg_render_scan = value < 0 ? -1 : value != 0; return F3DG_OK; }
    if (strcmp(name, "render_scan_min") == 0) { g_f3dg_render_scan_min = value < 0 ? 0 : value > 64 ? 64 : value; return F3DG_OK; }
    if (strcmp(name, "render_scan_th") == 0) { g_f3dg_render_scan_th = value < 0 ? 0 : value > 64 ? 64 : value; return F3DG_OK; }
    if (strcmp(name, "render_lowocc") == 0) { g_f3dg_render_lowocc = value < 0 ? 1 : value > 64 ? 64 : value; return F3DG_OK; }
    if (strcmp(name, "render_unroll") == 0) { g_f3dg_render_unroll = value < 0 ? F3DG_RENDER_UNROLL_DEFAULT : value < 1 ? 1 : value > 2 ? 2 : value; return F3DG_OK; }
    if (strcmp(name, "bwd_dense") == 0) { g_f3dg_bwd_dense = value != 0; return F3DG_OK; }
    if (strcmp(name, "bwd_occ") == 0) { g_f3dg_bwd_occ = (value >= 2 && value <= 6) ? value : 5; return F3DG_OK; }
    // diagnostics
    if (strcmp(name, "render_count") == 0) { g_f3dg_render_count = value != 0; return F3DG_OK; }
    if (strcmp(name, "time_launches") == 0) { g_f3dg_time_launches = value != 0; return F3DG_OK; }
#ifndef F3DG_LAB
    // small launches: 1 = producer + consumer waves (render3p), 2 / 3 = consumer + evaluators + producer (render3q, one view); -1 / 0 = by arithmetic
    if (strcmp(name, "render_split") == 0) { g_f3dg_render_split = value < 1 ? F3DG_RENDER_SPLIT_DEFAULT : value > 3 ? 3 : value; return F3DG_OK; }
#else
    // ---- lab builds (-DF3DG_LAB: builder-side experiments; the default library neither compiles the kernels behind these nor knows the names)
    if (strcmp(name, "render_split") == 0) { g_f3dg_render_split = value < 0 ? F3DG_RENDER_SPLIT_DEFAULT : value > 3 ? 3 : value; return F3DG_OK; }   // 0: render3l
    if (strcmp(name, "render_kernel") == 0) {       // 1 / 2 / 3: the kernel generations; 4: shorthand for render3s + the packed kernel everywhere
        g_f3dg_render_kernel = value == 1 ? 1 : value == 2 ? 2 : 3;
        g_f3dg_render_pack = value == 4 ? 1 : -1;   // (leaving the shorthand restores the default: ADVICE r05)
        return F3DG_OK;
    }
    if (strcmp(name, "tile_split") == 0) { g_f3dg_tile_split = value != 0; return F3DG_OK; }
    if (strcmp(name, "render_scan_lanes") == 0) { g_f3dg_render_scan_lanes = value == 2 ? 2 : 4; return F3DG_OK; }
    if (strcmp(name, "render_pretest") == 0) { g_f3dg_render_pretest = value != 0; return F3DG_OK; }
    if (strcmp(name, "render_queue") == 0) { g_f3dg_render_queue = value != 0; return F3DG_OK; }
    if (strcmp(name, "render_cull") == 0) { g_f3dg_render_cull = value != 0; return F3DG_OK; }
    if (strcmp(name, "render_slide") == 0) { g_f3dg_render_slide = value != 0; return F3DG_OK; }
    if (strcmp(name, "render_tail") == 0) { g_f3dg_render_tail = value < 0 ? F3DG_RENDER_TAIL_DEFAULT : value > 64 ? 64 : value; return F3DG_OK; }
    if (strcmp(name, "render_wpb") == 0) { g_f3dg_render_wpb = value == 4 ? 4 : 1; return F3DG_OK; }
    if (strcmp(name, "render_dma") == 0) { g_f3dg_render_dma = value != 0; return F3DG_OK; }
    if (strcmp(name, "render_round") == 0) { g_f3dg_render_round = value == 256 ? 256 : 192; return F3DG_OK; }
    if (strcmp(name, "render_lds_pad") == 0) { g_f3dg_render_lds_pad = value < 0 ? 0 : value; return F3DG_OK; }
    if (strcmp(name, "render_replay") == 0) { g_f3dg_render_replay = value; return F3DG_OK; }
    if (strcmp(name, "small_debug") == 0) { g_f3dg_small_debug = value; return F3DG_OK; }
    if (strcmp(name, "sort_wide_groups") == 0) { g_f3dg_sort_wide_groups = value != 0; return F3DG_OK; }
    if (strcmp(name, "sort_fused_rects") == 0) { g_f3dg_sort_fused_rects = value != 0; return F3DG_OK; }
    if (strcmp(name, "pre_hoist") == 0) { g_f3dg_pre_hoist = value != 0; return F3DG_OK; }
    if (strcmp(name, "pre_order") == 0) { g_f3dg_pre_order = value & 3; return F3DG_OK; }
    if (strcmp(name, "debug_skip_all") == 0) { g_f3dg_debug_skip_all = value != 0; return F3DG_OK; }
#endif
    return F3DG_ERR_BAD_ARG;
}

int g_f3dg_time_launches = 0;
namespace { struct LaunchSite { const char* file; int line; long long ns, n, max_ns; }; std::vector<LaunchSite> g_sites; }
long long f3dg_now_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
void f3dg_note_launch_time(const char* file, int line, long long ns)
{
    std::lock_guard<std::mutex> lock(g_sites_mutex);
    for (LaunchSite& s : g_sites)
        if (s.file == file && s.line == line) { s.ns += ns; s.n++; if (ns > s.max_ns) s.max_ns = ns; return; }
    g_sites.push_back({file, line, ns, 1, ns});
}
// diagnostic: prints (stderr) the host time spent inside hipLaunchKernelGGL per launch site since the last reset
extern "C" int f3dg_debug_launch_times(int reset)
{
    std::lock_guard<std::mutex> lock(g_sites_mutex);
    for (const LaunchSite& s : g_sites) {
        const char* base = strrchr(s.file, '/');
        fprintf(stderr, "%-22s:%4d  %6lld launches  %8.2f us avg  %9.1f us max\n", base ? base + 1 : s.file, s.line, s.n,
                1e-3 * (double)s.ns / (double)s.n, 1e-3 * (double)s.max_ns);
    }
    if (reset) g_sites.clear();
    return F3DG_OK;
}

extern "C" long long f3dg_debug_launch_count(int reset)
{
    return (long long)(reset ? g_f3dg_kernel_launches.exchange(0ull) : g_f3dg_kernel_launches.load());
}

extern "C" int f3dg_profile_enable(int on)
{
    g_prof.enabled = on != 0;
    return F3DG_OK;
}

// BLOCKING: waits for the recorded events, adds up per-stage milliseconds of every forward call recorded since
// the previous collect: h_stage_ms[0] preprocess, [1] binning (scan + keys + sort + ranges), [2] compositing, and of every
// f3dg_backward call: [3] compositing backward, [4] per-Gaussian backward. h_stage_ms holds FIVE doubles.
extern "C" int f3dg_profile_collect(double* h_stage_ms, int* h_calls) { return f3dg_profile_collect_calls(h_stage_ms, h_calls, nullptr, 0); }

// The same, and the three forward stage times of every recorded call, in call order, in h_per_call[max_calls][3] (calls beyond
// max_calls only enter the sums): what bench.py's per-launch min / median / max come from.
extern "C" int f3dg_profile_collect_calls(double* h_stage_ms, int* h_calls, double* h_per_call, int max_calls)
{
    double sum[ST_COUNT] = { 0, 0, 0 };
    int k = 0;
    for (ProfCall& c : g_prof.calls) {
        F3DG_HIP_CHECK(hipEventSynchronize(c.ev[ST_COUNT]));
        for (int i = 0; i < ST_COUNT; i++) {
            float ms = 0;
            F3DG_HIP_CHECK(hipEventElapsedTime(&ms, c.ev[i], c.ev[i + 1]));
            sum[i] += ms;
            if (h_per_call && k < max_calls) h_per_call[3 * k + i] = ms;
        }
        k++;
        g_prof.pool.push_back(c);
    }
    double bsum[BW_COUNT] = { 0, 0 };
    for (ProfBwd& b : g_prof.bwd) {
        F3DG_HIP_CHECK(hipEventSynchronize(b.ev[BW_COUNT]));
        for (int i = 0; i < BW_COUNT; i++) {
            float ms = 0;
            F3DG_HIP_CHECK(hipEventElapsedTime(&ms, b.ev[i], b.ev[i + 1]));
            bsum[i] += ms;
        }
        for (int i = 0; i <= BW_COUNT; i++) (void)hipEventDestroy(b.ev[i]);
    }
    g_prof.bwd.clear();
    if (h_calls) *h_calls = (int)g_prof.calls.size();
    g_prof.calls.clear();
    if (h_stage_ms) {
        for (int i = 0; i < ST_COUNT; i++) h_stage_ms[i] = sum[i];
        for (int i = 0; i < BW_COUNT; i++) h_stage_ms[ST_COUNT + i] = bsum[i];
    }
    return F3DG_OK;
}

int f3dg_set_hip_error(hipError_t e, const char* where)
{
    snprintf(g_last_error, sizeof g_last_error, "%s: %s", where, hipGetErrorString(e));
    return F3DG_ERR_HIP;
}

#ifdef F3DG_LAB
extern "C" const char* f3dg_version(void) { return "f3dg-hip gfx950 0.2.0 lab"; }
#else
extern "C" const char* f3dg_version(void) { return "f3dg-hip gfx950 0.2.0"; }
#endif
extern "C" const char* f3dg_last_error(void) { return g_last_error; }

int f3dg_sort_passes(int V, int T);

F3dgLayout f3dg_layout(int P, int W, int H, int V, long long cap)
{
    F3dgLayout L;
    memset(&L, 0, sizeof L);
    const size_t VP = (size_t)V * (size_t)(P > 0 ? P : 1);
    const size_t HW = (size_t)W * H;
    const size_t T = (size_t)((W + F3DG_TILE - 1) / F3DG_TILE) * ((H + F3DG_TILE - 1) / F3DG_TILE);
    const size_t C = (size_t)(cap > 0 ? cap : 1);
    L.sort_blocks = (unsigned)((C + F3DG_SORT_CHUNK - 1) / F3DG_SORT_CHUNK);
    const size_t scan_a = (VP + F3DG_SCAN_CHUNK - 1) / F3DG_SCAN_CHUNK;
    const size_t gsort_blocks = (size_t)V * (((size_t)(P > 0 ? P : 1) + F3DG_SORT_CHUNK - 1) / F3DG_SORT_CHUNK);      // (view, chunk) blocks of the depth sort
    const size_t tile_blocks = (size_t)L.sort_blocks + (size_t)V;      // (view, chunk) blocks of the tile pass: every view rounds up
    const size_t hist_blocks = gsort_blocks > tile_blocks ? gsort_blocks : tile_blocks;
    const size_t scan_b = ((size_t)256 * hist_blocks + F3DG_SCAN_CHUNK - 1) / F3DG_SCAN_CHUNK;
    L.scan_tmp_elems = (unsigned)((scan_a > scan_b ? scan_a : scan_b) + 1);

    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.header = take(sizeof(F3dgHeader));
    L.rec = take(VP * sizeof(F3dgRec));
    L.means2D = take(VP * sizeof(float2));
    L.bbox = take(VP * sizeof(float4));
    L.cull = take(VP * sizeof(float4));
    L.depths = take(VP * sizeof(float));
    L.conic = take(VP * sizeof(float4));
    L.radii = take(VP * sizeof(int));
    L.tiles = take(VP * sizeof(unsigned));
    L.offsets = take(VP * sizeof(unsigned));
    L.clamped = take(VP);
    L.rects = take(VP * sizeof(uint2));
    // (planes 4..6 exist for option sort_fused_rects only -- off by default, 3 x V x P x 4 bytes: 400 MB at C3's 512 views x 65,536)
    L.gsort = take((g_f3dg_sort_fused_rects ? 7 : 4) * VP * sizeof(unsigned));
    L.scan_tmp = take((size_t)L.scan_tmp_elems * sizeof(unsigned));
    L.keys[0] = take(C * 8);
    L.keys[1] = take(C * 4);
    L.vals[0] = take(C * 4);
    L.vals[1] = take(C * 4);
    L.segtab_minmax = (unsigned)((size_t)2 * (V + 1) + (size_t)8 * (V / 8 + 2));      // followed by minmax[2 V], chunk_minmax[2 V cps]
    L.segtab = take(((size_t)L.segtab_minmax + (size_t)2 * V + 2 * gsort_blocks) * sizeof(unsigned));   // + the chunks' ranges
    L.hist = take((size_t)256 * hist_blocks * sizeof(unsigned));
    L.ranges = take((size_t)V * T * sizeof(uint2));
    L.final_T = take((size_t)V * 4 * HW * sizeof(float));
    L.n_contrib = take((size_t)V * 2 * HW * sizeof(unsigned));
    L.bwd_acc = take(VP * 16 * sizeof(double));      // 128 bytes per (view, Gaussian): ten float64 sums + (dense backward) seven float32 ones in ONE line
    L.small_cap = f3dg_small_shape(P, W, H, V) ? (unsigned)F3DG_SMALL_CAP : 0u;
    L.small_boxes = take(L.small_cap ? (size_t)V * ((P + 63) / 64) * sizeof(uint2) : 0);
    L.small_cnt = take((size_t)V * T * sizeof(unsigned));
    L.small_list = take((size_t)V * T * L.small_cap * sizeof(unsigned));
    L.total = off;
    return L;
}

extern "C" size_t f3dg_workspace_bytes(int P, int W, int H, int n_views, long long max_rendered)
{
    if (P < 0 || W <= 0 || H <= 0 || n_views <= 0 || max_rendered < 0) return 0;
    return f3dg_layout(P, W, H, n_views, max_rendered).total;
}

namespace {

int check_gaussian_args(int P, int D, int M, const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                        const float* view2gaussian_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos)
{
    (void)P;
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos) return F3DG_ERR_BAD_ARG;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return F3DG_ERR_BAD_ARG;       // exactly one (rast_py:205)
    const bool have_sr = scales != nullptr && rotations != nullptr;
    if (have_sr == (cov3D_precomp != nullptr)) return F3DG_ERR_BAD_ARG;                 // exactly one (rast_py:208)
    if (!have_sr && view2gaussian_precomp == nullptr) return F3DG_ERR_BAD_ARG;          // view2gaussian needs scale/rot
    if (shs && (M <= 0 || D < 0 || D > 3 || (D + 1) * (D + 1) > M)) return F3DG_ERR_BAD_ARG;
    return F3DG_OK;
}

// projection + binning of n_views views: everything of Rasterizer::forward / ::integrate before the tile kernel
int run_geometry(hipStream_t s, char* ws, const F3dgLayout& L, int n_views, int views_per_set, int P, int D, int M, int W, int H,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* view2gaussian_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size,
                 int* radii_used, int save_aux, int need_box, int tile_cull, ProfCall* prof, F3dgHeaderInit init, int small = 0)
{
    // option pre_hoist: the per-Gaussian scratch (96 bytes each) borrows the backward's float64 accumulator plane, which nothing touches
    // before f3dg_backward -- if it is large enough (80 bytes per (view, Gaussian): at least ~1.2 views per set) and a Gaussian serves
    // enough views for the extra pass to pay
    const int n_sets = views_per_set > 0 ? n_views / views_per_set : 1;
    float4* hoist = nullptr;
    if (g_f3dg_pre_hoist && n_views / n_sets >= 4 && (size_t)n_views * 80 >= (size_t)n_sets * 96)
        hoist = reinterpret_cast<float4*>(ws + L.bwd_acc);
    int rc = f3dg_launch_preprocess(s, n_views, views_per_set, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs,
                                    cov3D_precomp, colors_precomp, view2gaussian_precomp, viewmatrix, projmatrix,
                                    cam_pos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, kernel_size,
                                    reinterpret_cast<F3dgRec*>(ws + L.rec), reinterpret_cast<float2*>(ws + L.means2D),
                                    reinterpret_cast<float*>(ws + L.depths), reinterpret_cast<unsigned*>(ws + L.gsort),
                                    reinterpret_cast<uint2*>(ws + L.rects),
                                    need_box ? reinterpret_cast<float4*>(ws + L.bbox) : nullptr,
                                    reinterpret_cast<float4*>(ws + L.cull),
                                    reinterpret_cast<float4*>(ws + L.conic), radii_used,
                                    reinterpret_cast<unsigned*>(ws + L.tiles),
                                    reinterpret_cast<unsigned char*>(ws + L.clamped), save_aux, tile_cull, init, hoist, n_sets,
                                    small ? reinterpret_cast<uint2*>(ws + L.small_boxes) : nullptr);
    if (rc != F3DG_OK) return rc;
    prof_mark(prof, ST_PREPROCESS, s);
    rc = small ? f3dg_launch_small_bin(s, n_views, P, W, H, L, ws) : f3dg_launch_binning(s, n_views, P, W, H, L, ws, save_aux);
    if (rc != F3DG_OK) return rc;
    prof_mark(prof, ST_BINNING, s);
    return F3DG_OK;
}

} // namespace

extern "C" int f3dg_forward_batched(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                    int n_views, int P, int D, int M,
                                    const float* background, int W, int H,
                                    const float* means3D, const float* shs, const float* colors_precomp,
                                    const float* opacities, const float* scales, float scale_modifier,
                                    const float* rotations, const float* cov3D_precomp,
                                    const float* view2gaussian_precomp,
                                    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                    float tan_fovx, float tan_fovy, float kernel_size,
                                    float* out_color, int* radii, unsigned flags)
{
    return f3dg_forward_sets(stream, workspace, workspace_bytes, max_rendered, 1, n_views, P, D, M, background, W, H, means3D, shs,
                             colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, view2gaussian_precomp,
                             viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, out_color, radii, flags);
}

extern "C" int f3dg_forward_sets(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                 int n_sets, int views_per_set, int P, int D, int M,
                                 const float* background, int W, int H,
                                 const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp,
                                 const float* view2gaussian_precomp,
                                 const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                 float tan_fovx, float tan_fovy, float kernel_size,
                                 float* out_color, int* radii, unsigned flags)
{
    if (n_sets <= 0 || views_per_set <= 0 || (long long)n_sets * views_per_set > 0x7FFFFFF0ll) return F3DG_ERR_BAD_ARG;
    const int n_views = n_sets * views_per_set;
    hipStream_t s = (hipStream_t)stream;
    if (n_views <= 0 || P < 0 || W <= 0 || H <= 0 || max_rendered < 0 || !out_color || !background || !workspace)
        return F3DG_ERR_BAD_ARG;
    if (max_rendered > 0xFFFFFFF0ll || (long long)n_views * P > 0xFFFFFFF0ll || P > (int)F3DG_ID_MASK)
        return F3DG_ERR_BAD_ARG;                       // instance and (view,Gaussian) indices are 32-bit, list entries hold 28-bit ids
    const F3dgLayout L = f3dg_layout(P, W, H, n_views, max_rendered);
    if (workspace_bytes < L.total) return F3DG_ERR_WORKSPACE;
    char* ws = static_cast<char*>(workspace);
    F3dgHeader* hdr = reinterpret_cast<F3dgHeader*>(ws + L.header);
    const size_t HW = (size_t)W * H;

    const int save_aux = (flags & F3DG_FLAG_SAVE_AUX) ? 1 : 0;
    // several Gaussian sets in one call are an inference path: f3dg_backward and the per-Gaussian backward index the Gaussian inputs
    // without a set offset, and view2gaussian_precomp is [n_views, P, 10] of ONE set
    if (n_sets > 1 && (save_aux || view2gaussian_precomp != nullptr)) return F3DG_ERR_BAD_ARG;
    // small-call path (f3dg_small.hip): inference calls of one or two views of a modest set go projection -> per-tile sort -> compositing
    // what this call runs with: the process-wide defaults of f3dg_set_option, overridden by the call's own flags
    const int fast = (flags & F3DG_FLAG_EXACT) ? 0 : (flags & F3DG_FLAG_FAST) ? 1 : f3dg_render_uses_fast(save_aux);
    const int tile_cull = (flags & F3DG_FLAG_NO_TILE_CULL) ? 0 : g_f3dg_tile_cull;
    const int small = g_f3dg_small_path && !(flags & F3DG_FLAG_NO_SMALL_PATH) && (!save_aux || g_f3dg_small_path_aux) && n_sets == 1 && P > 0 && L.small_cap != 0 &&
                      g_f3dg_render_kernel == 3 && !small_disabled((unsigned)P, (unsigned)n_views, (unsigned)W, (unsigned)H);
    // (the header is initialised by the first workgroup of the projection kernel; without Gaussians there is no such launch)
    const F3dgHeaderInit hinit = { hdr, (unsigned)max_rendered, (unsigned)fast, (unsigned)save_aux, (unsigned)small,
                                   { (unsigned)P, (unsigned)n_views, (unsigned)W, (unsigned)H } };
    if (P == 0)
        F3DG_KLAUNCH(init_header_kernel, dim3(1), dim3(64), 0, s, hdr, (unsigned)max_rendered, (unsigned)fast, (unsigned)save_aux);

    if (P == 0) {
        F3DG_KLAUNCH(fill_background_kernel, dim3(1024), dim3(256), 0, s, n_views, HW, background,
                           (flags & F3DG_FLAG_BG_PER_VIEW) ? 1 : 0, out_color);
        F3DG_HIP_CHECK(hipGetLastError());
        return F3DG_OK;
    }
    int rc = check_gaussian_args(P, D, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                 view2gaussian_precomp, viewmatrix, projmatrix, cam_pos);
    if (rc != F3DG_OK) return rc;

    const float focal_y = H / (2.0f * tan_fovy);       // float32, rasterizer_impl.cu:274-275
    const float focal_x = W / (2.0f * tan_fovx);
    int* radii_used = radii ? radii : reinterpret_cast<int*>(ws + L.radii);

    ProfCall* prof = prof_begin(s);
    rc = run_geometry(s, ws, L, n_views, views_per_set, P, D, M, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                      rotations, cov3D_precomp, view2gaussian_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                      tan_fovy, focal_x, focal_y, kernel_size, radii_used, save_aux,
                      save_aux || g_f3dg_render_kernel == 1 /* the culling box: backward + the pixel-lane kernel */, tile_cull, prof, hinit, small);
    if (rc != F3DG_OK) return rc;

    rc = f3dg_launch_render(s, n_views, P, W, H, focal_x, focal_y, hdr,
                              reinterpret_cast<const uint2*>(ws + L.ranges),
                              reinterpret_cast<const unsigned*>(ws + (small ? L.small_list : L.vals[0])),
                              reinterpret_cast<const F3dgRec*>(ws + L.rec),
                              reinterpret_cast<const float4*>(ws + L.bbox),
                              reinterpret_cast<const float4*>(ws + L.cull), background,
                              (flags & F3DG_FLAG_BG_PER_VIEW) ? 1 : 0, out_color,
                              reinterpret_cast<float*>(ws + L.final_T),
                              reinterpret_cast<unsigned*>(ws + L.n_contrib), save_aux,
                              flags & (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION), fast, (flags & F3DG_FLAG_SCAN) ? 1 : 0);
    prof_mark(prof, ST_RENDER, s);
    return rc;
}

extern "C" size_t f3dg_integrate_workspace_bytes(int P, int PN, int W, int H, long long max_rendered)
{
    if (P < 0 || PN < 0 || W <= 0 || H <= 0 || max_rendered < 0) return 0;
    return f3dg_integ_layout(P, PN, W, H, max_rendered).total;
}

extern "C" size_t f3dg_integrate_workspace_bytes_batched(int P, int PN, int W, int H, int n_views, long long max_rendered)
{
    if (P < 0 || PN < 0 || W <= 0 || H <= 0 || n_views <= 0 || max_rendered < 0) return 0;
    return f3dg_integ_layout(P, PN, W, H, max_rendered, n_views).total;
}

extern "C" long long f3dg_integrate_prepare_batched(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                                    int n_views, int PN_max, int P, int D, int M, const float* background, int W, int H,
                                                    const float* means3D, const float* shs, const float* colors_precomp,
                                                    const float* opacities, const float* scales, float scale_modifier,
                                                    const float* rotations, const float* cov3D_precomp,
                                                    const float* view2gaussian_precomp, const float* viewmatrix,
                                                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                                    float kernel_size, float* out_color, int* radii, long long* h_needed)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_views <= 0 || PN_max < 0 || P <= 0 || W <= 0 || H <= 0 || max_rendered < 0 || !out_color || !workspace || !background)
        return F3DG_ERR_BAD_ARG;
    if (max_rendered > 0xFFFFFFF0ll || (long long)n_views * P > 0xFFFFFFF0ll || P > (int)F3DG_ID_MASK) return F3DG_ERR_BAD_ARG;
    const F3dgLayout L = f3dg_layout(P, W, H, n_views, max_rendered);
    const F3dgIntegLayout I = f3dg_integ_layout(P, PN_max, W, H, max_rendered, n_views);
    if (workspace_bytes < I.total) return F3DG_ERR_WORKSPACE;
    char* ws = static_cast<char*>(workspace);
    F3dgHeader* hdr = reinterpret_cast<F3dgHeader*>(ws + L.header);
    if (h_needed) *h_needed = 0;
    const F3dgHeaderInit hinit = { hdr, (unsigned)max_rendered, 0u, 0u, 0u, { (unsigned)P, (unsigned)n_views, (unsigned)W, (unsigned)H } };
    int rc = check_gaussian_args(P, D, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                 view2gaussian_precomp, viewmatrix, projmatrix, cam_pos);
    if (rc != F3DG_OK) return rc;
    const float focal_y = H / (2.0f * tan_fovy);           // rasterizer_impl.cu:567-568
    const float focal_x = W / (2.0f * tan_fovx);
    int* radii_used = radii ? radii : reinterpret_cast<int*>(ws + L.radii);
    rc = run_geometry(s, ws, L, n_views, n_views, P, D, M, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                      rotations, cov3D_precomp, view2gaussian_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                      tan_fovy, focal_x, focal_y, kernel_size, radii_used, 0, 1 /* pass 1 culls by the box */,
                      0 /* the points of a tile are not its pixel centres: the reference's tile lists */, nullptr, hinit);
    if (rc != F3DG_OK) return rc;
    rc = f3dg_launch_integrate_pass1(s, n_views, P, W, H, focal_x, focal_y, L, I, ws, background, out_color);
    if (rc != F3DG_OK) return rc;
    long long n = 0;
    rc = f3dg_read_status(stream, workspace, &n);
    if (h_needed) *h_needed = n;
    if (rc != F3DG_OK) return rc;
    return n;
}

extern "C" long long f3dg_integrate_prepare(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                            int PN_max, int P, int D, int M, const float* background, int W, int H,
                                            const float* means3D, const float* shs, const float* colors_precomp,
                                            const float* opacities, const float* scales, float scale_modifier,
                                            const float* rotations, const float* cov3D_precomp,
                                            const float* view2gaussian_precomp, const float* viewmatrix,
                                            const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                            float kernel_size, float* out_color, int* radii, long long* h_needed)
{
    return f3dg_integrate_prepare_batched(stream, workspace, workspace_bytes, max_rendered, 1, PN_max, P, D, M, background, W, H, means3D,
                                          shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                                          view2gaussian_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size,
                                          out_color, radii, h_needed);
}

extern "C" int f3dg_integrate_points_view(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                          int n_views, int view, int PN, int P, int W, int H, const float* points3D,
                                          const float* viewmatrix, float tan_fovx, float tan_fovy, float* out_color,
                                          float* out_alpha_integrated, float* out_color_integrated, float* alpha_min)
{
    hipStream_t s = (hipStream_t)stream;
    if (PN < 0 || P <= 0 || W <= 0 || H <= 0 || max_rendered < 0 || !out_color || !workspace || !viewmatrix || n_views <= 0 ||
        view < 0 || view >= n_views)
        return F3DG_ERR_BAD_ARG;
    if (PN == 0) return F3DG_OK;
    if (!points3D) return F3DG_ERR_BAD_ARG;
    const F3dgLayout L = f3dg_layout(P, W, H, n_views, max_rendered);
    const F3dgIntegLayout I = f3dg_integ_layout(P, PN, W, H, max_rendered, n_views);
    if (workspace_bytes < I.total) return F3DG_ERR_WORKSPACE;
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    return f3dg_launch_integrate_points(s, view, P, PN, W, H, focal_x, focal_y, L, I, static_cast<char*>(workspace), points3D,
                                        viewmatrix, out_color, out_alpha_integrated, out_color_integrated, alpha_min);
}

extern "C" int f3dg_debug_integrate_redo(void* stream, const void* workspace, int P, int PN_max, int W, int H, int n_views,
                                         long long max_rendered, int* h_tiles)
{
    if (!workspace || !h_tiles || P <= 0 || W <= 0 || H <= 0 || n_views <= 0 || PN_max < 0 || max_rendered < 0) return F3DG_ERR_BAD_ARG;
    const F3dgIntegLayout I = f3dg_integ_layout(P, PN_max, W, H, max_rendered, n_views);
    const size_t T = (size_t)((W + F3DG_TILE - 1) / F3DG_TILE) * ((H + F3DG_TILE - 1) / F3DG_TILE);
    std::vector<unsigned> flags((size_t)n_views * T);
    F3DG_HIP_CHECK(hipMemcpyAsync(flags.data(), static_cast<const char*>(workspace) + I.redo, flags.size() * sizeof(unsigned),
                                  hipMemcpyDeviceToHost, (hipStream_t)stream));
    F3DG_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    int n = 0;
    for (unsigned f : flags) n += f != 0u;
    *h_tiles = n;
    return F3DG_OK;
}

extern "C" int f3dg_integrate_points(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                     int PN, int P, int W, int H, const float* points3D, const float* viewmatrix,
                                     float tan_fovx, float tan_fovy, float* out_color, float* out_alpha_integrated,
                                     float* out_color_integrated, float* alpha_min)
{
    return f3dg_integrate_points_view(stream, workspace, workspace_bytes, max_rendered, 1, 0, PN, P, W, H, points3D, viewmatrix, tan_fovx,
                                      tan_fovy, out_color, out_alpha_integrated, out_color_integrated, alpha_min);
}

extern "C" long long f3dg_integrate(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                    int PN, int P, int D, int M, const float* background, int W, int H,
                                    const float* points3D, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* scales,
                                    float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                    const float* view2gaussian_precomp, const float* viewmatrix,
                                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                    float kernel_size, const float* subpixel_offset, int prefiltered,
                                    float* out_color, int* radii, float* out_alpha_integrated,
                                    float* out_color_integrated, long long* h_needed)
{
    (void)subpixel_offset;   // forward.cu:838 loads it into `depth_input`, which nothing reads
    (void)prefiltered;       // only traps on a culled point (auxiliary.h:194-198)
    hipStream_t s = (hipStream_t)stream;
    if (PN < 0 || P < 0 || W <= 0 || H <= 0 || max_rendered < 0 || !out_color || !workspace) return F3DG_ERR_BAD_ARG;
    if (PN > 0 && (!points3D || !out_alpha_integrated || !out_color_integrated)) return F3DG_ERR_BAD_ARG;
    if (h_needed) *h_needed = 0;
    if (P == 0 || PN == 0) {                               // rasterize_points.cu:300: nothing runs, the fills stay
        if (workspace_bytes < sizeof(F3dgHeader)) return F3DG_ERR_WORKSPACE;
        F3DG_KLAUNCH(init_header_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<F3dgHeader*>(workspace), (unsigned)max_rendered);
        int rc = f3dg_launch_integrate_fill(s, W, H, PN, out_color, out_alpha_integrated, out_color_integrated);
        if (rc != F3DG_OK) return rc;
        F3DG_HIP_CHECK(hipStreamSynchronize(s));
        return 0;
    }
    // = prepare (projection + binning + the per-pixel pass) followed by the point stage on the same workspace
    const long long n = f3dg_integrate_prepare(stream, workspace, workspace_bytes, max_rendered, PN, P, D, M, background, W, H,
                                               means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                               cov3D_precomp, view2gaussian_precomp, viewmatrix, projmatrix, cam_pos,
                                               tan_fovx, tan_fovy, kernel_size, out_color, radii, h_needed);
    if (n < 0) return n;
    const int rc = f3dg_integrate_points(stream, workspace, workspace_bytes, max_rendered, PN, P, W, H, points3D, viewmatrix,
                                         tan_fovx, tan_fovy, out_color, out_alpha_integrated, out_color_integrated, nullptr);
    if (rc != F3DG_OK) return rc;
    return n;
}

extern "C" int f3dg_read_status(void* stream, const void* workspace, long long* h_num_rendered)
{
    if (!workspace) return F3DG_ERR_BAD_ARG;
    F3dgHeader h;
    F3DG_HIP_CHECK(hipMemcpyAsync(&h, workspace, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream));
    F3DG_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (h_num_rendered) *h_num_rendered = (long long)h.num_rendered;
    if (h.small_overflow) small_disable(h.small_shape[0], h.small_shape[1], h.small_shape[2], h.small_shape[3]);     // the retry takes the general path
    return h.overflow ? F3DG_ERR_OVERFLOW : F3DG_OK;
}

// ---- non-blocking status (f3dg.h: f3dg_status_post / f3dg_status_poll)
namespace {
struct StatusSlot { F3dgHeader* host; hipEvent_t ev; bool busy; };
std::vector<StatusSlot> g_status;       // grows on demand; slots are recycled
std::mutex g_status_mutex;
int finish_status(const F3dgHeader& h, long long* h_num_rendered)
{
    if (h_num_rendered) *h_num_rendered = (long long)h.num_rendered;
    if (h.small_overflow) small_disable(h.small_shape[0], h.small_shape[1], h.small_shape[2], h.small_shape[3]);     // the retry takes the general path
    return h.overflow ? F3DG_ERR_OVERFLOW : F3DG_OK;
}
} // namespace

extern "C" int f3dg_status_post(void* stream, const void* workspace)
{
    if (!workspace) return F3DG_ERR_BAD_ARG;
    int ticket = -1;
    StatusSlot sl = { nullptr, nullptr, false };
    {
        // the slot is COPIED under the lock (another thread's post may grow the vector), and a slot whose creation fails is not kept
        std::lock_guard<std::mutex> lock(g_status_mutex);
        for (size_t i = 0; i < g_status.size(); i++)
            if (!g_status[i].busy) { ticket = (int)i; break; }
        if (ticket < 0) {
            StatusSlot fresh = { nullptr, nullptr, false };
            F3DG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&fresh.host), sizeof(F3dgHeader), hipHostMallocDefault));
            const hipError_t e = hipEventCreateWithFlags(&fresh.ev, hipEventDisableTiming);
            if (e != hipSuccess) { (void)hipHostFree(fresh.host); return f3dg_set_hip_error(e, "hipEventCreateWithFlags"); }
            g_status.push_back(fresh);
            ticket = (int)g_status.size() - 1;
        }
        g_status[ticket].busy = true;
        sl = g_status[ticket];
    }
    hipError_t e = hipMemcpyAsync(sl.host, workspace, sizeof(F3dgHeader), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventRecord(sl.ev, (hipStream_t)stream);
    if (e != hipSuccess) {          // the slot goes back: nothing will ever complete it
        std::lock_guard<std::mutex> lock(g_status_mutex);
        g_status[ticket].busy = false;
        return f3dg_set_hip_error(e, "f3dg_status_post");
    }
    return ticket;
}

extern "C" int f3dg_status_poll(int ticket, int wait, long long* h_num_rendered)
{
    StatusSlot sl;
    {
        std::lock_guard<std::mutex> lock(g_status_mutex);
        if (ticket < 0 || ticket >= (int)g_status.size() || !g_status[ticket].busy) return F3DG_ERR_BAD_ARG;
        sl = g_status[ticket];
    }
    if (wait) F3DG_HIP_CHECK(hipEventSynchronize(sl.ev));
    else {
        const hipError_t q = hipEventQuery(sl.ev);
        if (q == hipErrorNotReady) return F3DG_PENDING;
        F3DG_HIP_CHECK(q);
    }
    const int rc = finish_status(*sl.host, h_num_rendered);
    std::lock_guard<std::mutex> lock(g_status_mutex);
    g_status[ticket].busy = false;
    return rc;
}

extern "C" int f3dg_backward_pairs(void* stream, const void* workspace, long long* h_pairs)
{
    if (!workspace || !h_pairs) return F3DG_ERR_BAD_ARG;
    F3dgHeader h;
    F3DG_HIP_CHECK(hipMemcpyAsync(&h, workspace, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream));
    F3DG_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *h_pairs = (long long)h.bwd_pairs;
    // the backward found a workspace whose last forward kept no auxiliary planes / took the small-call path: it walked nothing
    return h.bwd_stale ? F3DG_ERR_STATE : F3DG_OK;
}

extern "C" long long f3dg_forward(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                  int P, int D, int M, const float* background, int W, int H,
                                  const float* means3D, const float* shs, const float* colors_precomp,
                                  const float* opacities, const float* scales, float scale_modifier,
                                  const float* rotations, const float* cov3D_precomp,
                                  const float* view2gaussian_precomp,
                                  const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                  float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
                                  float* out_color, int* radii, unsigned flags, long long* h_needed)
{
    (void)prefiltered;   // the reference only uses it to trap on a culled point (auxiliary.h:194-198)
    int rc = f3dg_forward_batched(stream, workspace, workspace_bytes, max_rendered, 1, P, D, M, background, W, H,
                                  means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                  cov3D_precomp, view2gaussian_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                                  tan_fovy, kernel_size, out_color, radii, flags & ~F3DG_FLAG_BG_PER_VIEW);
    if (rc != F3DG_OK) return rc;
    long long n = 0;
    rc = f3dg_read_status(stream, workspace, &n);
    if (h_needed) *h_needed = n;
    if (rc != F3DG_OK) return rc;
    return n;
}

// ---- debug/inspection: copy internal per-Gaussian state to caller buffers (used by the stage-wise parity
// tests; mirrors the oracle's accessors). Any pointer may be NULL.
extern "C" int f3dg_debug_export(void* stream, const void* workspace, int P, int W, int H, int n_views,
                                 long long max_rendered, float* rec /*[V*P*16]*/, float* means2D /*[V*P*2]*/,
                                 float* conic /*[V*P*4]*/, unsigned* tiles /*[V*P]*/, unsigned* offsets /*[V*P]*/,
                                 unsigned char* clamped /*[V*P]*/, unsigned long long* keys_sorted /*[cap]*/,
                                 unsigned* point_list /*[cap]*/, unsigned* ranges /*[V*T*2]*/,
                                 float* final_T /*[V*4*HW]*/, unsigned* n_contrib /*[V*2*HW]*/, float* depths /*[V*P]*/)
{
    hipStream_t s = (hipStream_t)stream;
    if (!workspace) return F3DG_ERR_BAD_ARG;
    const F3dgLayout L = f3dg_layout(P, W, H, n_views, max_rendered);
    const char* ws = static_cast<const char*>(workspace);
    const size_t VP = (size_t)n_views * P, HW = (size_t)W * H;
    F3dgHeader h;       // (BLOCKING: an inspection hook)
    F3DG_HIP_CHECK(hipMemcpyAsync(&h, ws + L.header, sizeof h, hipMemcpyDeviceToHost, s));
    F3DG_HIP_CHECK(hipStreamSynchronize(s));
    // these planes are written by a SAVE_AUX forward only (the keys are rebuilt from `depths`): refuse to hand out stale memory
    if ((means2D || conic || tiles || offsets || clamped || keys_sorted || final_T || n_contrib || depths) && !h.save_aux)
        return F3DG_ERR_BAD_ARG;
    if (h.small_path) {
        // the small-call path keeps its lists in per-tile slots: hand them out in the general path's layout (gap-free, (view, tile) order)
        if (h.overflow) return F3DG_ERR_OVERFLOW;
        // (no instance offsets and no 64-bit keys on that path: a caller that inspects them renders with F3DG_FLAG_NO_SMALL_PATH)
        if (offsets || keys_sorted) return F3DG_ERR_STATE;
        const int rcs = f3dg_launch_small_export(s, n_views, W, H, L, ws, point_list, ranges);
        if (rcs != F3DG_OK) return rcs;
        point_list = nullptr;
        ranges = nullptr;
    }
    const size_t T = (size_t)((W + F3DG_TILE - 1) / F3DG_TILE) * ((H + F3DG_TILE - 1) / F3DG_TILE);
    const size_t C = (size_t)max_rendered;
    if (keys_sorted && P > 0) {      // the 64-bit keys are not kept by the forward: rebuild them from the final list
        const int rck = f3dg_launch_export_keys(s, n_views, P, W, H, L, const_cast<char*>(ws));
        if (rck != F3DG_OK) return rck;
    }
#define CP(dst, off, bytes) if (dst && (bytes)) F3DG_HIP_CHECK(hipMemcpyAsync(dst, ws + (off), (bytes), hipMemcpyDeviceToDevice, s))
    // (an inference call writes the record only for (view, Gaussian) pairs that are in a tile list: the other rows of `rec` hold whatever the
    // workspace held -- documented in f3dg.h; mask with radii > 0, and with the culled lists also with "is in a list")
    CP(rec, L.rec, VP * sizeof(F3dgRec));
    CP(means2D, L.means2D, VP * 8);
    CP(conic, L.conic, VP * 16);
    CP(tiles, L.tiles, VP * 4);
    CP(offsets, L.offsets, VP * 4);
    CP(clamped, L.clamped, VP);
    CP(keys_sorted, L.keys[0], C * 8);
    if (point_list && C)        // the Gaussian ids without the quadrant masks of the compositing kernel (F3DG_ID_BITS)
        F3DG_KLAUNCH(export_ids_kernel, dim3((unsigned)((C + 1023) / 1024 < 4096 ? (C + 1023) / 1024 : 4096)), dim3(256), 0, s,
                           reinterpret_cast<const unsigned*>(ws + L.vals[0]), C, point_list);
    CP(ranges, L.ranges, (size_t)n_views * T * 8);
    CP(final_T, L.final_T, (size_t)n_views * 4 * HW * 4);
    CP(n_contrib, L.n_contrib, (size_t)n_views * 2 * HW * 4);
    CP(depths, L.depths, VP * 4);
#undef CP
    return F3DG_OK;
}
