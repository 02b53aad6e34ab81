// f3dg_small.hip -- the small-call path of the forward: ONE kernel between the projection and the compositing kernel.
//
// The reference renders one view per rasterizer call (visualize.py:293-314, 387-416: 8 cycle views and the orbit, 65,536 to
// 589,824 Gaussians each). For such a call the general binning stage (f3dg_binning.hip: per-view depth sort, instance generation, tile
// pass; 26 dependent launches built for hundreds of views per call) is all launch latency: ~4.5 us per kernel whatever it does,
// ~150 us per call at 65,536 Gaussians. Here ONE kernel does the binning, one workgroup per (view, tile):
//   1. its sixteen waves find the Gaussians whose tile rectangle holds the tile, in two levels (round 5): the projection leaves the
//      union of the rectangles of every 64 consecutive Gaussians, a wave tests those unions first (one lane per 64 Gaussians) and only
//      the hit chunks have their rectangles read (8 B each, from L2), shared round-robin among the waves; a point cloud in no particular
//      id order hits most unions and is scanned flat, a sixteenth per wave. One ballot per 64 Gaussians notes which rectangles hold the
//      tile; the ballots are expanded into the hits' ids in id order (a prefix over the waves' counts gives every wave its place),
//      their depth keys gathered densely -- no atomics, no lists in global memory;
//   2. a stable LSD radix sort of the (depth bits, id) pairs in LDS (wave-ballot ranking, four 8-bit passes, passes whose digit is the
//      same for every entry are skipped) -- exactly the order of the reference's stable sort of (tile | depth) keys
//      (rasterizer_impl.cu:70-111, 358-363);
//   3. the list goes to the tile's slot together with its range.
// Four launches per call (header, projection, this, compositing) instead of 29.
//
// A slot holds F3DG_SMALL_CAP entries -- the limit is the tile's total, however its hits are spread over the waves (round 3 gave each
// wave a sixteenth of the slot, which pixel-ordered predicted Gaussians overflow: all hits of a tile come from the one or two waves
// that scan its image rows); a longer list sets the overflow flags and the caller re-runs the call on the general path (f3dg_read_status remembers the shape).
// Forwards with auxiliary planes take it too (round 5, option small_path_aux): f3dg_backward reads the header and walks the slots.
#include "f3dg_common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

#define SMALL_THREADS 1024
#define SMALL_WAVES (SMALL_THREADS / 64)
#define SMALL_MAXP (1u << 18)                                  // f3dg_small_shape: at most 2^18 Gaussians
#define SMALL_STEPS (SMALL_MAXP / 64u / SMALL_WAVES)          // 64-Gaussian steps of one wave's share: 256

struct SmallShared {
    // four arrays of F3DG_SMALL_CAP words: [2 c] depth bits and [2 c + 1] Gaussian id | quadrant mask << F3DG_ID_BITS of ping-pong
    // buffer c. While the Gaussians are scanned, buffer 0 (32 KB) holds every wave's hit masks instead: one 64-bit ballot per step.
    u32 buf[4][F3DG_SMALL_CAP];
    u32 hist[SMALL_WAVES][256];                // per-wave digit counts, then the waves' write offsets
    u32 wave_n[SMALL_WAVES];
    u32 wave_min[SMALL_WAVES], wave_max[SMALL_WAVES];
    u64 chit[SMALL_WAVES][SMALL_STEPS / 64];    // which chunks of every wave's share hold the tile
    u32 wsum[4];
    u32 skip;
};
static_assert(sizeof(u64) * SMALL_STEPS * SMALL_WAVES <= 2 * sizeof(u32) * F3DG_SMALL_CAP, "the hit masks must fit buffer 0");

// One workgroup of 16 waves per (view, tile): a single 256^2 view is 256 workgroups, one per CU, and the scan of the view's
// Gaussians is a latency chain per wave -- the sixteen waves of a CU each take a sixteenth of it.
__global__ void __launch_bounds__(SMALL_THREADS)
small_bin_kernel(u32 P, u32 T, u32 grid_x, F3dgHeader* __restrict__ hdr, const uint2* __restrict__ rects, const uint2* __restrict__ boxes,
                 const u32* __restrict__ sort_keys, u32* __restrict__ list, u32* __restrict__ cnt, uint2* __restrict__ ranges, int debug_stop)
{
    __shared__ SmallShared sh;
    if (debug_stop == 1) return;
    const u32 seg = blockIdx.x, view = seg / T, tile = seg % T;
    const u32 tx = tile % grid_x, ty = tile / grid_x;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u64 lt = (1ull << lane) - 1ull;
    const uint2* vrect = rects + (size_t)view * P;
    const u32* vkey = sort_keys + (size_t)view * P;

    // ---- 1. collect. Two levels: the projection left, for every CHUNK of 64 consecutive Gaussians, the union of their tile rectangles
    // (boxes); a wave first tests the boxes of its share of the Gaussians (wave w owns [w Pq, (w + 1) Pq): one lane per chunk), and only
    // the chunks whose box holds the tile have their 64 rectangles read -- by whichever wave comes next in a round-robin over ALL the
    // workgroup's hit chunks: predicted Gaussians are pixel-ordered (id = y * 256 + x), a wave's share is sixteen image rows = one tile
    // row, and all hits of a tile lie in the shares of one or two waves. A chunk's ballot -- which of its 64 Gaussians hit the tile --
    // goes to the OWNER's row of hit masks (buffer 0, one 64-bit word per chunk), so everything after this step is unchanged. A test only
    // looks at the packed tile rectangle (rmin <= t < rmax in both halves of a word with one packed 16-bit subtraction). The limit of a
    // tile is its TOTAL (F3DG_SMALL_CAP entries), however the hits are spread over the waves.
    const u32 Pq = ((P + 64u * SMALL_WAVES - 1u) / (64u * SMALL_WAVES)) * 64u;
    const u32 g0 = min(P, wave * Pq);
    const u32 CW = Pq / 64u;                     // chunks per wave: <= SMALL_STEPS
    const u32 nchunks = (P + 63u) / 64u;
    const uint2* vbox = boxes + (size_t)view * nchunks;
    u64* const wbal_all = reinterpret_cast<u64*>(&sh.buf[0][0]);
    u64* wbal = wbal_all + wave * SMALL_STEPS;
    typedef short pk16 __attribute__((ext_vector_type(2)));
    const u32 cxw = (tx + 1u) | ((tx + 1u) << 16), cyw = (ty + 1u) | ((ty + 1u) << 16);
    const pk16 cx = __builtin_bit_cast(pk16, cxw), cy = __builtin_bit_cast(pk16, cyw);
    // low half: rmin - (t + 1) < 0  <=>  rmin <= t;  high half: rmax - (t + 1) >= 0  <=>  t < rmax  (an empty rectangle fails the second)
    auto holds_tile = [&](const uint2 r) {
        const u32 dx = __builtin_bit_cast(u32, __builtin_bit_cast(pk16, r.x & 0x7FFF7FFFu) - cx);
        const u32 dy = __builtin_bit_cast(u32, __builtin_bit_cast(pk16, r.y & 0x7FFF7FFFu) - cy);
        return (((dx ^ 0x8000u) | (dy ^ 0x8000u)) & 0x80008000u) == 0u;
    };
    const u32 nsteps = CW;
    for (u32 s0 = 0; s0 < CW; s0 += 64u) {
        const u32 c = wave * CW + s0 + lane;
        const bool in = s0 + lane < CW && c < nchunks && holds_tile(vbox[min(c, nchunks - 1u)]);
        const u64 hb = __ballot(in);
        if (lane == 0) sh.chit[wave][s0 >> 6] = hb;
        if (s0 + lane < CW) wbal[s0 + lane] = 0ull;
    }
    __syncthreads();
    // how many chunks hold the tile (the same number in every wave): Gaussians in no particular id order -- a caller's own point
    // cloud -- put the tile into most boxes, and then every wave simply reads its own share, 32 chunks in flight
    u32 total_hits = lane < SMALL_WAVES * (SMALL_STEPS / 64u) && (lane % (SMALL_STEPS / 64u)) * 64u < CW ? (u32)__popcll(sh.chit[lane / (SMALL_STEPS / 64u)][lane % (SMALL_STEPS / 64u)]) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total_hits += (u32)__shfl_xor((int)total_hits, o, 64);
    if (total_hits * 4u > nchunks) {
        constexpr int UN = 32;
        const u32 g1 = min(P, g0 + Pq);
        u32 st = 0;
        for (u32 base = g0; base < g1; base += 64u * UN, st += UN) {
            uint2 r[UN];
#pragma unroll
            for (int u = 0; u < UN; u++)        // (unconditional loads from a clamped index: a guarded load is a branch + a full wait each)
                r[u] = vrect[min(base + 64u * u + lane, P - 1u)];
            u32 keep_lo = 0, keep_hi = 0;
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const u64 bal = __ballot(holds_tile(r[u]) && base + 64u * u + lane < g1);
                if (lane == (u32)u) { keep_lo = (u32)bal; keep_hi = (u32)(bal >> 32); }
            }
            if (lane < (u32)UN) wbal[st + lane] = ((u64)keep_hi << 32) | keep_lo;
        }
    } else {
        // hit chunk number k (in id order) is read by wave k % SMALL_WAVES, UNB chunks in flight
        constexpr int UNB = 16;
        u32 pend[UNB];
        int np = 0;
        u32 k = 0;
        auto flush = [&]() {
            uint2 r[UNB];
#pragma unroll
            for (int u = 0; u < UNB; u++)
                if (u < np) r[u] = vrect[min(pend[u] * 64u + lane, P - 1u)];
#pragma unroll
            for (int u = 0; u < UNB; u++)
                if (u < np) {
                    const bool in = holds_tile(r[u]) && pend[u] * 64u + lane < P;
                    const u64 bal = __ballot(in);
                    if (lane == 0) wbal_all[(pend[u] / CW) * SMALL_STEPS + pend[u] % CW] = bal;
                }
            np = 0;
        };
        const u32 words = (CW + 63u) / 64u;
        for (u32 ww = 0; ww < SMALL_WAVES; ww++)
            for (u32 wd = 0; wd < words; wd++) {
                u64 hb = sh.chit[ww][wd];
                while (hb != 0ull) {
                    const u32 bit = (u32)__builtin_ctzll(hb);
                    hb &= hb - 1ull;
                    if ((k++ & (SMALL_WAVES - 1u)) == wave) {
                        pend[np++] = ww * CW + wd * 64u + bit;
                        if (np == UNB) flush();
                    }
                }
            }
        if (np != 0) flush();
    }
    __syncthreads();
    u32 nw = 0;                                 // hits of this wave's share (wave-uniform)
    for (u32 s0 = 0; s0 < CW; s0 += 64u)
        nw += s0 + lane < CW ? (u32)__popcll(wbal[s0 + lane]) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nw += (u32)__shfl_xor((int)nw, o, 64);
    if (lane == 0) sh.wave_n[wave] = nw;
    __syncthreads();
    u32 n = 0, off = 0;
#pragma unroll
    for (u32 w = 0; w < SMALL_WAVES; w++) {
        const u32 c = sh.wave_n[w];
        if (w < wave) off += c;
        n += c;
    }
    const bool over = n > (u32)F3DG_SMALL_CAP;
    const u32 slot_base = seg * (u32)F3DG_SMALL_CAP;
    if (threadIdx.x == 0) {
        const u32 before = atomicAdd(&hdr->num_rendered, n);            // the call's instance count, as on the general path
        if (over) { hdr->overflow = 1u; hdr->small_overflow = 1u; }
        if (before + n > hdr->capacity || before + n < before) hdr->overflow = 1u;
        cnt[seg] = over ? 0u : n;
        ranges[seg] = over ? make_uint2(0u, 0u) : make_uint2(slot_base, slot_base + n);
    }
    if (over || n == 0u || debug_stop == 2)
        return;

    // the hits' ids, in id order, into buffer 1 at the wave's offset: lane l expands the ballots of steps l, l + 64, ...
    u32* k1 = sh.buf[2];
    u32* v1 = sh.buf[3];
    {
        u32 run = off;
        for (u32 s0 = 0; s0 < nsteps; s0 += 64u) {
            u64 m = s0 + lane < nsteps ? wbal[s0 + lane] : 0ull;
            u32 x = (u32)__popcll(m);
            const u32 mine = x;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 y = (u32)__shfl_up((int)x, o, 64);
                if (lane >= (u32)o) x += y;
            }
            u32 at = run + x - mine;
            const u32 gbase = g0 + 64u * (s0 + lane);
            while (m != 0ull) {
                v1[at++] = gbase + (u32)__builtin_ctzll(m);
                m &= m - 1ull;
            }
            run += (u32)__shfl((int)x, 63, 64);
        }
    }
    __syncthreads();
    if (debug_stop == 4) return;
    // depth key and quadrant mask of every hit (dense: at most four per thread), the key range of the list
    u32 kmin = 0xFFFFFFFFu, kmax = 0u;
    u32 mykey[F3DG_SMALL_CAP / SMALL_THREADS];
#pragma unroll
    for (u32 j = 0; j < F3DG_SMALL_CAP / SMALL_THREADS; j++) {
        const u32 i = threadIdx.x + j * SMALL_THREADS;
        mykey[j] = 0u;
        if (i < n) {
            const u32 g = v1[i];
            const uint2 r = vrect[g];
            const u32 key = vkey[g];
            const u32 rminx = r.x & F3DG_RECT_COORD, rmaxx = (r.x >> 16) & F3DG_RECT_COORD;
            const u32 rminy = r.y & F3DG_RECT_COORD, rmaxy = (r.y >> 16) & F3DG_RECT_COORD;
            // quadrant mask of the instance, as duplicate_sorted_kernel (f3dg_binning.hip) derives it
            u32 mx = 3u, my = 3u;
            if (tx == rminx && (r.x & F3DG_RECT_SKIP_LO)) mx &= ~1u;
            if (tx + 1u == rmaxx && (r.x & F3DG_RECT_SKIP_HI)) mx &= ~2u;
            if (ty == rminy && (r.y & F3DG_RECT_SKIP_LO)) my &= ~1u;
            if (ty + 1u == rmaxy && (r.y & F3DG_RECT_SKIP_HI)) my &= ~2u;
            const u32 qm = ((my & 1u) ? mx : 0u) | ((my & 2u) ? mx << 2 : 0u);
            v1[i] = g | (qm << F3DG_ID_BITS);
            mykey[j] = key;
            kmin = min(kmin, key);
            kmax = max(kmax, key);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, (u32)__shfl_xor((int)kmin, o, 64));
        kmax = max(kmax, (u32)__shfl_xor((int)kmax, o, 64));
    }
    if (lane == 0) { sh.wave_min[wave] = kmin; sh.wave_max[wave] = kmax; }
    __syncthreads();
#pragma unroll
    for (u32 w = 0; w < SMALL_WAVES; w++) {
        kmin = min(kmin, sh.wave_min[w]);
        kmax = max(kmax, sh.wave_max[w]);
    }
    // (keys relative to the list's smallest: the depths of a tile's list differ in ~22 bits, three 8-bit passes instead of four)
#pragma unroll
    for (u32 j = 0; j < F3DG_SMALL_CAP / SMALL_THREADS; j++) {
        const u32 i = threadIdx.x + j * SMALL_THREADS;
        if (i < n) k1[i] = mykey[j] - kmin;
    }
    __syncthreads();
    if (debug_stop == 5) return;
    const u32 span = kmax - kmin;
    const int key_bits = debug_stop == 3 ? 0 : span == 0u ? 0 : 32 - __builtin_clz(span);

    // ---- 2. stable LSD radix sort by the depth bits; wave w owns entries [w q, (w + 1) q) of the current buffer
    const u32 q = ((n + 64u * SMALL_WAVES - 1u) / (64u * SMALL_WAVES)) * 64u;
    const u32 e0 = min(n, wave * q), e1 = min(n, e0 + q);
    u32 cur = 1;
    for (int shift = 0; shift < key_bits; shift += 8) {
        sh.hist[wave][lane] = 0; sh.hist[wave][lane + 64] = 0; sh.hist[wave][lane + 128] = 0; sh.hist[wave][lane + 192] = 0;
        if (threadIdx.x == 0) sh.skip = 0;
        __syncthreads();
        for (u32 i = e0 + lane; i < e1; i += 64u)
            atomicAdd(&sh.hist[wave][(sh.buf[2u * cur][i] >> shift) & 255u], 1u);
        __syncthreads();
        // thread d < 256: total of digit d, exclusive scan over the digits, per-wave offsets
        u32 tot = 0, x = 0;
        if (threadIdx.x < 256u) {
#pragma unroll
            for (u32 w = 0; w < SMALL_WAVES; w++) tot += sh.hist[w][threadIdx.x];
            if (tot == n) sh.skip = 1;                      // every entry has this digit: the pass is the identity
            x = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 y = __shfl_up(x, o, 64);
                if (lane >= (u32)o) x += y;
            }
            if (lane == 63) sh.wsum[wave] = x;
        }
        __syncthreads();
        if (threadIdx.x < 256u) {
            u32 run = x - tot;
            for (u32 w = 0; w < wave; w++) run += sh.wsum[w];
#pragma unroll
            for (u32 w = 0; w < SMALL_WAVES; w++) {
                const u32 h = sh.hist[w][threadIdx.x];
                sh.hist[w][threadIdx.x] = run;
                run += h;
            }
        }
        __syncthreads();
        if (sh.skip) {
            __syncthreads();
            continue;
        }
        for (u32 i0 = e0; i0 < e1; i0 += 64u) {
            const u32 i = i0 + lane;
            const bool valid = i < e1;
            const u32 key = valid ? sh.buf[2u * cur][i] : 0u, val = valid ? sh.buf[2u * cur + 1u][i] : 0u;
            const u32 d = (key >> shift) & 255u;
            u64 m = __ballot(valid);                        // lanes of this step with my digit
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const u64 bal = __ballot((d >> b) & 1u);
                m &= ((d >> b) & 1u) ? bal : ~bal;
            }
            if (valid) {
                const u32 pos = sh.hist[wave][d] + (u32)__popcll(m & lt);
                sh.buf[2u * (cur ^ 1u)][pos] = key;
                sh.buf[2u * (cur ^ 1u) + 1u][pos] = val;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (valid && (m & lt) == 0ull)                  // the first lane of every digit group moves the wave's offset on
                sh.hist[wave][d] += (u32)__popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        cur ^= 1u;
        __syncthreads();
    }

    // ---- 3. the tile's list
    for (u32 i = threadIdx.x; i < n; i += SMALL_THREADS)
        list[(size_t)slot_base + i] = sh.buf[2u * cur + 1u][i];
}

// debug export: the lists in (view, tile) order without gaps, as the general path lays them out, and the matching ranges
__global__ void __launch_bounds__(F3DG_BLOCK)
small_export_kernel(const u32* __restrict__ cnt, const u32* __restrict__ list, u32* __restrict__ point_list, u32* __restrict__ ranges)
{
    __shared__ u32 part[F3DG_BLOCK];
    const u32 seg = blockIdx.x;
    u32 acc = 0;
    for (u32 i = threadIdx.x; i < seg; i += F3DG_BLOCK) acc += cnt[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (u32 s = F3DG_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    const u32 start = part[0], n = cnt[seg];
    if (threadIdx.x == 0 && ranges) { ranges[2u * seg] = n ? start : 0u; ranges[2u * seg + 1u] = n ? start + n : 0u; }
    if (point_list)
        for (u32 i = threadIdx.x; i < n; i += F3DG_BLOCK)
            point_list[start + i] = list[(size_t)seg * F3DG_SMALL_CAP + i] & F3DG_ID_MASK;
}

} // namespace

int f3dg_launch_small_bin(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws)
{
    const u32 grid_x = (u32)((W + F3DG_TILE - 1) / F3DG_TILE);
    const u32 T = grid_x * (u32)((H + F3DG_TILE - 1) / F3DG_TILE);
    F3DG_KLAUNCH(small_bin_kernel, dim3((u32)V * T), dim3(SMALL_THREADS), 0, s, (u32)P, T, grid_x, reinterpret_cast<F3dgHeader*>(ws + L.header),
                 reinterpret_cast<const uint2*>(ws + L.rects), reinterpret_cast<const uint2*>(ws + L.small_boxes), reinterpret_cast<const u32*>(ws + L.gsort),
                 reinterpret_cast<u32*>(ws + L.small_list), reinterpret_cast<u32*>(ws + L.small_cnt), reinterpret_cast<uint2*>(ws + L.ranges),
                 g_f3dg_small_debug);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

int f3dg_launch_small_export(hipStream_t s, int V, int W, int H, const F3dgLayout& L, const char* ws, unsigned* point_list, unsigned* ranges)
{
    const u32 T = (u32)(((W + F3DG_TILE - 1) / F3DG_TILE) * ((H + F3DG_TILE - 1) / F3DG_TILE));
    F3DG_KLAUNCH(small_export_kernel, dim3((u32)V * T), dim3(F3DG_BLOCK), 0, s, reinterpret_cast<const u32*>(ws + L.small_cnt),
                 reinterpret_cast<const u32*>(ws + L.small_list), point_list, ranges);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
