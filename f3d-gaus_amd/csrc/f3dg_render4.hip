// f3dg_render4.hip -- compositing forward with RANK-PACKED trips (option render_kernel = 4).
//
// render3s_fwd_kernel (f3dg_render.hip) gives every pixel of an 8x8 quadrant a lane; in a phase-2 trip every pixel that still has a
// passing entry in the window pops its next one and runs the whole loop body of renderCUDA (reference RAST/cuda_rasterizer/
// forward.cu:493-583) on it. A trip costs the wave ~95 issue slots whether 64 pixels take part or 3 -- and on pixel-aligned splats over
// a real depth map (the reference's production data, visualize.py:293-340) 56 % of the trips have fewer than 16 takers: the entries of
// a 32-entry half-window lie on one iso-depth contour, the pixels under it get ~14 of them, the average pixel 3 (lane utilisation 0.24;
// 0.63 on the synthetic C2 recipe).
//
// Two thirds of a trip do not depend on the pixel's running state (f3dg_blend.h: f3dg_pair_eval -- quadric, error-free quotient, exp,
// NDC depth, unit normal); only the last third (f3dg_pair_apply: the transmittance recurrence and the accumulators) is serial per
// pixel. This kernel keeps render3s's scan / staging / phase 1 / sliding half-windows and splits phase 2 of a slide in two regimes:
//   * FUSED trips, as before, while more than `pack_th` pixels take part;
//   * then PACKED batches. A batch takes the next R pending entries of every pixel (R ranks: as many as fit 64 pairs, at most what the
//     pixels that hold the window back still need), writes the (pixel, slot) pairs to a 64-entry LDS queue in pixel-major order
//     (offset of pixel p = sum over ranks of mbcnt of the rank's ballot), evaluates the stateless part ONE PAIR PER LANE (the ray of
//     the pair's pixel comes by ds_bpermute from the lane that owns it, the record from the staged window), parks the six numbers of
//     every pair in LDS, and then runs R short blend trips -- a divergent loop -- in which the owning lanes read their pairs' numbers
//     and apply the recurrence (branch-free in fast arithmetic: a rejected or saturating pair runs with weight 0).
// Per pixel the sequence of blended entries and every operation on them is unchanged, so the images -- and, in the SAVE_AUX variant, the
// auxiliary planes the backward reads -- are bit-identical to render3s in either arithmetic mode (tests/test_raster_forward_gpu.py).
// LDS per wave: 4 KB of records + 512 B of id ring + 1.5 KB of parking area (the queue aliases its first 128 bytes): 6,144 B (5,120 B
// without normals and distortion, 6,912 B with SAVE_AUX). Measured (profiles/r05_final/compositing_packed.md): in the reference's
// arithmetic, whose stateless part is 2.5 x as long, -38 % on the real merged set and -7 % at C2; in fast arithmetic equal to render3s
// (the hand-over and its scalar bookkeeping cost what the packing saves) -- hence the default of f3dg_launch_render: render_pack -1.
#include "f3dg_blend.h"
#include "f3dg_ellipse.h"
#include "f3dg_producer.h"

#include <stdio.h>
#include <string.h>

extern thread_local const char* g_f3dg_last_render_kernel;
int g_f3dg_render_pack_th = 32;       // option render_pack_th: trips with at most this many participating pixels are packed (0: never)

// work counters (option render_count = 1; f3dg_debug_render_counts rows 16..): [0] staged entries, [1] scanned, [2] fused trips,
// [3] slides, [4] lane-trips of fused trips, [5] waves, [6] packed batches (= dense trips), [7] blend trips of packed batches,
// [8] pairs evaluated in dense trips, [9] pairs blended or rejected in blend trips
__device__ unsigned long long g_f3dg_counts4[64][16];

namespace {

#ifndef F3DG_R4_PARK
#define F3DG_R4_PARK 1              // 1: a batch's results go to its blend trips through LDS (parked); 0: by ds_bpermute from the dense lanes' registers
#endif
#ifndef F3DG_R4_FLAT
#define F3DG_R4_FLAT 1              // 1: the blend trips of fast arithmetic run the branch-free recurrence (f3dg_pair_apply_flat)
#endif
#ifndef F3DG_R4_FLAT_FUSED
#define F3DG_R4_FLAT_FUSED 0        // 1: the fused trips of fast arithmetic run the branch-free recurrence as well (measured: no gain)
#endif
#define F3DG_R4_WIN 64
#define F3DG_R4_RING 128
#define F3DG_R4_MAXR 10             // ranks per packed batch (two 32-bit registers of 6-bit slots)
#define F3DG_R4_FLAG 0x80000000u

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// keeps the pixel state in ONE set of registers across the packed batches (without it the register allocator parks the twelve
// accumulators in a second set around the dense trip: 24 v_mov per batch)
#define F3DG_R4_PIN(st) asm volatile("" : "+v"((st).Tr), "+v"((st).C0), "+v"((st).C1), "+v"((st).C2), "+v"((st).C3), "+v"((st).C4), "+v"((st).C5), \
                                          "+v"((st).C6), "+v"((st).C7), "+v"((st).dist1), "+v"((st).dist2), "+v"((st).distortion))

__device__ __forceinline__ float pull(int addr, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}

// SAVE_AUX (a forward that f3dg_backward follows): also final_T [V][4][HW] and n_contrib [V][2][HW]; last_contributor / max_contributor are
// 1-based positions in the tile's list, kept per staged slot (sP) and translated when a half of the window retires, as in render3s.
#ifndef F3DG_R4_OCC
#define F3DG_R4_OCC 8               // waves per SIMD the register allocation aims at (the full variants' 6 KB of LDS allow 6.5)
#endif
template <bool FAST, bool NORMAL, bool DIST, bool COUNT, bool SAVE_AUX = false>
__global__ void __launch_bounds__(64, F3DG_R4_OCC)
render4_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                   const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                   const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                   const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                   float* __restrict__ out_color, int pack_th, float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    unsigned view, unit;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, 4u * (unsigned)T, view, unit);
    const unsigned tile = unit >> 2, quad = unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x;
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;
    const unsigned pix_x = qx0 + (lane & 7u), pix_y = qy0 + (lane >> 3);
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const unsigned n = range.y - range.x;

    __shared__ float4 sR[4][F3DG_R4_WIN];          // records, [16-byte chunk][slot]; slots 0..31 and 32..63 are the two halves of the window
    __shared__ unsigned sQ[F3DG_R4_RING];          // ids of kept entries not staged yet, ring
    __shared__ unsigned sQpos[SAVE_AUX ? F3DG_R4_RING : 1];     // ... and their positions in the tile's list
    __shared__ unsigned sP[SAVE_AUX ? F3DG_R4_WIN : 1];         // list position of every staged slot (the reference's `contributor`)
#if F3DG_R4_PARK
    // the parking area of a packed batch: what the dense trip hands to the blend trips, [position in the queue]. The queue itself
    // (sK: 64 x u16, (owning lane << 6) | physical slot) is read by the dense trip before it parks its results and aliases the first
    // 128 bytes. (Aliasing the id ring as well -- its pending ids parked in a register between slides, 5,632 instead of 6,144 bytes,
    // 29 instead of 26 waves per CU -- measured 1.5-2.5 % SLOWER on the real merged set: occupancy is not what limits this kernel.)
    __shared__ float4 sPark[(NORMAL || DIST) ? 64 : 32];       // full: (alpha, t, m, nn0)        lean: 64 x (alpha, t)
    __shared__ float2 sPark2[(NORMAL || DIST) ? 64 : 1];       // full: (nn1, nn2)
    unsigned short* sK = reinterpret_cast<unsigned short*>(sPark);
#else
    __shared__ unsigned short sK[64];              // pair queue of a packed batch: (owning lane << 6) | physical slot
#endif

    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned hl = lane & 31u;               // entry of a half this lane tests in phase 1 ...
    const unsigned row4 = (lane >> 5) * 4u;       // ... against the pixels of rows row4 .. row4 + 3

    bool done = !inside;
    F3dgPixel st;
    f3dg_pixel_init(st);

    auto translate = [&](unsigned half_or_all) {      // slots -> 1-based list positions for the slots of one physical half (2: both)
        if (SAVE_AUX) {
            const unsigned a = st.last_contributor - F3DG_R4_FLAG, b = st.max_contributor - F3DG_R4_FLAG;
            if (a < (unsigned)F3DG_R4_WIN && (half_or_all == 2u || (a >> 5) == half_or_all)) st.last_contributor = sP[a] + 1u;
            if (b < (unsigned)F3DG_R4_WIN && (half_or_all == 2u || (b >> 5) == half_or_all)) st.max_contributor = sP[b] + 1u;
        }
    };
    unsigned n_staged = 0, n_fused = 0, n_slides = 0, n_lane_fused = 0, n_batches = 0, n_blend_trips = 0, n_dense_pairs = 0, n_blend_pairs = 0;
    unsigned cursor = 0, qhead = 0, qpend = 0;    // wave-uniform: scan position, ring index of the first pending entry, pending entries
    unsigned flip = 0;                            // physical half (slots 32 flip ..) that holds the OLDER half of the window
    unsigned long long pass = 0ull;               // per pixel: bits 0..31 older half, 32..63 newer half, in list order
    unsigned idn = lane < n ? point_list[range.x + lane] : 0u;
    if (__ballot(!done) != 0ull)
    for (;;) {
        // ---- scan: keep the entries whose box reaches this quadrant until 32 are pending
        while (qpend < 32u && cursor < n) {
            const unsigned idm = idn, pos = cursor + lane;
            cursor += 64u;
            idn = cursor + lane < n ? point_list[range.x + cursor + lane] : 0u;
            const bool keep = pos < n && (idm & qbit) != 0u;
            const unsigned long long kb = __ballot(keep);
            if (keep) {
                const unsigned slot = (qhead + qpend + __builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u))) & (F3DG_R4_RING - 1);
                sQ[slot] = idm & F3DG_ID_MASK;
                if (SAVE_AUX) sQpos[slot] = pos;
            }
            qpend += (unsigned)__popcll(kb);
        }
        const unsigned m = qpend < 32u ? qpend : 32u;
        // every live pixel has finished the older half (bits 0..31 of `pass` are clear): retire it
        translate(flip);
        if (m == 0u && __ballot(pass != 0ull) == 0ull)
            break;                                // nothing left to stage, nothing left in the newer half
        wave_lds_fence();

        // ---- stage m entries into the retired half; lanes e and e + 32 both take entry e
        const unsigned base = flip * 32u;
        float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float ec = 0.0f;
        if (hl < m) {
            const unsigned id = sQ[(qhead + hl) & (F3DG_R4_RING - 1)];
            if (lane < 32u) {
                const float4* src = reinterpret_cast<const float4*>(vrec + id);
#pragma unroll
                for (int c = 0; c < 4; c++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)&sR[c][base], 16, 0, 0);
                if (SAVE_AUX) sP[base + lane] = sQpos[(qhead + hl) & (F3DG_R4_RING - 1)];
            }
            e4 = vcull[id];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        if (hl < m) ec = sR[3][base + hl].w;
        qhead += m;
        qpend -= m;
        if (COUNT) { n_staged += m; n_slides++; }

        // ---- phase 1: the 32 new entries against the quadrant's 64 pixels
        int fresh = 0;
        if (m != 0u) {
            const float u0 = hl < m ? (float)qx0 - e4.x : __builtin_nanf("");     // NaN: every comparison below is false
            const float v0 = (float)(qy0 + row4) - e4.y;
            float dxx[8], adx[8], dyy[4], cdy[4];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                dxx[q] = u0 + (float)q;
                adx[q] = e4.z * dxx[q];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                dyy[q] = v0 + (float)q;
                cdy[q] = ec * dyy[q] * dyy[q];
            }
            half_ballots<0>(fresh, fmaf(dxx[0], fmaf(e4.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4.w);
        }
        // ---- slide: the newer half becomes the older one, the fresh bits the newer one
        pass = (pass >> 32) | (done ? 0ull : ((unsigned long long)(unsigned)fresh << 32));
        flip ^= 1u;
        const unsigned xr = flip << 5;            // logical slot j (0..31 older, 32..63 newer) lives in physical slot j ^ xr

        // ---- phase 2a: fused trips while many pixels take part (a divergent loop: a pixel leaves it when its mask is empty; the
        // ballots are taken over the pixels still inside)
#ifdef F3DG_R4_NOPACK       // experiment: the fused loop without its population test
        while (pass != 0ull && __ballot((unsigned)pass != 0u) != 0ull) {
#else
        while (pass != 0ull && __ballot((unsigned)pass != 0u) != 0ull && (int)__popcll(__ballot(true)) > pack_th) {
#endif
            const unsigned j = (unsigned)__builtin_ctzll(pass) ^ xr;
            pass &= pass - 1;
            if (COUNT) n_lane_fused++;
            const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j], q3 = sR[3][j];
            const F3dgPair pr = f3dg_pair_eval<FAST, NORMAL, DIST, F3DG_R4_FLAT_FUSED != 0>(ray_x, ray_y, q0, q1, q2);
            // keeps the loads 16 bytes wide (ds_read_b96 takes twice the LDS cycles); placed behind the evaluation so that the
            // arithmetic on the first chunks starts while the last ones are still on their way
            asm volatile("" :: "v"(q2.w), "v"(q3.w), "v"(pr.alpha));
#if F3DG_R4_FLAT_FUSED
            if (FAST)
                done = f3dg_pair_apply_flat<NORMAL, DIST>(st, F3DG_R4_FLAG | j, pr, q3.x, q3.y, q3.z);
            else
#endif
            if (pr.alpha != 0.0f)
                done = f3dg_pair_apply<FAST, NORMAL, DIST>(st, F3DG_R4_FLAG | j, pr, q3.x, q3.y, q3.z);
            if (done) pass = 0ull;
            if (COUNT && __builtin_ctzll(__ballot(true)) == (int)lane) n_fused++;
        }

        // ---- phase 2b: packed batches until every live pixel has finished the older half
        // (bottom-tested on purpose: with the exit test at the top the structurizer routes the exit through the block behind the body,
        // the state at the loop header stays live across the body in a second set of registers, and every batch pays 24 v_mov)
        if (__ballot((unsigned)pass != 0u) != 0ull)
        do {
            F3DG_R4_PIN(st);
            const unsigned cnt = (unsigned)__popcll(pass);        // pending entries of this pixel in the window (0 for a finished pixel)
            const unsigned need = (unsigned)__popc((unsigned)pass);   // ... of them in the older half
            // rank 0 always fits (at most 64 pixels have a pending entry) and is always needed (some pixel holds the window back)
            unsigned R = 0, total = 0, off = 0;
            do {
                const unsigned long long b = __ballot(cnt > R);
                off = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, off));
                total += (unsigned)__popcll(b);
                R++;
            } while (__ballot(need > R) != 0ull && total + (unsigned)__popcll(__ballot(cnt > R)) <= 64u && R < F3DG_R4_MAXR);
            const unsigned c = cnt < R ? cnt : R;                 // this pixel's pairs of the batch, at queue positions off .. off + c - 1
            unsigned long long slots = 0ull;                      // their physical slots, 6 bits each
            {
                unsigned i = 0;
#pragma nounroll
                do {
                    if (i < c) {
                        const unsigned j = (unsigned)__builtin_ctzll(pass) ^ xr;
                        pass &= pass - 1;
                        sK[off + i] = (unsigned short)((lane << 6) | j);
                        slots |= (unsigned long long)j << (6u * i);
                    }
                } while (++i < R);
            }
            wave_lds_fence();

            // dense trip: lane q evaluates pair q of the queue (lanes beyond `total` evaluate a stale pair nobody pulls)
            F3dgPair pr;
            {
                const unsigned k = (unsigned)sK[lane];
                const int owner = (int)((k >> 6) << 2);
                const float rx = pull(owner, ray_x), ry = pull(owner, ray_y);
                const unsigned j = k & 63u;
                const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j];
                pr = f3dg_pair_eval<FAST, NORMAL, DIST, F3DG_R4_FLAT != 0>(rx, ry, q0, q1, q2);
                asm volatile("" :: "v"(q2.w), "v"(pr.alpha));
            }
            if (COUNT) { n_batches++; n_blend_trips += R; n_dense_pairs += total; }

#if F3DG_R4_PARK
            // park the results (every lane has read its queue entry: the queue's bytes may go), then the blend trips: a divergent loop,
            // every owning lane walks ITS pairs of the batch through the recurrence, the wave runs as long as the longest of them
            wave_lds_fence();
            if (NORMAL || DIST) {
                sPark[lane] = make_float4(pr.alpha, pr.t, pr.m, pr.nn0);
                sPark2[lane] = make_float2(pr.nn1, pr.nn2);
            } else {
                reinterpret_cast<float2*>(sPark)[lane] = make_float2(pr.alpha, pr.t);
            }
            wave_lds_fence();
            {
                unsigned i = 0;
                while (i < c && !done) {
                    const unsigned j = (unsigned)slots & 63u;
                    slots >>= 6;
                    const float4 q3 = sR[3][j];
                    F3dgPair mine;
                    if (NORMAL || DIST) {
                        const float4 a = sPark[off + i];
                        const float2 b = sPark2[off + i];
                        mine.alpha = a.x; mine.t = a.y; mine.m = a.z; mine.nn0 = a.w; mine.nn1 = b.x; mine.nn2 = b.y;
                    } else {
                        const float2 a = reinterpret_cast<const float2*>(sPark)[off + i];
                        mine.alpha = a.x; mine.t = a.y; mine.m = 0.0f; mine.nn0 = mine.nn1 = mine.nn2 = 0.0f;
                    }
                    if (COUNT) n_blend_pairs++;
#if F3DG_R4_FLAT
                    if (FAST)
                        done = f3dg_pair_apply_flat<NORMAL, DIST>(st, F3DG_R4_FLAG | j, mine, q3.x, q3.y, q3.z);
                    else
#endif
                    if (mine.alpha != 0.0f)
                        done = f3dg_pair_apply<FAST, NORMAL, DIST>(st, F3DG_R4_FLAG | j, mine, q3.x, q3.y, q3.z);
                    asm volatile("" :: "v"(q3.w));
                    i++;
                }
            }
            wave_lds_fence();          // the next batch's queue overwrites the parking area
#else
            // R blend trips: every owning lane pulls its pair's numbers and applies the recurrence
            F3DG_R4_PIN(st);
            int src = (int)(off << 2);
            unsigned i = 0;
#pragma nounroll
            do {
                // (the colour of this trip's pair: requested before the pulls, needed last)
                const unsigned j = (unsigned)slots & 63u;
                const float4 q3 = sR[3][j];
                F3dgPair mine;
                mine.alpha = pull(src, pr.alpha);
                mine.t = pull(src, pr.t);
                mine.m = DIST ? pull(src, pr.m) : 0.0f;
                mine.nn0 = NORMAL ? pull(src, pr.nn0) : 0.0f;
                mine.nn1 = NORMAL ? pull(src, pr.nn1) : 0.0f;
                mine.nn2 = NORMAL ? pull(src, pr.nn2) : 0.0f;
                if (i < c && !done) {
                    if (COUNT) n_blend_pairs++;
                    if (mine.alpha != 0.0f) {
                        asm volatile("" :: "v"(q3.w));
                        done = f3dg_pair_apply<FAST, NORMAL, DIST>(st, F3DG_R4_FLAG | j, mine, q3.x, q3.y, q3.z);
                    }
                }
                src += 4;
                slots >>= 6;
            } while (++i < R);
#endif
            if (done) pass = 0ull;
        } while (__ballot((unsigned)pass != 0u) != 0ull);

        if (__ballot(!done) == 0ull)
            break;
    }
    translate(2u);
    if (COUNT) {
        unsigned a = n_lane_fused, b = n_blend_pairs;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += (unsigned)__shfl_xor((int)a, o, 64);
            b += (unsigned)__shfl_xor((int)b, o, 64);
        }
        unsigned f = n_fused;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) f += (unsigned)__shfl_xor((int)f, o, 64);
        if (lane == 0) {
            unsigned long long* c = g_f3dg_counts4[blockIdx.x & 63u];
            atomicAdd(&c[0], (unsigned long long)n_staged);
            atomicAdd(&c[1], (unsigned long long)(cursor < n ? cursor : n));
            atomicAdd(&c[2], (unsigned long long)f);
            atomicAdd(&c[3], (unsigned long long)n_slides);
            atomicAdd(&c[4], (unsigned long long)a);
            atomicAdd(&c[5], 1ull);
            atomicAdd(&c[6], (unsigned long long)n_batches);
            atomicAdd(&c[7], (unsigned long long)n_blend_trips);
            atomicAdd(&c[8], (unsigned long long)n_dense_pairs);
            atomicAdd(&c[9], (unsigned long long)b);
        }
    }

    // (the pixel's coordinates are formed again from an opaque copy of the lane id: kept live across the loop they cost the two registers
    // that make the difference between 64 VGPRs and spilling)
    unsigned lane_e = threadIdx.x;
    asm volatile("" : "+v"(lane_e));
    const unsigned out_x = qx0 + (lane_e & 7u), out_y = qy0 + (lane_e >> 3);
    if (out_x < (unsigned)W && out_y < (unsigned)H) {
        const size_t HW = (size_t)H * W;
        const size_t pix_id = (size_t)W * out_y + out_x;
        const float* bg = background + (bg_per_view ? 3 * view : 0);
        const float Tr = st.Tr;
        const float distortion = (float)(st.distortion / ((1 - Tr) * (1 - Tr) + 1e-7));
        if (SAVE_AUX) {
            float* fT = final_T + (size_t)view * 4 * HW;
            fT[pix_id] = Tr;
            fT[pix_id + HW] = st.dist1;
            fT[pix_id + 2 * HW] = st.dist2;
            fT[pix_id + 3 * HW] = st.distortion;
            unsigned* nc = n_contrib + (size_t)view * 2 * HW;
            nc[pix_id] = st.last_contributor;
            nc[pix_id + HW] = st.max_contributor;
        }
        float* out = out_color + (size_t)view * F3DG_OUT_CHANNELS * HW;
        out[0 * HW + pix_id] = st.C0 + Tr * bg[0];
        out[1 * HW + pix_id] = st.C1 + Tr * bg[1];
        out[2 * HW + pix_id] = st.C2 + Tr * bg[2];
        if (NORMAL) {
            out[3 * HW + pix_id] = st.C3;
            out[4 * HW + pix_id] = st.C4;
            out[5 * HW + pix_id] = st.C5;
        }
        out[6 * HW + pix_id] = st.C6;
        out[7 * HW + pix_id] = st.C7;
        if (DIST) out[8 * HW + pix_id] = distortion;
    }
}

thread_local char g_kernel_name4[160] = "";


#define F3DG_R3U_RING 256
// ---- SMALL launches, two waves per quadrant: a PRODUCER wave prepares window k + 1 while the CONSUMER wave composites window k ------
// What is left of a lone wave's chain once phase 2 is shortened is everything else: with no entry passing the ellipse test at all
// (option debug_skip_all) the one-view kernel still takes 36 of its 66 us -- list chunks, the id-dependent record gathers, the 64
// ellipse ballots of phase 1, each a latency nobody fills. That part does not depend on any pixel's state, so here it runs on its own
// wave: workgroup = 2 waves on 2 SIMDs of a CU; wave 1 scans the list, requests the records of the next window into the other half
// of the double buffer (global_load_lds), waits for them, runs phase 1 and leaves the 64 pass masks in LDS; wave 0 owns the pixels and
// only runs phase 2 -- U entries per trip: the next U passing entries of every pixel are popped together, their stateless parts
// (f3dg_pair_eval) evaluated as U independent instruction streams the scheduler interleaves (the record reads of all U leave LDS together;
// a lone wave issues a dependent instruction every ~5 clocks, independent ones every 2.5), the recurrence applied in list order. One
// s_barrier per window. The time of a window is the longer of the two parts
// instead of their sum. Per pixel nothing changes: bit-identical outputs and auxiliary planes.
template <bool SAVE_AUX, bool FAST, int U>
__global__ void __launch_bounds__(128, 1)
render3p_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                    const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                    const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                    const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                    float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    unsigned view, unit;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, 4u * (unsigned)T, view, unit);
    const unsigned tile = unit >> 2, quad = unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u;
    const bool producer = threadIdx.x >= 64u;             // (wave-uniform)
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;

    __shared__ float4 sR[3][4][F3DG_PROD_WIN];            // three windows of records, [window % 3][16-byte chunk][entry]
    __shared__ uint2 sQ[F3DG_PROD_RING];                  // (list position, Gaussian id) of the kept entries (the producer's ring)
    __shared__ unsigned long long sPass[2][64];           // per window (k & 1): the pass mask of every pixel
    __shared__ uint2 sMH[2];                              // per window: (its number of entries (0: the list has ended), its first ring slot)
    __shared__ unsigned sStop[2];                         // [b]: set by the consumer when every pixel was done after the window in buffer b

    if (threadIdx.x < 2u) sStop[threadIdx.x] = 0u;

    if (producer) {
        // ================================================ wave 1: scan, gather, phase 1 (f3dg_producer.h) ================================
        f3dg_window_producer(lane, view, tile, quad, qx0, qy0, P, T, hdr, ranges, point_list, rec, cull, sR, sQ, sPass, sMH, sStop, 1u);
        return;
    }

    // ==================================================== wave 0: the pixels, phase 2 ====================================================
    const unsigned pix_x = qx0 + (lane & 7u), pix_y = qy0 + (lane >> 3);
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);
    bool done = !inside;
    F3dgPixel st;
    f3dg_pixel_init(st);
    {
        unsigned buf = 0, rb = 0;                          // window k: pass masks in [k & 1], records in [k % 3]
        for (;;) {
            __syncthreads();                                  // window `buf` is ready
            const unsigned m = sMH[buf].x;
            if (m == 0u || sStop[buf ^ 1u] != 0u)
                break;
            unsigned long long pass = done ? 0ull : sPass[buf][lane];
            while (pass != 0 && !done) {
                int j[U];
                bool have[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    have[u] = pass != 0ull;
                    j[u] = have[u] ? __builtin_ctzll(pass) : j[0];
                    pass &= pass - 1ull;
                }
                F3dgPair pr[U];
                float cr[U], cg[U], cb[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float4 q0 = sR[rb][0][j[u]], q1 = sR[rb][1][j[u]], q2 = sR[rb][2][j[u]], q3 = sR[rb][3][j[u]];
                    pr[u] = f3dg_pair_eval<FAST, true, true, FAST>(ray_x, ray_y, q0, q1, q2);
                    if (!have[u]) pr[u].alpha = 0.0f;
                    cr[u] = q3.x; cg[u] = q3.y; cb[u] = q3.z;
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (FAST) {
                        F3dgPair p = pr[u];
                        p.alpha = done ? 0.0f : p.alpha;
                        done |= f3dg_pair_apply_flat<true, true>(st, F3DG_R4_FLAG | (unsigned)j[u], p, cr[u], cg[u], cb[u]);
                    } else if (!done && pr[u].alpha != 0.0f) {
                        done = f3dg_pair_apply<false, true, true>(st, F3DG_R4_FLAG | (unsigned)j[u], pr[u], cr[u], cg[u], cb[u]);
                    }
                }
            }
            if (SAVE_AUX) {             // slots -> 1-based list positions (the reference's `contributor`)
                const unsigned head = sMH[buf].y;
                if (st.last_contributor - F3DG_R4_FLAG < (unsigned)F3DG_R4_WIN)
                    st.last_contributor = sQ[(head + (st.last_contributor - F3DG_R4_FLAG)) & (F3DG_PROD_RING - 1)].x + 1u;
                if (st.max_contributor - F3DG_R4_FLAG < (unsigned)F3DG_R4_WIN)
                    st.max_contributor = sQ[(head + (st.max_contributor - F3DG_R4_FLAG)) & (F3DG_PROD_RING - 1)].x + 1u;
            }
            if (__ballot(!done) == 0ull && lane == 0) sStop[buf] = 1u;  // (read by both waves after the next barrier)
            buf ^= 1u;
            rb = rb == 2u ? 0u : rb + 1u;
        }
    }

    if (inside) {
        const float* bg = background + (bg_per_view ? 3 * view : 0);
        const float Tr = st.Tr;
        const float distortion_before_normalized = st.distortion;
        const float distortion = (float)(st.distortion / ((1 - Tr) * (1 - Tr) + 1e-7));

        if (SAVE_AUX) {
            float* fT = final_T + (size_t)view * 4 * HW;
            fT[pix_id] = Tr;
            fT[pix_id + HW] = st.dist1;
            fT[pix_id + 2 * HW] = st.dist2;
            fT[pix_id + 3 * HW] = distortion_before_normalized;
            unsigned* nc = n_contrib + (size_t)view * 2 * HW;
            nc[pix_id] = st.last_contributor;
            nc[pix_id + HW] = st.max_contributor;
        }
        float* out = out_color + (size_t)view * F3DG_OUT_CHANNELS * HW;
        out[0 * HW + pix_id] = st.C0 + Tr * bg[0];
        out[1 * HW + pix_id] = st.C1 + Tr * bg[1];
        out[2 * HW + pix_id] = st.C2 + Tr * bg[2];
        out[3 * HW + pix_id] = st.C3;
        out[4 * HW + pix_id] = st.C4;
        out[5 * HW + pix_id] = st.C5;
        out[6 * HW + pix_id] = st.C6;
        out[7 * HW + pix_id] = st.C7;
        out[8 * HW + pix_id] = distortion;
    }
}

// ---- SMALL launches, a PIPELINE of waves per quadrant (option render_split = 2 / 3) -------------------------------------------------
// render3p above takes the window preparation off the pixels' wave; what remains on it is phase 2 -- ~48 of 58 us at 65,536
// Gaussians -- and two thirds of a phase-2 trip are the stateless part of the pair (f3dg_pair_eval). Here that part moves to E more
// waves: the workgroup of a quadrant is
//     wave 0          the CONSUMER: owns the 64 pixels' running state; per round it reads E parked pair results per pixel from LDS and
//                     applies the recurrence to them in list order (f3dg_pair_apply / _flat), nothing else;
//     waves 1 .. E    the EVALUATORS: every lane walks ITS pixel's pass mask of the current window; in round r evaluator e takes the
//                     pixel's entry of rank r E + e, reads its record, evaluates the stateless part and parks the result (six numbers,
//                     the colour, the slot) in a ring of K rounds;
//     wave E + 1      the PRODUCER of render3p: list scan, record gathers, phase 1 and the number of rounds of a window.
// No barrier after the first: the waves meet through monotonic LDS counters (release stores, acquire loads at workgroup scope) --
// windows published by the producer, windows finished by every evaluator and by the consumer (the producer reuses a record buffer
// when the window two before is finished), rounds published by every evaluator and rounds consumed (an evaluator runs at most K rounds
// ahead). An evaluator knows nothing of saturation: it evaluates every passing entry, the consumer ignores what lies behind a pixel's
// stop and raises a stop flag once all 64 pixels are done; every wait of the other waves watches that flag. Per pixel the sequence of
// blended entries and every operation on them is render3l's: bit-identical images and auxiliary planes.
struct R3qPark { float4 a[64]; float4 b[64]; float2 c[64]; };      // (alpha t m nn0) (nn1 nn2 r g) (b slot)

__device__ __forceinline__ void r3q_wait_ge(unsigned* flag, unsigned v)
{
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - v) < 0)
        __builtin_amdgcn_s_sleep(0);
}
// the same for the waves that must not outwait the consumer's stop: false = stopped
__device__ __forceinline__ bool r3q_wait_ge_or_stop(unsigned* flag, unsigned v, unsigned* stop)
{
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - v) < 0) {
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u)
            return false;
        __builtin_amdgcn_s_sleep(0);
    }
    return true;
}

template <bool SAVE_AUX, bool FAST, int E>
__global__ void __launch_bounds__(64 * (E + 2), 1)
render3q_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                    const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                    const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                    const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                    float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib, int prof)
{
    unsigned view, unit;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, 4u * (unsigned)T, view, unit);
    const unsigned tile = unit >> 2, quad = unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // option render_count: shader clocks of every role (f3dg_debug_render3q_clocks) -- g_f3dg_counts4 rows 32 + role: [0] total, [1] waiting
    // for a window (to be published / to be finished), [2] waiting for a round counter, [3] windows, [4] rounds, [5] waves; row 40: the
    // longest wave of every role
    const unsigned long long t_start = prof ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_win = 0ull, t_spin = 0ull, n_win = 0ull, n_rounds = 0ull;
#define R3Q_TIMED(acc, expr) do { const unsigned long long t0_ = prof ? __builtin_readcyclecounter() : 0ull; expr; if (prof) acc += __builtin_readcyclecounter() - t0_; } while (0)
#define R3Q_REPORT(role) do { if (prof && lane == 0) { const unsigned long long tt_ = __builtin_readcyclecounter() - t_start;                          \
        atomicAdd(&g_f3dg_counts4[32 + (role)][0], tt_); atomicAdd(&g_f3dg_counts4[32 + (role)][1], t_win); atomicAdd(&g_f3dg_counts4[32 + (role)][2], t_spin); \
        atomicAdd(&g_f3dg_counts4[32 + (role)][3], n_win); atomicAdd(&g_f3dg_counts4[32 + (role)][4], n_rounds); atomicAdd(&g_f3dg_counts4[32 + (role)][5], 1ull); \
        atomicMax(&g_f3dg_counts4[40][role], tt_); } } while (0)
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;

    constexpr int K = E == 2 ? 4 : 2;                     // rounds of parking (power of two): 20 KB (E = 2) / 15 KB (E = 3); four workgroups share a CU's 160 KB
    __shared__ float4 sR[2][4][F3DG_R4_WIN];              // two windows of records, [window][16-byte chunk][entry]
    __shared__ uint2 sQ[F3DG_R3U_RING];                   // (list position, Gaussian id) of the kept entries (the producer's ring)
    __shared__ unsigned long long sPass[2][64];           // per window: the pass mask of every pixel
    __shared__ unsigned sM[2], sHead[2], sRounds[2];      // per window: entries (0: the list has ended), first ring slot, rounds of E ranks
    __shared__ R3qPark sPark[K][E];
    __shared__ unsigned sProd;                            // windows published by the producer
    __shared__ unsigned sEvalWin[E], sConsWin;            // windows finished by evaluator e / by the consumer
    __shared__ unsigned sEvalDone[E], sConsumed;          // rounds published by evaluator e / consumed (over all windows)
    __shared__ unsigned sStop;                            // the consumer: every pixel of the quadrant is done

    if (threadIdx.x < (unsigned)E) { sEvalDone[threadIdx.x] = 0u; sEvalWin[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) { sConsumed = 0u; sConsWin = 0u; sProd = 0u; sStop = 0u; }
    __syncthreads();

    if (wave == (unsigned)E + 1u) {
        // ================================================ the producer: scan, gather, phase 1 ================================================
        uint2 range = ranges[(size_t)view * T + tile];
        if (hdr->overflow) range = make_uint2(0, 0);
        const unsigned n = range.y - range.x;
        const F3dgRec* vrec = rec + (size_t)view * P;
        const float4* vcull = cull + (size_t)view * P;
        const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
        const unsigned long long lt = (1ull << lane) - 1ull;
        unsigned cursor = 0, qhead = 0, qcount = 0;
        unsigned id0 = lane < n ? point_list[range.x + lane] : 0u;
        unsigned id1 = 64u + lane < n ? point_list[range.x + 64u + lane] : 0u;
        unsigned id2 = 128u + lane < n ? point_list[range.x + 128u + lane] : 0u;      // three 64-id chunks of the list in flight
        for (unsigned w = 0;; w++) {
            const unsigned buf = w & 1u;
            while (qcount < F3DG_R4_WIN && cursor < n) {
                const unsigned idm = id0, pos = cursor + lane;
                cursor += 64u;
                id0 = id1;
                id1 = id2;
                id2 = cursor + 128u + lane < n ? point_list[range.x + cursor + 128u + lane] : 0u;
                const bool keep = pos < n && (idm & qbit) != 0u;
                const unsigned long long kb = __ballot(keep);
                if (keep) sQ[(qhead + qcount + (unsigned)__popcll(kb & lt)) & (F3DG_R3U_RING - 1)] = make_uint2(pos, idm & F3DG_ID_MASK);
                qcount += (unsigned)__popcll(kb);
            }
            wave_lds_fence();
            const unsigned m = qcount < F3DG_R4_WIN ? qcount : F3DG_R4_WIN;
            // buffer `buf` held window w - 2: every reader of its records, pass masks and ring slots must have finished it
            if (w >= 2u) {
                bool go = true;
                const unsigned long long t0 = prof ? __builtin_readcyclecounter() : 0ull;
                for (int e = 0; e < E; e++) go = go && r3q_wait_ge_or_stop(&sEvalWin[e], w - 1u, &sStop);
                go = go && r3q_wait_ge_or_stop(&sConsWin, w - 1u, &sStop);
                if (prof) t_win += __builtin_readcyclecounter() - t0;
                if (!go) break;
            }
            float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (lane < m) {
                const unsigned id = sQ[(qhead + lane) & (F3DG_R3U_RING - 1)].y;
                const float4* src = reinterpret_cast<const float4*>(vrec + id);
#pragma unroll
                for (int c = 0; c < 4; c++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)&sR[buf][c][0], 16, 0, 0);
                e4 = vcull[id];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            unsigned rounds = 0;
            if (m != 0u) {
                const float ec = lane < m ? sR[buf][3][lane].w : 0.0f;
                int pass_lo = 0, pass_hi = 0;
                const float u0 = lane < m ? (float)qx0 - e4.x : __builtin_nanf("");
                const float v0 = (float)qy0 - e4.y;
                float dxx[8], adx[8], dyy[8], cdy[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    dxx[q] = u0 + (float)q;
                    adx[q] = e4.z * dxx[q];
                    dyy[q] = v0 + (float)q;
                    cdy[q] = ec * dyy[q] * dyy[q];
                }
                quad_ballots<0>(pass_lo, pass_hi, fmaf(dxx[0], fmaf(e4.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4.w);
                sPass[buf][lane] = ((unsigned long long)(unsigned)pass_hi << 32) | (unsigned)pass_lo;
                const unsigned most = (unsigned)__builtin_amdgcn_readfirstlane((int)__reduce_max_sync(~0ull, (int)(__popc((unsigned)pass_lo) + __popc((unsigned)pass_hi))));
                rounds = (most + (unsigned)E - 1u) / (unsigned)E;
            }
            if (lane == 0) { sM[buf] = m; sHead[buf] = qhead; sRounds[buf] = rounds; }
            wave_lds_fence();
            if (lane == 0) __hip_atomic_store(&sProd, w + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (m == 0u)
                break;
            n_win++;
            qhead += m;
            qcount -= m;
        }
        R3Q_REPORT(2);
        return;
    }

    const unsigned pix_x = qx0 + (lane & 7u), pix_y = qy0 + (lane >> 3);
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    if (wave != 0u) {
        // ================================================ evaluator e: the stateless part of rank r E + e ================================================
        const unsigned e = wave - 1u;
        unsigned rr = 0;                                      // rounds since the start (all windows)
        for (unsigned w = 0;; w++) {
            const unsigned buf = w & 1u;
            bool go;
            R3Q_TIMED(t_win, go = r3q_wait_ge_or_stop(&sProd, w + 1u, &sStop));
            if (!go) break;
            const unsigned m = sM[buf];
            if (m == 0u)
                break;
            const unsigned R = sRounds[buf];
            n_win++; n_rounds += R;
            unsigned long long pass = sPass[buf][lane];
            for (unsigned r = 0; r < R && go; r++, rr++) {
#pragma unroll
                for (int k = 0; k < E; k++)                   // (wave-uniform count: e ranks belong to the evaluators before this one)
                    if ((unsigned)k < e) pass &= pass - 1ull;
                const bool have = pass != 0ull;
                const int j = have ? __builtin_ctzll(pass) : 0;
                pass &= pass - 1ull;
#pragma unroll
                for (int k = 0; k < E - 1; k++)
                    if ((unsigned)k + e < (unsigned)E - 1u) pass &= pass - 1ull;
                const float4 q0 = sR[buf][0][j], q1 = sR[buf][1][j], q2 = sR[buf][2][j], q3 = sR[buf][3][j];
                F3dgPair pr = f3dg_pair_eval<FAST, true, true, FAST>(ray_x, ray_y, q0, q1, q2);
                if (!have) pr.alpha = 0.0f;
                R3Q_TIMED(t_spin, go = r3q_wait_ge_or_stop(&sConsumed, rr + 1u - (unsigned)K, &sStop));      // the ring slot of round rr - K has been read
                if (!go) break;
                R3qPark& pk = sPark[rr & (K - 1)][e];
                pk.a[lane] = make_float4(pr.alpha, pr.t, pr.m, pr.nn0);
                pk.b[lane] = make_float4(pr.nn1, pr.nn2, q3.x, q3.y);
                pk.c[lane] = make_float2(q3.z, __uint_as_float((unsigned)j));
                wave_lds_fence();
                if (lane == 0) __hip_atomic_store(&sEvalDone[e], rr + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (!go) break;
            if (lane == 0) __hip_atomic_store(&sEvalWin[e], w + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        R3Q_REPORT(1);
        return;
    }

    // ==================================================== the consumer: the pixels' recurrence ====================================================
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    bool done = !inside;
    F3dgPixel st;
    f3dg_pixel_init(st);
    {
        unsigned rr = 0;
        bool stop = false;
        for (unsigned w = 0; !stop; w++) {
            const unsigned buf = w & 1u;
            R3Q_TIMED(t_win, r3q_wait_ge(&sProd, w + 1u));
            const unsigned m = sM[buf];
            if (m == 0u)
                break;
            const unsigned R = sRounds[buf];
            n_win++; n_rounds += R;
            const unsigned rr_end = rr + R;
            for (; rr != rr_end; rr++) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    R3Q_TIMED(t_spin, r3q_wait_ge(&sEvalDone[e], rr + 1u));
                    const R3qPark& pk = sPark[rr & (K - 1)][e];
                    const float4 a = pk.a[lane], b = pk.b[lane];
                    const float2 c = pk.c[lane];
                    F3dgPair p;
                    p.alpha = a.x; p.t = a.y; p.m = a.z; p.nn0 = a.w; p.nn1 = b.x; p.nn2 = b.y;
                    const unsigned contributor = F3DG_R4_FLAG | __float_as_uint(c.y);
                    if (FAST) {
                        p.alpha = done ? 0.0f : p.alpha;
                        done |= f3dg_pair_apply_flat<true, true>(st, contributor, p, b.z, b.w, c.x);
                    } else if (!done && p.alpha != 0.0f) {
                        done = f3dg_pair_apply<false, true, true>(st, contributor, p, b.z, b.w, c.x);
                    }
                }
                wave_lds_fence();
                if (lane == 0) __hip_atomic_store(&sConsumed, rr + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__ballot(!done) == 0ull) {                 // nobody needs the rest of the list
                    stop = true;
                    break;
                }
            }
            if (SAVE_AUX) {             // slots -> 1-based list positions (the reference's `contributor`)
                const unsigned head = sHead[buf];
                if (st.last_contributor - F3DG_R4_FLAG < (unsigned)F3DG_R4_WIN)
                    st.last_contributor = sQ[(head + (st.last_contributor - F3DG_R4_FLAG)) & (F3DG_R3U_RING - 1)].x + 1u;
                if (st.max_contributor - F3DG_R4_FLAG < (unsigned)F3DG_R4_WIN)
                    st.max_contributor = sQ[(head + (st.max_contributor - F3DG_R4_FLAG)) & (F3DG_R3U_RING - 1)].x + 1u;
            }
            wave_lds_fence();
            if (lane == 0) {
                if (stop) __hip_atomic_store(&sStop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(&sConsWin, w + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    R3Q_REPORT(0);
#undef R3Q_TIMED
#undef R3Q_REPORT

    if (inside) {
        const float* bg = background + (bg_per_view ? 3 * view : 0);
        const float Tr = st.Tr;
        const float distortion_before_normalized = st.distortion;
        const float distortion = (float)(st.distortion / ((1 - Tr) * (1 - Tr) + 1e-7));

        if (SAVE_AUX) {
            float* fT = final_T + (size_t)view * 4 * HW;
            fT[pix_id] = Tr;
            fT[pix_id + HW] = st.dist1;
            fT[pix_id + 2 * HW] = st.dist2;
            fT[pix_id + 3 * HW] = distortion_before_normalized;
            unsigned* nc = n_contrib + (size_t)view * 2 * HW;
            nc[pix_id] = st.last_contributor;
            nc[pix_id + HW] = st.max_contributor;
        }
        float* out = out_color + (size_t)view * F3DG_OUT_CHANNELS * HW;
        out[0 * HW + pix_id] = st.C0 + Tr * bg[0];
        out[1 * HW + pix_id] = st.C1 + Tr * bg[1];
        out[2 * HW + pix_id] = st.C2 + Tr * bg[2];
        out[3 * HW + pix_id] = st.C3;
        out[4 * HW + pix_id] = st.C4;
        out[5 * HW + pix_id] = st.C5;
        out[6 * HW + pix_id] = st.C6;
        out[7 * HW + pix_id] = st.C7;
        out[8 * HW + pix_id] = distortion;
    }
}

} // namespace

int f3dg_launch_render_small(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                         const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                         float* out_color, int fast, int save_aux, float* final_T, unsigned* n_contrib, int unroll, int split, int count)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const dim3 grid((unsigned)V * (unsigned)T * 4u);
    if (split >= 2) {       // the pipeline of waves: consumer + E evaluators + producer
        const int E = split >= 3 ? 3 : 2;
#define F3DG_LAUNCH3Q(AUX, FST, EE) F3DG_KLAUNCH((render3q_fwd_kernel<AUX, FST, EE>), grid, dim3(64 * (EE + 2)), 0, s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, \
                                                 point_list, rec, cull, background, bg_per_view, out_color, final_T, n_contrib, count)
#define F3DG_LAUNCH3Q_E(AUX, FST) do { if (E == 3) F3DG_LAUNCH3Q(AUX, FST, 3); else F3DG_LAUNCH3Q(AUX, FST, 2); } while (0)
        if (save_aux) { if (fast) F3DG_LAUNCH3Q_E(true, true); else F3DG_LAUNCH3Q_E(true, false); }
        else { if (fast) F3DG_LAUNCH3Q_E(false, true); else F3DG_LAUNCH3Q_E(false, false); }
#undef F3DG_LAUNCH3Q_E
#undef F3DG_LAUNCH3Q
        snprintf(g_kernel_name4, sizeof g_kernel_name4, "render3q_fwd_kernel<SAVE_AUX=%s, FAST=%s, E=%d>", save_aux ? "true" : "false", fast ? "true" : "false", E);
        g_f3dg_last_render_kernel = g_kernel_name4;
        F3DG_HIP_CHECK(hipGetLastError());
        return F3DG_OK;
    }
    if (split) {
        const int U = unroll >= 2 ? 2 : 1;
#define F3DG_LAUNCH3P(AUX, FST, UU) F3DG_KLAUNCH((render3p_fwd_kernel<AUX, FST, UU>), grid, dim3(128), 0, s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, \
                                                 point_list, rec, cull, background, bg_per_view, out_color, final_T, n_contrib)
#define F3DG_LAUNCH3P_U(AUX, FST) do { if (U == 2) F3DG_LAUNCH3P(AUX, FST, 2); else F3DG_LAUNCH3P(AUX, FST, 1); } while (0)
        if (save_aux) { if (fast) F3DG_LAUNCH3P_U(true, true); else F3DG_LAUNCH3P_U(true, false); }
        else { if (fast) F3DG_LAUNCH3P_U(false, true); else F3DG_LAUNCH3P_U(false, false); }
#undef F3DG_LAUNCH3P_U
#undef F3DG_LAUNCH3P
        snprintf(g_kernel_name4, sizeof g_kernel_name4, "render3p_fwd_kernel<SAVE_AUX=%s, FAST=%s, U=%d>", save_aux ? "true" : "false", fast ? "true" : "false", U);
        g_f3dg_last_render_kernel = g_kernel_name4;
        F3DG_HIP_CHECK(hipGetLastError());
        return F3DG_OK;
    }
    (void)unroll;
    return F3DG_ERR_BAD_ARG;      // (split == 0 is render3l_fwd_kernel's: f3dg_launch_render does not come here)
}

int f3dg_launch_render4(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                        const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                        float* out_color, int fast, unsigned skip_channels, int count, int save_aux, float* final_T, unsigned* n_contrib)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const dim3 grid((unsigned)V * (unsigned)T * 4u);
    const bool lean = !save_aux && (skip_channels & (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION)) == (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION);
    const int th = g_f3dg_render_pack_th;
#define F3DG_R4_ARGS s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color, th, final_T, n_contrib
#define F3DG_LAUNCH4(FST, NRM, DST, CNT, AUX) F3DG_KLAUNCH((render4_fwd_kernel<FST, NRM, DST, CNT, AUX>), grid, dim3(64), 0, F3DG_R4_ARGS)
    if (save_aux) { if (fast) F3DG_LAUNCH4(true, true, true, false, true); else F3DG_LAUNCH4(false, true, true, false, true); }
    else if (count && !lean) { if (fast) F3DG_LAUNCH4(true, true, true, true, false); else F3DG_LAUNCH4(false, true, true, true, false); }
    else if (lean) { if (fast) F3DG_LAUNCH4(true, false, false, false, false); else F3DG_LAUNCH4(false, false, false, false, false); }
    else { if (fast) F3DG_LAUNCH4(true, true, true, false, false); else F3DG_LAUNCH4(false, true, true, false, false); }
#undef F3DG_LAUNCH4
#undef F3DG_R4_ARGS
    snprintf(g_kernel_name4, sizeof g_kernel_name4, "render4_fwd_kernel<FAST=%s, NORMAL=%s, DIST=%s%s%s, pack_th=%d>", fast ? "true" : "false",
             lean ? "false" : "true", lean ? "false" : "true", count && !lean && !save_aux ? ", COUNT=true" : "", save_aux ? ", SAVE_AUX=true" : "", th);
    g_f3dg_last_render_kernel = g_kernel_name4;
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// debug: the work counters of render4's counting variant (option render_count = 1), summed over all launches since the last reset:
// h_out[16] = { staged, scanned, fused trips, slides, lane-trips of fused trips, waves, packed batches, blend trips, pairs evaluated in
// dense trips, pairs that reached a blend trip, 0... }
extern "C" int f3dg_debug_render4_counts(unsigned long long* h_out, int reset)
{
    static unsigned long long rows[64][16];
    F3DG_HIP_CHECK(hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_f3dg_counts4), sizeof rows));
    if (h_out)
        for (int k = 0; k < 16; k++) {
            h_out[k] = 0;
            for (int r = 0; r < 32; r++) h_out[k] += rows[r][k];      // (rows 32..: render3q's role clocks)
        }
    if (reset) {
        memset(rows, 0, sizeof rows);
        F3DG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_f3dg_counts4), rows, sizeof rows));
    }
    return F3DG_OK;
}

// debug: the role clocks of render3q_fwd_kernel (option render_count = 1), summed over the launches since the last reset of
// f3dg_debug_render4_counts: h_out[4][16] = rows { consumer, evaluators, producer } x { shader clocks in total, at the window barrier,
// waiting for a counter, windows, rounds, waves }, row 3 = the longest { consumer, evaluator, producer } wave of any workgroup
extern "C" int f3dg_debug_render3q_clocks(unsigned long long* h_out)
{
    static unsigned long long rows[64][16];
    if (!h_out) return F3DG_ERR_BAD_ARG;
    F3DG_HIP_CHECK(hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_f3dg_counts4), sizeof rows));
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 16; k++) h_out[16 * r + k] = rows[32 + r][k];
    for (int k = 0; k < 16; k++) h_out[48 + k] = rows[40][k];
    return F3DG_OK;
}
