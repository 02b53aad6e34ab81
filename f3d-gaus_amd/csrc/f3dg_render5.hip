// f3dg_render5.hip -- compositing forward with SPLIT PIXELS: the entries of the pixels that hold a quadrant's window back are blended by
// helper lanes and combined with a segmented wave scan (per-call flag F3DG_FLAG_SCAN, option render_scan; fast arithmetic only).
//
// render3s_fwd_kernel (f3dg_render.hip) gives every pixel of an 8x8 quadrant a lane and lets it walk its own passing entries of the
// window through renderCUDA's loop body (reference RAST/cuda_rasterizer/forward.cu:493-583); a phase-2 trip costs the wave ~85 issue
// slots whether 64 pixels take part or 3, and the number of trips is the busiest pixel's. On pixel-aligned splats over a real depth
// map (visualize.py:293-340) the entries of a half-window lie on one iso-depth contour: the pixels under it get ~14 of them, the
// average pixel 3, and 57 % of the trips have fewer than 16 takers (lane utilisation 0.24). render4 (f3dg_render4.hip) moved the
// stateless two thirds of a pair's arithmetic to dense trips and kept the recurrence -- T, the accumulators -- serial in the lane that
// owns the pixel; in fast arithmetic the hand-over cost what that saved.
//
// Front-to-back compositing is a scan: with x_i = 1 - alpha_i the transmittance in front of entry i is T_front * prod_{j<i} x_j, and
// every accumulator is a sum of terms that depend on the pair and on that product only (the distortion term needs two more prefix
// sums). So this kernel keeps render3s's list scan / staging / phase 1 / sliding half-windows and its FUSED trips while more than
// `th` pixels take part in a trip, and then
//   * COMPACTS the pending entries (both halves of the window) of the pixels that still hold the older half back into a ring of
//     (run position, pixel, slot) triples in pixel-major order -- a scalar loop over those pixels: the pass mask of a pixel is read
//     into SGPRs, every lane that is a set bit of it writes its own triple at ring position tail + mbcnt(mask);
//   * runs DENSE BATCHES of 64 triples, one (pixel, entry) pair per lane: the pair's ray and the pixel's T_front (and distortion
//     prefix sums) come by ds_bpermute from the lane that owns the pixel, the stateless part is f3dg_pair_eval, then
//       - a segmented inclusive product scan of x over the lanes of a pixel's run gives T_before and test_T of every pair
//         (Hillis-Steele, 6 DPP steps: row_shr 1 2 4 8 inside the 16-lane rows, row_bcast 15 and 31 across them; a lane's
//         distance to the start of its run decides whether a step applies to it),
//       - the 1e-4 stop (forward.cu:543-548: the entry that would take T below 1e-4 is not blended and ends the pixel) becomes a
//         per-lane predicate on test_T, which is non-increasing along a run: every pair from the stopping one on gets weight 0,
//       - the weighted channels are summed by segmented scans, ONE v_fmac_f32_dpp per channel and step (x += shift(x) * flag),
//       - the owning lane pulls its run's totals from the run's last lane, its new T (or, if the run holds the stop, T_front minus
//         the blended weights: the transmittance in front of the stopping entry), and the depth of the last blended entry in front
//         of which T was still above 0.5 (the median-depth switch, forward.cu:571-575).
// Per pixel the SET of blended entries is the reference's; the sums are associated differently (a tree over the run instead of a
// chain), so the mode is not bit-identical to render3s: it is gated on its own at the north_star's 1e-4 against the oracle
// (tests/test_scan_mode_gpu.py). LDS per wave: 4 KB of records + 512 B id ring + 512 B triple ring.
#include "f3dg_blend.h"
#include "f3dg_ellipse.h"
#include "f3dg_producer.h"
#include "f3dg_segscan.h"

#include <stdio.h>
#include <string.h>

extern thread_local const char* g_f3dg_last_render_kernel;
int g_f3dg_render_scan = -1;          // option render_scan: -1 (default) = calls that ask for it (F3DG_FLAG_SCAN); 1 = every fast inference launch of the general path; 0 = never
int g_f3dg_render_scan_min = 4;       // option render_scan_min: stragglers that hold fewer than this many older-half entries each finish the slide in fused trips (0: always compact)
int g_f3dg_render_scan_th = 12;       // option render_scan_th: fused trips while more than this many pixels take part (64: every trip is compacted)
int g_f3dg_render_scan_lanes = 4;     // lab option render_scan_lanes: lanes per pixel of the one- and two-view kernel (4, or 2: measured 54.8 against 52.7 us)

// work counters (option render_count = 1; f3dg_debug_render5_counts): [0] staged entries, [1] scanned, [2] fused trips, [3] slides,
// [4] lane-trips of fused trips, [5] waves, [6] dense batches, [7] pairs in dense batches, [8] pixels compacted, [9] slides with a compaction
__device__ unsigned long long g_f3dg_counts5[64][16];

namespace {

#define F3DG_R5_WIN 64
#define F3DG_R5_RING 128
#ifndef F3DG_R5_OCC
#define F3DG_R5_OCC 8               // 59 VGPRs (52 without normals and distortion), 5 KB of LDS: 8 waves per SIMD
#endif

__device__ __forceinline__ void wave_lds_fence5()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float pull5(int addr, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}

template <bool NORMAL, bool DIST, bool COUNT>
__global__ void __launch_bounds__(64, F3DG_R5_OCC)
render5_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                   const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                   const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                   const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                   float* __restrict__ out_color, int th, int min_trips)
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

    __shared__ float4 sR[4][F3DG_R5_WIN];          // records, [16-byte chunk][slot]; slots 0..31 and 32..63 are the two halves of the window
    __shared__ unsigned sQ[F3DG_R5_RING];          // ids of kept entries not staged yet, ring
    __shared__ unsigned sK[F3DG_R5_RING];          // (position in the pixel's run << 12) | (owning lane << 6) | physical slot, ring

    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned hl = lane & 31u;               // entry of a half this lane tests in phase 1 ...
    const unsigned row4 = (lane >> 5) * 4u;       // ... against the pixels of rows row4 .. row4 + 3

    bool done = !inside;
    F3dgPixel st;
    f3dg_pixel_init(st);

    unsigned n_staged = 0, n_fused = 0, n_slides = 0, n_lane_fused = 0, n_batches = 0, n_batch_pairs = 0, n_compact = 0, n_cslides = 0;
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
            if (keep)
                sQ[(qhead + qpend + __builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u))) & (F3DG_R5_RING - 1)] = idm & F3DG_ID_MASK;
            qpend += (unsigned)__popcll(kb);
        }
        const unsigned m = qpend < 32u ? qpend : 32u;
        if (m == 0u && __ballot(pass != 0ull) == 0ull)
            break;                                // nothing left to stage, nothing left in the newer half
        wave_lds_fence5();

        // ---- stage m entries into the retired half; lanes e and e + 32 both take entry e
        const unsigned base = flip * 32u;
        float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float ec = 0.0f;
        if (hl < m) {
            const unsigned id = sQ[(qhead + hl) & (F3DG_R5_RING - 1)];
            if (lane < 32u) {
                const float4* src = reinterpret_cast<const float4*>(vrec + id);
#pragma unroll
                for (int c = 0; c < 4; c++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)&sR[c][base], 16, 0, 0);
            }
            e4 = vcull[id];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence5();
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
        while (pass != 0ull && __ballot((unsigned)pass != 0u) != 0ull && (int)__popcll(__ballot(true)) > th) {
            const unsigned j = (unsigned)__builtin_ctzll(pass) ^ xr;
            pass &= pass - 1;
            if (COUNT) n_lane_fused++;
            const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j], q3 = sR[3][j];
            const F3dgPair pr = f3dg_pair_eval<true, NORMAL, DIST, false>(ray_x, ray_y, q0, q1, q2);
            asm volatile("" :: "v"(q2.w), "v"(q3.w), "v"(pr.alpha));
            if (pr.alpha != 0.0f)
                done = f3dg_pair_apply<true, NORMAL, DIST>(st, 0u, pr, q3.x, q3.y, q3.z);
            if (done) pass = 0ull;
            if (COUNT && __builtin_ctzll(__ballot(true)) == (int)lane) n_fused++;
        }

        // ---- phase 2b: the pixels that still hold the older half back hand their pending entries to dense batches: all of the older
        // half (they gate the slide) and as many of the newer half, in pixel order, as fill the last batch -- those entries are resident,
        // would have to be blended later anyway, and a lane of a batch costs the same empty or not
        unsigned long long tk = __ballot((unsigned)pass != 0u);
        if (tk != 0ull && min_trips > 0) {
            // what the busiest straggler still holds in the older half = the fused trips the rest of this slide would take: a compaction +
            // a batch cost about four of them
            int rem = __popc((unsigned)pass);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rem = max(rem, __shfl_xor(rem, o, 64));
            if (__builtin_amdgcn_readfirstlane(rem) < min_trips) {
                while (pass != 0ull && __ballot((unsigned)pass != 0u) != 0ull) {
                    const unsigned j = (unsigned)__builtin_ctzll(pass) ^ xr;
                    pass &= pass - 1;
                    if (COUNT) n_lane_fused++;
                    const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j], q3 = sR[3][j];
                    const F3dgPair pr = f3dg_pair_eval<true, NORMAL, DIST, false>(ray_x, ray_y, q0, q1, q2);
                    asm volatile("" :: "v"(q2.w), "v"(q3.w), "v"(pr.alpha));
                    if (pr.alpha != 0.0f)
                        done = f3dg_pair_apply<true, NORMAL, DIST>(st, 0u, pr, q3.x, q3.y, q3.z);
                    if (done) pass = 0ull;
                    if (COUNT && __builtin_ctzll(__ballot(true)) == (int)lane) n_fused++;
                }
                tk = 0ull;
            }
        }
        if (tk != 0ull) {
            unsigned plo = (unsigned)pass, phi = (unsigned)(pass >> 32);
            unsigned budget;                        // newer-half entries the batches of this slide have room for
            {
                unsigned t = (unsigned)__popc(plo);  // older-half entries of this pixel; summed over the wave into lane 63
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, F3DG_DPP_ROW_SHR(1), 0xf, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, F3DG_DPP_ROW_SHR(2), 0xf, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, F3DG_DPP_ROW_SHR(4), 0xf, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, F3DG_DPP_ROW_SHR(8), 0xf, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, F3DG_DPP_BCAST15, 0xa, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, F3DG_DPP_BCAST31, 0xc, 0xf, true);
                const unsigned must = (unsigned)__builtin_amdgcn_readlane((int)t, 63);
                budget = (0u - must) & 63u;
            }
            unsigned rs = 0u, re = 0u;              // per owning lane: its run is ring positions [rs, re)
            unsigned qh = 0u, qt = 0u;              // wave-uniform ring counters of this slide
            if (COUNT) n_cslides++;
            do {
                // compaction: pixel after pixel until a batch is full
                while (tk != 0ull && qt - qh < 64u) {
                    const int p = __builtin_ctzll(tk);
                    tk &= tk - 1ull;
                    const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)plo, p), mhi = (unsigned)__builtin_amdgcn_readlane((int)phi, p);
                    const unsigned clo = (unsigned)__popc(mlo), chi = (unsigned)__popc(mhi);
                    const unsigned extra = chi < budget ? chi : budget;
                    budget -= extra;
                    const unsigned cnt = clo + extra;
                    const unsigned r = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                    const unsigned word = lane < 32u ? mlo : mhi;
                    const bool take = ((word >> (lane & 31u)) & 1u) != 0u && r < cnt;
                    if (take)
                        sK[(qt + r) & (F3DG_R5_RING - 1)] = (r << 12) | ((unsigned)p << 6) | (lane ^ xr);
                    const unsigned left = mhi & ~(unsigned)(__ballot(take) >> 32);       // newer-half entries that stay pending
                    // lane p: its run's ring positions; its pending mask loses what was taken (v_writelane_b32 ignores EXEC; lane in M0)
                    asm volatile("s_mov_b32 m0, %[p]\n\t"
                                 "s_nop 0\n\t"
                                 "v_writelane_b32 %[rs], %[a], m0\n\t"
                                 "v_writelane_b32 %[re], %[b], m0\n\t"
                                 "v_writelane_b32 %[lo], 0, m0\n\t"
                                 "v_writelane_b32 %[hi], %[l], m0"
                                 : [rs] "+v"(rs), [re] "+v"(re), [lo] "+v"(plo), [hi] "+v"(phi)
                                 : [p] "s"(p), [a] "s"(qt), [b] "s"(qt + cnt), [l] "s"(left)
                                 : "m0");
                    qt += cnt;
                    if (COUNT) n_compact++;
                }
                const unsigned nb = qt - qh < 64u ? qt - qh : 64u;
                wave_lds_fence5();
                if (COUNT) { n_batches++; n_batch_pairs += nb; }

                // ---- one dense batch: lane q takes triple q of the ring
                const unsigned kk = sK[(qh + lane) & (F3DG_R5_RING - 1)];
                const bool valid = lane < nb;
                // (lanes beyond the batch re-evaluate pair 0 -- finite numbers -- with alpha forced to 0, each its own run)
                const unsigned k = valid ? kk : (unsigned)__builtin_amdgcn_readfirstlane((int)kk);
                const unsigned rrun = k >> 12;
                const unsigned rr = valid ? (rrun < lane ? rrun : lane) : 0u;       // lanes between this pair and the start of its run INSIDE the batch
                const int oaddr = (int)(((k >> 6) & 63u) << 2);
                const unsigned j = k & 63u;
                const float rx = pull5(oaddr, ray_x), ry = pull5(oaddr, ray_y);
                const float Tf = pull5(oaddr, done ? 0.0f : st.Tr);                    // a pixel that has stopped: every later pair of it is killed
                float D1 = 0.0f, D2 = 0.0f;
                if (DIST) { D1 = pull5(oaddr, st.dist1); D2 = pull5(oaddr, st.dist2); }
                const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j], q3 = sR[3][j];
                const F3dgPair pr = f3dg_pair_eval<true, NORMAL, DIST, true>(rx, ry, q0, q1, q2);
                asm volatile("" :: "v"(q2.w), "v"(q3.w), "v"(pr.alpha));
                const float alpha = valid ? pr.alpha : 0.0f;

                const SegFlags sf = seg_flags(rr, lane);
                const float Pi = seg_product(1.0f - alpha, sf);                          // prod over the run up to and including this pair
                const float Psh = F3DG_DPP(1.0f, Pi, F3DG_DPP_WAVE_SHR1, 0xf);
                const float Pe = sf.c1 ? Psh : 1.0f;                                      // ... excluding it
                const float Tb = Tf * Pe, tT = Tf * Pi;                                  // T in front of the pair, test_T (forward.cu:543)
                const bool killed = !(tT >= 0.0001f);                                    // the stop, or a pair behind it
                const float w = killed ? 0.0f : alpha * Tb;

                constexpr int KM = 4 + (NORMAL ? 3 : 0) + (DIST ? 1 : 0);
                static_assert(KM == 4 || KM == 8, "seg_sums is written for the lean (rgb, alpha) and the full channel set");
                float v[KM];
                v[0] = q3.x * w; v[1] = q3.y * w; v[2] = q3.z * w; v[3] = w;
                if constexpr (NORMAL) { v[4] = pr.nn0 * w; v[5] = pr.nn1 * w; v[6] = pr.nn2 * w; }
                float a[2] = { 0.0f, 0.0f };
                if constexpr (DIST) {
                    // forward.cu:552-557: error = m^2 (1 - T) + dist2 - 2 m dist1 with the running sums IN FRONT of the pair
                    const float mw = pr.m * w, m2w = pr.m * mw;
                    a[0] = mw; a[1] = m2w;
                    seg_sums(a, sf);
                    const float E1 = (a[0] - mw) + D1, E2 = (a[1] - m2w) + D2;
                    const float error = fmaf(-2.0f * pr.m, E1, fmaf(pr.m * pr.m, 1.0f - Tb, E2));
                    v[KM - 1] = error * w;
                }
                seg_sums(v, sf);

                // ---- the owning lanes take their run's totals from its last lane in this batch
                const bool mine_in = rs < qh + nb && re > qh;
                const unsigned lastq = (re < qh + nb ? re : qh + nb) - 1u - qh;
                const int saddr = (int)((lastq & 63u) << 2);
                const float tTl = pull5(saddr, tT);
                float sv[KM];
#pragma unroll
                for (int c = 0; c < KM; c++) sv[c] = pull5(saddr, v[c]);
                float s1 = 0.0f, s2 = 0.0f;
                if (DIST) { s1 = pull5(saddr, a[0]); s2 = pull5(saddr, a[1]); }
                // median depth (forward.cu:571-575): the last blended pair of the run in front of which T was still above 0.5
                const unsigned long long G = __ballot(!killed && alpha != 0.0f && Tb > 0.5f);
                if (G != 0ull) {
                    const unsigned firstq = (rs > qh ? rs : qh) - qh;
                    const unsigned long long sel = mine_in ? ((G >> (firstq & 63u)) << (firstq & 63u)) & (~0ull >> (63u - (lastq & 63u))) : 0ull;
                    const unsigned Lq = sel != 0ull ? 63u - (unsigned)__builtin_clzll(sel) : 0u;
                    const float tl = pull5((int)(Lq << 2), pr.t);
                    if (sel != 0ull) st.C6 = tl;
                }
                if (mine_in) {
                    st.C0 += sv[0]; st.C1 += sv[1]; st.C2 += sv[2]; st.C7 += sv[3];
                    if constexpr (NORMAL) { st.C3 += sv[4]; st.C4 += sv[5]; st.C5 += sv[6]; }
                    if constexpr (DIST) { st.dist1 += s1; st.dist2 += s2; st.distortion += sv[KM - 1]; }
                    const bool stop = !(tTl >= 0.0001f);
                    // the run holds the stop: T in front of the stopping entry = T_front - the weights blended before it
                    st.Tr = stop ? st.Tr - sv[3] : tTl;
                    done = done || stop;
                }
                qh += nb;
                wave_lds_fence5();          // the next compaction overwrites ring slots this batch has read
            } while (tk != 0ull || qt != qh);
            pass = done ? 0ull : ((unsigned long long)phi << 32) | plo;      // (a pixel that stopped in a batch drops the newer-half entries it kept)
        }

        if (__ballot(!done) == 0ull)
            break;
    }
    if (COUNT) {
        unsigned a = n_lane_fused;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += (unsigned)__shfl_xor((int)a, o, 64);
        unsigned f = n_fused;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) f += (unsigned)__shfl_xor((int)f, o, 64);
        if (lane == 0) {
            unsigned long long* c = g_f3dg_counts5[blockIdx.x & 63u];
            atomicAdd(&c[0], (unsigned long long)n_staged);
            atomicAdd(&c[1], (unsigned long long)(cursor < n ? cursor : n));
            atomicAdd(&c[2], (unsigned long long)f);
            atomicAdd(&c[3], (unsigned long long)n_slides);
            atomicAdd(&c[4], (unsigned long long)a);
            atomicAdd(&c[5], 1ull);
            atomicAdd(&c[6], (unsigned long long)n_batches);
            atomicAdd(&c[7], (unsigned long long)n_batch_pairs);
            atomicAdd(&c[8], (unsigned long long)n_compact);
            atomicAdd(&c[9], (unsigned long long)n_cslides);
        }
    }

    unsigned lane_e = threadIdx.x;
    asm volatile("" : "+v"(lane_e));
    const unsigned out_x = qx0 + (lane_e & 7u), out_y = qy0 + (lane_e >> 3);
    if (out_x < (unsigned)W && out_y < (unsigned)H) {
        const size_t HW = (size_t)H * W;
        const size_t pix_id = (size_t)W * out_y + out_x;
        const float* bg = background + (bg_per_view ? 3 * view : 0);
        const float Tr = st.Tr;
        const float distortion = (float)(st.distortion / ((1 - Tr) * (1 - Tr) + 1e-7));
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


// ---- SMALL launches (one or two views: every wave alone on its SIMD) in the split-pixel arithmetic --------------------------------------
// A one-view launch is 1,024 quadrants on 1,024 SIMDs: a quadrant lasts as long as its pixels' chains of dependent instructions (render3p,
// f3dg_render4.hip: a producer wave prepares the next window while a consumer wave composites, ~85 dependent instructions per entry and
// pixel). Here every pixel gets FOUR lanes (LPP; two were measured slower, eight do not fit the CU's wave slots): a workgroup is the producer
// wave of render3p (f3dg_producer.h) + four consumer waves of 16 pixels x 4 lanes; a trip takes the pixel's next four passing entries, one per lane, evaluates their stateless parts side by side
// (f3dg_pair_eval), and a two-step segmented product over the quad (DPP quad_perm) gives every lane the transmittance in front of ITS
// entry -- hence its weight, the 1e-4 stop as a per-lane predicate, the median-depth candidate. Nothing else crosses lanes per trip: each
// lane adds its own contributions to its own accumulators (the distortion term needs the quad's prefix sums of m w and m^2 w), and the four
// partial sums of a pixel meet once, at the end. ~150 dependent instructions per FOUR entries. The arithmetic class of F3DG_FLAG_SCAN (same
// blended entries per pixel, sums associated differently): gated with it (tests/test_scan_mode_gpu.py).
#define F3DG_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define F3DG_QDPP(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false))

template <int LPP>          // lanes per pixel: 4 (four consumer waves of 16 pixels) or 2 (two consumer waves of 32 pixels)
__global__ void __launch_bounds__(64 * (LPP + 1), 1)
render5p_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                    const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                    const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                    const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                    float* __restrict__ out_color)
{
    unsigned view, unit;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, 4u * (unsigned)T, view, unit);
    const unsigned tile = unit >> 2, quad = unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = threadIdx.x >> 6;                 // 0..LPP-1: consumers, LPP: the producer (wave-uniform)
    const bool producer = wv == (unsigned)LPP;
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;

    __shared__ float4 sR[3][4][F3DG_PROD_WIN];            // three windows of records, [window % 3][16-byte chunk][entry]
    __shared__ uint2 sQ[F3DG_PROD_RING];                   // (list position, Gaussian id) of the kept entries (the producer's ring)
    __shared__ unsigned long long sPass[2][64];           // per window: the pass mask of every pixel
    __shared__ uint2 sMH[2];                              // per window: (its number of entries (0: the list has ended), its first ring slot)
    __shared__ unsigned sStop[2];                         // [b]: bit w set by consumer wave w when its 16 pixels were done after the window in buffer b

    if (threadIdx.x < 2u) sStop[threadIdx.x] = 0u;

    if (producer) {
        // ================================================ wave 4: scan, gather, phase 1 (f3dg_producer.h) ================================
        f3dg_window_producer(lane, view, tile, quad, qx0, qy0, P, T, hdr, ranges, point_list, rec, cull, sR, sQ, sPass, sMH, sStop, (1u << LPP) - 1u);
        return;
    }

    // ================================================== waves 0..3: 16 pixels x 4 lanes each ==================================================
    const unsigned sub = lane & (unsigned)(LPP - 1);      // which of the pixel's next LPP entries this lane takes
    const unsigned q = (64u / LPP) * wv + lane / LPP;     // the pixel, 0..63 in the quadrant
    const unsigned pix_x = qx0 + (q & 7u), pix_y = qy0 + (q >> 3);
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);
    bool done = !inside;
    float Tf = 1.0f, D1 = 0.0f, D2 = 0.0f;               // per pixel (the same in its four lanes): T, the distortion's running sums
    float C0 = 0, C1 = 0, C2 = 0, C3 = 0, C4 = 0, C5 = 0, C7 = 0, Cd = 0;      // per LANE: the sums of its own entries
    float med_t = 0.0f;                                   // ... and the depth of its last entry in front of which T was above 0.5,
    unsigned med_rank = 0u;                               // with that entry's rank in the walk (0: none)
    unsigned rank_base = 0u;
    {
        unsigned buf = 0, rb = 0;                          // window k: pass masks in [k & 1], records in [k % 3]
        for (;;) {
            __syncthreads();                                  // window `buf` is ready
            const unsigned m = sMH[buf].x;
            if (m == 0u || sStop[buf ^ 1u] == (1u << LPP) - 1u)
                break;
            unsigned long long pass = done ? 0ull : sPass[buf][q];
            while (pass != 0ull) {
                // the pixel's next LPP passing entries: lane `sub` takes the one of that rank
                unsigned long long mine;
                {
                    const unsigned long long p1 = pass & (pass - 1ull);
                    if constexpr (LPP == 4) {
                        const unsigned long long p2 = p1 & (p1 - 1ull), p3 = p2 & (p2 - 1ull);
                        mine = sub == 0u ? pass : sub == 1u ? p1 : sub == 2u ? p2 : p3;
                        pass = p3 & (p3 - 1ull);
                    } else {
                        mine = sub == 0u ? pass : p1;
                        pass = p1 & (p1 - 1ull);
                    }
                }
                const bool have = mine != 0ull;
                const int j = have ? __builtin_ctzll(mine) : 0;
                const float4 q0 = sR[rb][0][j], q1 = sR[rb][1][j], q2 = sR[rb][2][j], q3 = sR[rb][3][j];
                const F3dgPair pr = f3dg_pair_eval<true, true, true, true>(ray_x, ray_y, q0, q1, q2);
                asm volatile("" :: "v"(q2.w), "v"(q3.w), "v"(pr.alpha));
                const float alpha = have ? pr.alpha : 0.0f;
                // transmittance in front of this lane's entry: T of the pixel x the product over the group's earlier lanes
                // (DPP quad_perm patterns: SHR1 = the lane before within the group, LAST = the group's last lane, swaps for the sums)
                constexpr int SHR1 = LPP == 4 ? F3DG_QP(0, 0, 1, 2) : F3DG_QP(0, 0, 2, 2);
                constexpr int LAST = LPP == 4 ? F3DG_QP(3, 3, 3, 3) : F3DG_QP(1, 1, 3, 3);
                float Pi = 1.0f - alpha;
                float t = F3DG_QDPP(Pi, SHR1);
                float Psh;
                if constexpr (LPP == 4) {
                    Pi *= sub >= 1u ? t : 1.0f;
                    t = F3DG_QDPP(Pi, F3DG_QP(0, 1, 0, 1));
                    Pi *= sub >= 2u ? t : 1.0f;
                    Psh = F3DG_QDPP(Pi, SHR1);
                } else {
                    Psh = t;                                 // lane 1's exclusive product is lane 0's factor
                    Pi *= sub >= 1u ? t : 1.0f;
                }
                const float Tb = Tf * (sub >= 1u ? Psh : 1.0f), tT = Tf * Pi;
                const bool killed = !(tT >= 0.0001f);         // the stop (forward.cu:543-548), or an entry behind it
                const float w = killed ? 0.0f : alpha * Tb;
                C0 = fmaf(q3.x, w, C0); C1 = fmaf(q3.y, w, C1); C2 = fmaf(q3.z, w, C2); C7 += w;
                C3 = fmaf(pr.nn0, w, C3); C4 = fmaf(pr.nn1, w, C4); C5 = fmaf(pr.nn2, w, C5);
                // distortion (forward.cu:552-557): the running sums IN FRONT of this entry = the pixel's + the group's earlier lanes'
                const float a1 = pr.m * w, a2 = pr.m * a1;
                float s1 = a1, s2 = a2;
                t = F3DG_QDPP(s1, SHR1); s1 += sub >= 1u ? t : 0.0f;
                t = F3DG_QDPP(s2, SHR1); s2 += sub >= 1u ? t : 0.0f;
                if constexpr (LPP == 4) {
                    t = F3DG_QDPP(s1, F3DG_QP(0, 1, 0, 1)); s1 += sub >= 2u ? t : 0.0f;
                    t = F3DG_QDPP(s2, F3DG_QP(0, 1, 0, 1)); s2 += sub >= 2u ? t : 0.0f;
                }
                const float E1 = (s1 - a1) + D1, E2 = (s2 - a2) + D2;
                Cd = fmaf(fmaf(-2.0f * pr.m, E1, fmaf(pr.m * pr.m, 1.0f - Tb, E2)), w, Cd);
                D1 += F3DG_QDPP(s1, LAST);
                D2 += F3DG_QDPP(s2, LAST);
                if (!killed && alpha != 0.0f && Tb > 0.5f) { med_t = pr.t; med_rank = rank_base + (unsigned)j + 1u; }
                // the pixel's new T; if one of the group stopped: T in front of the stopping entry = T - the weights blended before it
                float ws = w;
                ws += F3DG_QDPP(ws, F3DG_QP(1, 0, 3, 2));
                if constexpr (LPP == 4) ws += F3DG_QDPP(ws, F3DG_QP(2, 3, 0, 1));
                const float tT3 = F3DG_QDPP(tT, LAST);
                const bool stop = !(tT3 >= 0.0001f);
                Tf = stop ? Tf - ws : tT3;
                if (stop) { done = true; pass = 0ull; }
            }
            rank_base += 64u;
            if (__ballot(!done) == 0ull && lane == 0) atomicOr(&sStop[buf], 1u << wv);      // (read by all waves after the next barrier)
            buf ^= 1u;
            rb = rb == 2u ? 0u : rb + 1u;
        }
    }
    // the four partial sums of a pixel meet; the median depth is the candidate of the highest rank
#define F3DG_QSUM(x) do { x += F3DG_QDPP(x, F3DG_QP(1, 0, 3, 2)); if constexpr (LPP == 4) x += F3DG_QDPP(x, F3DG_QP(2, 3, 0, 1)); } while (0)
    F3DG_QSUM(C0); F3DG_QSUM(C1); F3DG_QSUM(C2); F3DG_QSUM(C3); F3DG_QSUM(C4); F3DG_QSUM(C5); F3DG_QSUM(C7); F3DG_QSUM(Cd);
#undef F3DG_QSUM
    {
        unsigned r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)med_rank, F3DG_QP(1, 0, 3, 2), 0xf, 0xf, false);
        float tt = F3DG_QDPP(med_t, F3DG_QP(1, 0, 3, 2));
        if (r > med_rank) { med_rank = r; med_t = tt; }
        if constexpr (LPP == 4) {
            r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)med_rank, F3DG_QP(2, 3, 0, 1), 0xf, 0xf, false);
            tt = F3DG_QDPP(med_t, F3DG_QP(2, 3, 0, 1));
            if (r > med_rank) { med_rank = r; med_t = tt; }
        }
    }
    if (inside && sub == 0u) {
        const size_t HW = (size_t)H * W;
        const size_t pix_id = (size_t)W * pix_y + pix_x;
        const float* bg = background + (bg_per_view ? 3 * view : 0);
        const float distortion = (float)(Cd / ((1 - Tf) * (1 - Tf) + 1e-7));
        float* out = out_color + (size_t)view * F3DG_OUT_CHANNELS * HW;
        out[0 * HW + pix_id] = C0 + Tf * bg[0];
        out[1 * HW + pix_id] = C1 + Tf * bg[1];
        out[2 * HW + pix_id] = C2 + Tf * bg[2];
        out[3 * HW + pix_id] = C3;
        out[4 * HW + pix_id] = C4;
        out[5 * HW + pix_id] = C5;
        out[6 * HW + pix_id] = med_t;
        out[7 * HW + pix_id] = C7;
        out[8 * HW + pix_id] = distortion;
    }
}

thread_local char g_kernel_name5[160] = "";

} // namespace

int f3dg_launch_render5(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                        const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                        float* out_color, unsigned skip_channels, int count)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const dim3 grid((unsigned)V * (unsigned)T * 4u);
    const bool lean = (skip_channels & (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION)) == (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION);
    const int th = g_f3dg_render_scan_th;
#define F3DG_LAUNCH5(NRM, DST, CNT) F3DG_KLAUNCH((render5_fwd_kernel<NRM, DST, CNT>), grid, dim3(64), 0, s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, \
                                                 point_list, rec, cull, background, bg_per_view, out_color, th, g_f3dg_render_scan_min)
    if (lean) { if (count) F3DG_LAUNCH5(false, false, true); else F3DG_LAUNCH5(false, false, false); }
    else { if (count) F3DG_LAUNCH5(true, true, true); else F3DG_LAUNCH5(true, true, false); }
#undef F3DG_LAUNCH5
    snprintf(g_kernel_name5, sizeof g_kernel_name5, "render5_fwd_kernel<NORMAL=%s, DIST=%s%s, scan_th=%d>", lean ? "false" : "true", lean ? "false" : "true",
             count ? ", COUNT=true" : "", th);
    g_f3dg_last_render_kernel = g_kernel_name5;
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// one- and two-view launches with F3DG_FLAG_SCAN: four lanes per pixel (render5p_fwd_kernel)
int f3dg_launch_render5_small(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                              const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                              float* out_color)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const int lpp = g_f3dg_render_scan_lanes == 2 ? 2 : 4;
    if (lpp == 4)
        F3DG_KLAUNCH(render5p_fwd_kernel<4>, dim3((unsigned)V * (unsigned)T * 4u), dim3(320), 0, s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges,
                     point_list, rec, cull, background, bg_per_view, out_color);
    else
        F3DG_KLAUNCH(render5p_fwd_kernel<2>, dim3((unsigned)V * (unsigned)T * 4u), dim3(192), 0, s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges,
                     point_list, rec, cull, background, bg_per_view, out_color);
    snprintf(g_kernel_name5, sizeof g_kernel_name5, "render5p_fwd_kernel<%d lanes per pixel>", lpp);
    g_f3dg_last_render_kernel = g_kernel_name5;
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// debug: the work counters of render5's counting variant (option render_count = 1), summed over all launches since the last reset:
// h_out[16] = { staged, scanned, fused trips, slides, lane-trips of fused trips, waves, dense batches, pairs in dense batches, pixels
// compacted, slides with a compaction, 0... }
extern "C" int f3dg_debug_render5_counts(unsigned long long* h_out, int reset)
{
    static unsigned long long rows[64][16];
    F3DG_HIP_CHECK(hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_f3dg_counts5), sizeof rows));
    if (h_out)
        for (int k = 0; k < 16; k++) {
            h_out[k] = 0;
            for (int r = 0; r < 64; r++) h_out[k] += rows[r][k];
        }
    if (reset) {
        memset(rows, 0, sizeof rows);
        F3DG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_f3dg_counts5), rows, sizeof rows));
    }
    return F3DG_OK;
}
