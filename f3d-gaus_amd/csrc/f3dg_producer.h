// The PRODUCER wave of the one- and two-view compositing kernels (render3p_fwd_kernel in f3dg_render4.hip, render5p_fwd_kernel in
// f3dg_render5.hip): everything of a quadrant's walk that does not depend on a pixel's state. Per window of up to 64 kept entries:
//   scan      64-id chunks of the tile's list (three in flight), entries whose quadrant bit is set go to a ring of (list position, id);
//   gather    the 64-byte records of the window's entries, global_load_lds straight into LDS ([16-byte chunk][entry]) + the cull
//             record (centre, conic) into registers;
//   phase 1   the conservative ellipse test of the 64 entries against the 64 pixels: 64 ballots, the pass mask of pixel q lands in
//             lane q (quad_ballots, f3dg_ellipse.h) and is left in LDS for the consumers;
//   barrier   the window is handed over; the consumers have finished the window before it.
// The reference has no such split: a CUDA block fetches 256 records cooperatively and every thread tests its own pixel
// (submodules/diff-gof-rasterization/cuda_rasterizer/forward.cu:452-478).
//
// Round 6: the gathers of window k + 1 are REQUESTED before phase 1 of window k runs, into a third record buffer. With nothing passing
// (lab option debug_skip_all, --tile-cull 0) the one-view launch still took 35.6 us of its 52-61: a window cost the producer
// latency(gather) + phase 1 (~330 dependent instructions), back to back. Now the ~1.5 us a gather is in flight lie behind the ballots of
// the window before. Which entries form a window, their order, the pass masks: unchanged, so the consumers see the same data.
//
// LDS the caller declares (shared by the producer and the consumers):
//   float4 sR[3][4][64]          three windows of records
//   uint2  sQ[F3DG_PROD_RING]    the ring; a window's entries stay in it until the consumers are done with the window (slot -> list
//                                position for the auxiliary planes). Occupancy: window k - 1 (being composited) + k (phase 1) +
//                                k + 1 (in flight) + a scan that stops at the first chunk reaching 64 (at most 63 + 64 - 64 more than a
//                                window): 64 + 64 + 127 = 255 <= 256.
//   u64    sPass[2][64], uint2 sMH[2]                 per window (k & 1): pass masks, (entry count (0: end of list), first ring slot)
//   unsigned sStop[2]            sStop[b] == stop_full: every pixel was done after the window with (k & 1) == b
// Window k lives in sR[k % 3]; consumers keep their own k.
#pragma once
#include "f3dg_common.h"
#include "f3dg_ellipse.h"

#define F3DG_PROD_RING 256
#define F3DG_PROD_WIN 64

__device__ __forceinline__ void f3dg_prod_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS accesses the compiler must not see: it cannot tell the buffers of an LDS array apart, so every ds_read / ds_write it schedules
// while a global_load_lds is outstanding gets an s_waitcnt vmcnt(0) in front (SIInsertWaitcnts) -- which is exactly the wait this
// producer wants to postpone. Between requesting window k + 1 and waiting for it, the producer touches LDS only through these.
__device__ __forceinline__ unsigned f3dg_lds_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void f3dg_lds_store64(const void* p, unsigned lo, unsigned hi)
{
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    asm volatile("ds_write_b64 %0, %1" :: "v"(f3dg_lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned f3dg_lds_load32_wait(const void* p)
{
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(f3dg_lds_addr(p)) : "memory");
    return v;
}
// s_barrier with the LDS traffic drained and the record gathers left in flight (what __syncthreads would not do)
__device__ __forceinline__ void f3dg_barrier_keep_vm()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void f3dg_window_producer(unsigned lane, unsigned view, unsigned tile, unsigned quad, unsigned qx0, unsigned qy0,
                                                     int P, int T, const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                                                     const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                                                     const float4* __restrict__ cull, float4 (*sR)[4][F3DG_PROD_WIN], uint2* sQ,
                                                     unsigned long long (*sPass)[64], uint2* sMH, const unsigned* sStop,
                                                     unsigned stop_full)
{
    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const unsigned n = range.y - range.x;
    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned cursor = 0, qcount = 0;                  // qcount: kept entries behind the window whose gathers were requested last
    unsigned id0 = lane < n ? point_list[range.x + lane] : 0u;
    unsigned id1 = 64u + lane < n ? point_list[range.x + 64u + lane] : 0u;
    unsigned id2 = 128u + lane < n ? point_list[range.x + 128u + lane] : 0u;      // three 64-id chunks of the list in flight

    // scan until `tail` (first free ring slot = qnext + qcount) holds a full window or the list ends
    auto scan = [&](unsigned qnext) {
        while (qcount < F3DG_PROD_WIN && cursor < n) {
            const unsigned idm = id0, pos = cursor + lane;
            cursor += 64u;
            id0 = id1;
            id1 = id2;
            id2 = cursor + 128u + lane < n ? point_list[range.x + cursor + 128u + lane] : 0u;
            const bool keep = pos < n && (idm & qbit) != 0u;
            const unsigned long long kb = __ballot(keep);
            if (keep) sQ[(qnext + qcount + (unsigned)__popcll(kb & lt)) & (F3DG_PROD_RING - 1)] = make_uint2(pos, idm & F3DG_ID_MASK);
            qcount += (unsigned)__popcll(kb);
        }
        f3dg_prod_fence();
    };
    // request the records of ring entries [head, head + m) into record buffer rb; returns the lane's cull record (centre, conic a b)
    // and conic c (the last float of the record: a register copy, so that phase 1 reads no LDS)
    auto gather = [&](unsigned head, unsigned m, unsigned rb, float& ec) -> float4 {
        float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        ec = 0.0f;
        if (lane < m) {
            const unsigned id = sQ[(head + lane) & (F3DG_PROD_RING - 1)].y;
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
#pragma unroll
            for (int c = 0; c < 4; c++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                 (__attribute__((address_space(3))) void*)&sR[rb][c][0], 16, 0, 0);
            e4 = vcull[id];
            ec = reinterpret_cast<const float*>(vrec + id)[15];
        }
        return e4;
    };

    // vmcnt(0) as an instruction the compiler models (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15): after it its scoreboard knows that
    // nothing is pending, and it adds no waits of its own in front of phase 1
#define F3DG_PROD_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

    unsigned qhead = 0, rb = 0, pb = 0;
    scan(0u);
    unsigned m = qcount < F3DG_PROD_WIN ? qcount : F3DG_PROD_WIN;
    // one window: (e4c, ecc) are its cull records, loaded a window ago; (e4n, ecn) receive the next window's. Two register sets that
    // swap roles (the loop below is unrolled by two): copying next -> current at the end of a trip would wait for the loads.
    auto window = [&](const float4& e4c, const float& ecc, float4& e4n, float& ecn) -> bool {
        // the window after this one: its ids (ring), then -- once this window's records have arrived -- its gathers
        const unsigned qnext = qhead + m;
        scan(qnext);
        const unsigned m_next = qcount < F3DG_PROD_WIN ? qcount : F3DG_PROD_WIN;
        F3DG_PROD_WAIT_VM0();
        f3dg_prod_fence();
        const unsigned rb_next = rb == 2u ? 0u : rb + 1u;
        e4n = gather(qnext, m_next, rb_next, ecn);        // (buffer of window k - 2: the consumers left it before the last barrier)
        qcount -= m_next;
        if (m != 0u) {
            int pass_lo = 0, pass_hi = 0;
            const float u0 = lane < m ? (float)qx0 - e4c.x : __builtin_nanf("");
            const float v0 = (float)qy0 - e4c.y;
            float dxx[8], adx[8], dyy[8], cdy[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                dxx[q] = u0 + (float)q;
                adx[q] = e4c.z * dxx[q];
                dyy[q] = v0 + (float)q;
                cdy[q] = ecc * dyy[q] * dyy[q];
            }
            quad_ballots<0>(pass_lo, pass_hi, fmaf(dxx[0], fmaf(e4c.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4c.w);
            f3dg_lds_store64(&sPass[pb][lane], (unsigned)pass_lo, (unsigned)pass_hi);
        }
        if (lane == 0) f3dg_lds_store64(&sMH[pb], m, qhead);
        f3dg_barrier_keep_vm();                           // window k is ready; the consumers have finished window k - 1
        if (m == 0u || f3dg_lds_load32_wait(&sStop[pb ^ 1u]) == stop_full)       // (the flag of the window composited before this barrier)
            return true;
        qhead = qnext;
        m = m_next;
        rb = rb_next;
        pb ^= 1u;
        return false;
    };
    float ecA, ecB = 0.0f;
    float4 e4A = gather(0u, m, 0u, ecA), e4B = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    qcount -= m;
    for (;;) {
        if (window(e4A, ecA, e4B, ecB)) break;
        if (window(e4B, ecB, e4A, ecA)) break;
    }
    F3DG_PROD_WAIT_VM0();                                 // the last requested window lands in LDS before the wave leaves
#undef F3DG_PROD_WAIT_VM0
}
