// f3dg_binning.hip -- tile binning: prefix sum, key duplication, stable LSD radix sort, tile ranges.
//
// Replaces, for ALL views of a call at once (reference RAST/cuda_rasterizer/rasterizer_impl.cu):
//   cub::DeviceScan::InclusiveSum      :332      -> scan_* kernels (hand-written reduce / scan / propagate)
//   the blocking D2H of num_rendered   :336      -> count stays on the device (workspace header)
//   duplicateWithKeys                  :70-111   -> duplicate_sorted_kernel
//   cub::DeviceRadixSort::SortPairs    :358-363  -> gsort_* (depth, per Gaussian) and radix2_*_var (tile, per instance) passes of 8 bits
//   cudaMemset(ranges) + identifyTileRanges :365, :149-171 -> memset + group_bounds_kernel
//
// The final list is the reference's: per view, ascending (tile, depth bits, Gaussian id) -- what a stable sort of its 64-bit keys
// (tile << 32 | float_bits(depth)) gives. It is built depth-first (see "depth-first binning" below): the Gaussians of every view are
// sorted by depth once, the instances are generated in that order, and ONE stable pass over the tile bits (two above 256 tiles)
// groups them; no instance is ever sorted by depth. Depths are > 0.2 so their IEEE bits order as unsigned ints.
//
// Wave64 notes: the in-block ranking of the scatter uses 64-lane ballots (one per digit bit) to find, for every
// lane, the set of lanes holding the same digit; ranks are popcounts of that 64-bit mask below the lane. Keys
// are consumed in (wave, round, lane) order which IS memory order, so stability needs no extra bookkeeping.
#include "f3dg_common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

// ------------------------------------------------------------------------------------------------ scan
// Three-kernel scan of n u32 values: per-block sums -> scan of block sums (one workgroup) -> per-block scan.
__global__ void __launch_bounds__(F3DG_BLOCK)
scan_reduce_kernel(const u32* __restrict__ in, u64 n, u32* __restrict__ block_sums)
{
    __shared__ u32 wsum[F3DG_BLOCK / 64];
    const u64 base = (u64)blockIdx.x * F3DG_SCAN_CHUNK;
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < F3DG_SCAN_ITEMS; i++) {
        const u64 k = base + (u64)i * F3DG_BLOCK + threadIdx.x;
        if (k < n) s += in[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Exclusive scan of the block sums in place, by one workgroup looping over them; also publishes the total.
__global__ void __launch_bounds__(1024)
scan_blocksums_kernel(u32* __restrict__ block_sums, u32 nblocks, F3dgHeader* __restrict__ hdr)
{
    __shared__ u32 wtot[16];
    __shared__ u32 carry_s;
    __shared__ u64 carry64_s, wtot64[16];   // the same sum without wrap-around: a batch can hold more than 2^32 instances (V * P * tiles)
    if (threadIdx.x == 0) { carry_s = 0; carry64_s = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 start = 0; start < nblocks; start += 1024) {
        const u32 i = start + threadIdx.x;
        const u32 v = i < nblocks ? block_sums[i] : 0;
        u32 x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wtot[wave] = x;
        unsigned long long v64 = v;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v64 += __shfl_down(v64, off, 64);
        if (lane == 0) wtot64[wave] = v64;
        __syncthreads();
        u32 wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += wtot[w];
        const u32 carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + wave_off + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry_s = carry + wave_off + x;
            u64 t = 0;
            for (int w = 0; w < 16; w++) t += wtot64[w];
            carry64_s += t;
        }
        __syncthreads();
    }
    if (hdr && threadIdx.x == 0) {
        // a wrapped 32-bit total must not pass for a small one: later kernels index the instance arrays with it
        const u64 total = carry64_s;
        hdr->num_rendered = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)total;
        hdr->overflow = total > (u64)hdr->capacity ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(F3DG_BLOCK)
scan_apply_kernel(const u32* in, u32* out /* may alias in */, u64 n, const u32* __restrict__ block_sums,
                  int exclusive)
{
    // each thread owns SCAN_ITEMS CONSECUTIVE values (blocked arrangement) so the scan is a plain running sum
    __shared__ u32 wtot[F3DG_BLOCK / 64];
    const u64 base = (u64)blockIdx.x * F3DG_SCAN_CHUNK + (u64)threadIdx.x * F3DG_SCAN_ITEMS;
    u32 v[F3DG_SCAN_ITEMS];
    u32 s = 0;
    if (base + F3DG_SCAN_ITEMS <= n) {           // 64 contiguous bytes per thread: four 16-byte loads
#pragma unroll
        for (int i = 0; i < F3DG_SCAN_ITEMS / 4; i++) {
            const uint4 w = reinterpret_cast<const uint4*>(in + base)[i];
            v[4 * i] = w.x; v[4 * i + 1] = w.y; v[4 * i + 2] = w.z; v[4 * i + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < F3DG_SCAN_ITEMS; i++) v[i] = (base + i < n) ? in[base + i] : 0;
    }
#pragma unroll
    for (int i = 0; i < F3DG_SCAN_ITEMS; i++) s += v[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 x = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    u32 run = block_sums[blockIdx.x] + x - s;
    for (int w = 0; w < wave; w++) run += wtot[w];
    u32 o[F3DG_SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < F3DG_SCAN_ITEMS; i++) {
        const u32 before = run;
        run += v[i];
        o[i] = exclusive ? before : run;
    }
    if (base + F3DG_SCAN_ITEMS <= n) {
#pragma unroll
        for (int i = 0; i < F3DG_SCAN_ITEMS / 4; i++)
            reinterpret_cast<uint4*>(out + base)[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < F3DG_SCAN_ITEMS; i++)
            if (base + i < n) out[base + i] = o[i];
    }
}

// ================================================================================================ depth-first binning
// Round 1 grouped the instances by tile first and then sorted every (view, tile) group by depth in LDS; that per-tile sort was the
// largest binning kernel (1.0 of 2.3 ms at C2, 4.9 of 12 ms at C5: latency-bound), and it sorted every INSTANCE (R = 3-7 per
// Gaussian). A Gaussian's depth is the same in all its tiles, so the depth order is established once per (view, Gaussian) instead:
//   1. gsort: stable LSD radix sort of each view's P Gaussians by their depth bits (4 passes of 8 bits over V*P keys; culled
//      Gaussians get the key ~0 and produce no instances) -> perm[V][P], ascending (depth, Gaussian id);
//   2. the tile rectangles gathered in that order, prefix sum of their areas, and the instances are GENERATED in (view, depth, id)
//      order, as two streams: group (view << tile_bits | tile, u16 or u32) and Gaussian id;
//   3. stable pass(es) over the tile bits INSIDE every view's segment of the instance arrays: the result is in (view, tile, depth,
//      id) order, i.e. the final list, and a view's instances never leave one XCD's L2;
//   4. identifyTileRanges on the final group stream.
// Same final list as a stable 64-bit sort of (tile << 32 | depth) keys, i.e. the reference's.

// ------------------------------------------------------------------------------------------------ segmented radix passes
// Two-stream (key, payload) stable 8-bit radix pass over SEGMENTS of the arrays (a segment = one view): elements never leave their
// segment, so the arrays stay view-major and a view's data stays in one XCD's L2 (f3dg_xcd_map / the per-XCD work lists below).
// A workgroup takes one chunk of F3DG_SORT_CHUNK elements of one segment. The counts of digit d of chunk c go to
// hist[obase + d * cps + c] (obase = 256 * chunks of all earlier segments), so that ONE exclusive scan of the whole array yields the
// global output positions: every segment owns exactly its own index range of the output.
struct SegChunk {
    u64 seg_base;   // first element of the segment in the key / payload arrays
    u64 obase;      // first entry of the segment's [256][cps] block of the histogram
    u32 n;          // elements of the segment
    u32 cps;        // chunks of the segment
    u32 c;          // the chunk
    u32 seg;        // the segment (fixed segments only)
};

// Variable-length segments (the instances of every view): built on the device by view_segments_kernel
struct SegTable {
    const u32* vstart;      // [V + 1] first instance of every view
    const u32* bglob;       // [V + 1] chunks of all earlier views
    const u32* xprefix;     // [8][xstride] XCD x works on the views x, x + 8, ...: chunks of its earlier views
    u32 xstride;            // V / 8 + 2
    u32 V;
};

// digit of a key in one pass
struct ShiftDigit {
    int shift;
    u32 mask = 255u;        // (a pass over fewer than 8 bits: the tile passes split their bits evenly)
    __device__ __forceinline__ u32 operator()(u32 k) const { return (k >> shift) & mask; }
};
// third and LAST pass of a view's depth sort when all its keys lie within 2^24 of kbase (a multiple of 2^16: the low 16 bits of
// key - kbase are the key's own, which passes 0 and 1 sorted). Keys of Gaussians that touch no tile (~0) land in digit 255; they
// produce no instances, so their place in the order is irrelevant.
struct CompactDigit {
    u32 kbase;
    __device__ __forceinline__ u32 operator()(u32 k) const { const u32 d = (k - kbase) >> 16; return d < 255u ? d : 255u; }
};

template <typename K, typename DIGIT, bool MINMAX = false>
__device__ __forceinline__ void hist_chunk(const K* __restrict__ keys, const SegChunk& ck, const DIGIT digit, u32* __restrict__ hist,
                                           u32 (*h)[256], u32* __restrict__ minmax = nullptr)
{
    u32 kmin = 0xFFFFFFFFu, kmax = 0u;        // MINMAX: range of the keys other than ~0 (Gaussians that touch no tile)
#pragma unroll
    for (int w = 0; w < F3DG_BLOCK / 64; w++) h[w][threadIdx.x] = 0;
    __syncthreads();
    u32* hw = h[threadIdx.x >> 6];
    constexpr int VEC = 16 / (int)sizeof(K);                   // keys per 16-byte load
    typedef K __attribute__((ext_vector_type(VEC))) KV;
    // the chunk's elements [g0, g1) of the array, read as ALIGNED 16-byte vectors (a segment may start anywhere): the first and the
    // last vector are masked. (The few bytes read outside the chunk are inside the array, whose regions are padded to 256 bytes.)
    const u64 base = (u64)ck.c * F3DG_SORT_CHUNK;
    const u64 g0 = ck.seg_base + base;
    const u64 g1 = ck.seg_base + ((u64)ck.n - base < (u64)F3DG_SORT_CHUNK ? (u64)ck.n : base + F3DG_SORT_CHUNK);
    for (u64 a = (g0 & ~(u64)(VEC - 1)) + (u64)threadIdx.x * VEC; a < g1; a += (u64)F3DG_BLOCK * VEC) {
        const KV w = *reinterpret_cast<const KV*>(keys + a);
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            if (a + q >= g0 && a + q < g1) {
                const u32 key = (u32)w[q];
                atomicAdd(&hw[digit(key)], 1u);
                if (MINMAX && key != 0xFFFFFFFFu) { kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax; }
            }
        }
    }
    __shared__ u32 wmm[2][F3DG_BLOCK / 64];
    if (MINMAX) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u32 a = __shfl_down(kmin, off, 64), b = __shfl_down(kmax, off, 64);
            kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
        }
        if ((threadIdx.x & 63) == 0) { wmm[0][threadIdx.x >> 6] = kmin; wmm[1][threadIdx.x >> 6] = kmax; }
    }
    __syncthreads();
    if (MINMAX && threadIdx.x == 0) {       // the chunk's range (kmin > kmax: no key other than ~0); reduced per view by gsort_range_kernel
        u32 a = wmm[0][0], b = wmm[1][0];
#pragma unroll
        for (int w = 1; w < F3DG_BLOCK / 64; w++) { a = wmm[0][w] < a ? wmm[0][w] : a; b = wmm[1][w] > b ? wmm[1][w] : b; }
        minmax[0] = a; minmax[1] = b;
    }
    hist[ck.obase + (u64)threadIdx.x * ck.cps + ck.c] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}

struct ScatterShared {
    u32 cnt[F3DG_BLOCK / 64][256];
    u32 lbase[256];
    u32 gdelta[256];
    u32 wtot[F3DG_BLOCK / 64];
    u32 sval[F3DG_SORT_CHUNK];
};

// What a view's LAST depth pass adds with option sort_fused_rects (off by default: measured equal): the element's payload is a Gaussian id and its final
// position is known, so the pass also fetches that Gaussian's tile rectangle and writes it -- and its area = tiles touched, the input
// of the prefix sum that places the instances -- at the sorted position. This is the one random gather of the path (an 8-byte read
// per 128-byte line, from the view's 8 P bytes in its XCD's L2); inside the pass it flies behind the other chunks' LDS work of the CU
// instead of being a launch of its own that reads the order back.
struct RectSink {
    const uint2* rects;     // the segment's rectangles, id-indexed
    u32* tiles;             // [V P] sorted order
    u32* rx;
    u32* ry;
};

// IOTA: the payload of the input is its position inside the segment (first pass of the depth sort)
template <typename K, bool IOTA, typename DIGIT, bool RECTS = false>
__device__ __forceinline__ void scatter_chunk(const K* __restrict__ keys_in, const u32* __restrict__ vals_in, K* __restrict__ keys_out,
                                              u32* __restrict__ vals_out, const SegChunk& ck, const DIGIT digit,
                                              const u32* __restrict__ offsets /* exclusive scan of hist */, ScatterShared& sh, K* skey,
                                              const RectSink sink = RectSink{nullptr, nullptr, nullptr, nullptr})
{
    // the chunk is digit-sorted inside LDS (stable), then written out in coalesced runs
    const u32 n = ck.n;
    const u64 block_base = (u64)ck.c * F3DG_SORT_CHUNK;
    const u32 in_block = (u32)((n - block_base) < (u64)F3DG_SORT_CHUNK ? (n - block_base) : (u64)F3DG_SORT_CHUNK);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < F3DG_BLOCK / 64; w++) sh.cnt[w][threadIdx.x] = 0;
    __syncthreads();

    const u64 wave_base = block_base + (u64)wave * (64 * F3DG_SORT_ITEMS);
    u32 key[F3DG_SORT_ITEMS];
    u32 val[F3DG_SORT_ITEMS];
    u32 rank[F3DG_SORT_ITEMS];
    const u64 lane_lt = ((u64)1 << lane) - 1;
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {          // all loads first
        const u64 i = wave_base + (u64)r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? (u32)keys_in[ck.seg_base + i] : 0u;
        val[r] = IOTA ? (u32)i : (valid ? vals_in[ck.seg_base + i] : 0u);
    }
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
        const u64 i = wave_base + (u64)r * 64 + lane;
        const bool valid = i < n;
        const u32 d = digit(key[r]);
        // lanes with the same digit: a lane differs from lane L in bit b exactly where ballot(bit b) disagrees with L's own bit
        // (sign-extended over the mask: one v_bfe_i32); the eight mismatch masks are OR-ed (v_or3) and inverted once
        u64 diff = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const u32 own = (u32)((int)(d << (31 - b)) >> 31);
            const u64 bal = __ballot(own != 0u);
            diff |= bal ^ (((u64)own << 32) | own);
        }
        const u64 same = __ballot(valid) & ~diff;
        const u32 below = (u32)__popcll(same & lane_lt);
        const u32 prev = sh.cnt[wave][d];
        rank[r] = prev + below;
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) sh.cnt[wave][d] = prev + (u32)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        const u32 d = threadIdx.x;
        const u32 c0 = sh.cnt[0][d], c1 = sh.cnt[1][d], c2 = sh.cnt[2][d], c3 = sh.cnt[3][d];
        const u32 tot = c0 + c1 + c2 + c3;
        u32 x = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) sh.wtot[wave] = x;
        __syncthreads();
        u32 excl = x - tot;
        for (int w = 0; w < wave; w++) excl += sh.wtot[w];
        sh.lbase[d] = excl;
        sh.gdelta[d] = offsets[ck.obase + (u64)d * ck.cps + ck.c] - excl;
        sh.cnt[0][d] = excl;
        sh.cnt[1][d] = excl + c0;
        sh.cnt[2][d] = excl + c0 + c1;
        sh.cnt[3][d] = excl + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
        const u64 i = wave_base + (u64)r * 64 + lane;
        if (i < n) {
            const u32 d = digit(key[r]);
            const u32 slot = sh.cnt[wave][d] + rank[r];
            skey[slot] = (K)key[r];
            sh.sval[slot] = val[r];
        }
    }
    __syncthreads();
    if constexpr (RECTS) {
        constexpr int GI = 4;                         // gathers in flight per thread
        for (u32 s0 = threadIdx.x; s0 < in_block; s0 += GI * F3DG_BLOCK) {
            u32 pos[GI], id[GI];
            uint2 r[GI];
#pragma unroll
            for (int g = 0; g < GI; g++) {
                const u32 slot = s0 + (u32)g * F3DG_BLOCK;
                const bool ok = slot < in_block;
                id[g] = ok ? sh.sval[slot] : 0u;
                pos[g] = ok ? sh.gdelta[digit((u32)skey[ok ? slot : 0u])] + slot : 0u;
            }
#pragma unroll
            for (int g = 0; g < GI; g++)
                r[g] = s0 + (u32)g * F3DG_BLOCK < in_block ? sink.rects[id[g]] : make_uint2(0u, 0u);
#pragma unroll
            for (int g = 0; g < GI; g++) {
                if (s0 + (u32)g * F3DG_BLOCK < in_block) {
                    vals_out[pos[g]] = id[g];
                    sink.tiles[pos[g]] = (((r[g].x >> 16) & F3DG_RECT_COORD) - (r[g].x & F3DG_RECT_COORD)) *
                                         (((r[g].y >> 16) & F3DG_RECT_COORD) - (r[g].y & F3DG_RECT_COORD));
                    sink.rx[pos[g]] = r[g].x;
                    sink.ry[pos[g]] = r[g].y;
                }
            }
        }
        return;
    }
    for (u32 slot = threadIdx.x; slot < in_block; slot += F3DG_BLOCK) {
        const K k = skey[slot];
        const u32 d = digit((u32)k);
        const u32 pos = sh.gdelta[d] + slot;          // global position (the scan spans all segments)
        if (keys_out) keys_out[pos] = k;
        vals_out[pos] = sh.sval[slot];
    }
}

// ---- equal segments of seg_len elements (the per-view depth sort of the Gaussians): one chunk per workgroup.
// Pass 1 also finds every chunk's key range, reduced to every view's [kmin, kmax] (minmax[2 v], [2 v + 1]) by gsort_range_kernel. A view whose
// range fits 2^24 above kbase = kmin & ~0xFFFF is "compact": its third pass sorts by (key - kbase) >> 16 and is its last, the fourth
// returns at once (its histogram is a stand-in with the right sum so that the scan keeps the other views' positions), and the
// view's order is the OUTPUT OF PASS 2 (see gsort_perm_is_pass2). Depth ranges within a factor of 2-4 -- object-centric cameras --
// are compact; others take the four plain passes.
__device__ __forceinline__ SegChunk fixed_chunk(u32 seg_len, u32 cps)
{
    u32 seg, c;
    f3dg_xcd_map(blockIdx.x, gridDim.x / cps, cps, seg, c);
    SegChunk ck;
    ck.seg_base = (u64)seg * seg_len; ck.obase = (u64)seg * 256u * cps; ck.n = seg_len; ck.cps = cps; ck.c = c; ck.seg = seg;
    return ck;
}

__device__ __forceinline__ bool gsort_compact(const u32* __restrict__ minmax, u32 view, u32& kbase)
{
    const u32 kmin = minmax[2 * view], kmax = minmax[2 * view + 1];
    kbase = kmin & 0xFFFF0000u;
    return kmin <= kmax && ((kmax - kbase) >> 24) == 0u;
}

template <int PASS>
__global__ void __launch_bounds__(F3DG_BLOCK)
gsort_hist_kernel(const u32* __restrict__ keys, u32 seg_len, u32 cps, u32* __restrict__ hist, const u32* __restrict__ minmax,
                  u32* __restrict__ chunk_minmax)
{
    __shared__ u32 h[F3DG_BLOCK / 64][256];
    const SegChunk ck = fixed_chunk(seg_len, cps);
    if constexpr (PASS == 0) {
        hist_chunk<u32, ShiftDigit>(keys, ck, ShiftDigit{0}, hist, h);
    } else if constexpr (PASS == 1) {        // (pass 1 reads the keys from the cache; pass 0 reads them cold)
        hist_chunk<u32, ShiftDigit, true>(keys, ck, ShiftDigit{8}, hist, h, chunk_minmax + 2 * ((size_t)ck.seg * cps + ck.c));
    } else {
        u32 kbase;
        const bool compact = gsort_compact(minmax, ck.seg, kbase);
        if (!compact) {
            hist_chunk<u32, ShiftDigit>(keys, ck, ShiftDigit{8 * PASS}, hist, h);
        } else if constexpr (PASS == 2) {
            hist_chunk<u32, CompactDigit>(keys, ck, CompactDigit{kbase}, hist, h);
        } else {
            const u64 base = (u64)ck.c * F3DG_SORT_CHUNK;
            const u32 in_chunk = (u32)((ck.n - base) < (u64)F3DG_SORT_CHUNK ? (ck.n - base) : (u64)F3DG_SORT_CHUNK);
            hist[ck.obase + (u64)threadIdx.x * ck.cps + ck.c] = threadIdx.x == 0 ? in_chunk : 0u;
        }
    }
}

// exclusive scan of one view's [256][cps] histogram block in place, starting from the view's first output position: the views'
// position ranges are known in advance (seg_len each), so every view is scanned by its own workgroup -- one launch per pass
// instead of the three of the general scan
__global__ void __launch_bounds__(1024)
gsort_scan_kernel(u32 n_per_seg /* 256 cps */, u32 seg_len, u32* __restrict__ hist)
{
    __shared__ u32 wtot[16];
    __shared__ u32 carry_s;
    u32* h = hist + (size_t)blockIdx.x * n_per_seg;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = blockIdx.x * seg_len;
    __syncthreads();
    for (u32 base = 0; base < n_per_seg; base += 4096u) {
        const u32 i = base + 4u * threadIdx.x;                 // n_per_seg is a multiple of 256: whole uint4s
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < n_per_seg) v = *reinterpret_cast<const uint4*>(h + i);
        const u32 tot = v.x + v.y + v.z + v.w;
        u32 x = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off, 64);
            if (lane >= (u32)off) x += y;
        }
        if (lane == 63u) wtot[wave] = x;
        __syncthreads();
        u32 before = carry_s + x - tot;
        for (u32 w = 0; w < wave; w++) before += wtot[w];
        if (i < n_per_seg)
            *reinterpret_cast<uint4*>(h + i) = make_uint4(before, before + v.x, before + v.x + v.y, before + v.x + v.y + v.z);
        __syncthreads();
        if (threadIdx.x == 1023u) carry_s = before + tot;
        __syncthreads();
    }
}

// key range of every view from the ranges of its chunks (one wave per view)
__global__ void __launch_bounds__(64)
gsort_range_kernel(u32 cps, const u32* __restrict__ chunk_minmax, u32* __restrict__ minmax)
{
    u32 kmin = 0xFFFFFFFFu, kmax = 0u;
    for (u32 c = threadIdx.x; c < cps; c += 64) {
        const u32 a = chunk_minmax[2 * ((size_t)blockIdx.x * cps + c)], b = chunk_minmax[2 * ((size_t)blockIdx.x * cps + c) + 1];
        kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u32 a = __shfl_down(kmin, off, 64), b = __shfl_down(kmax, off, 64);
        kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
    }
    if (threadIdx.x == 0) { minmax[2 * blockIdx.x] = kmin; minmax[2 * blockIdx.x + 1] = kmax; }
}

template <int PASS, bool RECTS = false>
__global__ void __launch_bounds__(F3DG_BLOCK)
gsort_scatter_kernel(const u32* __restrict__ keys_in, const u32* __restrict__ vals_in, u32* __restrict__ keys_out,
                     u32* __restrict__ vals_out, u32 seg_len, u32 cps, const u32* __restrict__ offsets, const u32* __restrict__ minmax,
                     RectSink sink)
{
    __shared__ ScatterShared sh;
    __shared__ u32 skey[F3DG_SORT_CHUNK];
    const SegChunk ck = fixed_chunk(seg_len, cps);
    sink.rects += ck.seg_base;
    if constexpr (PASS == 0) {           // the payload of the input is its position inside the segment
        scatter_chunk<u32, true, ShiftDigit>(keys_in, vals_in, keys_out, vals_out, ck, ShiftDigit{0}, offsets, sh, skey);
    } else if constexpr (PASS == 1) {
        scatter_chunk<u32, false, ShiftDigit>(keys_in, vals_in, keys_out, vals_out, ck, ShiftDigit{8}, offsets, sh, skey);
    } else {
        u32 kbase;
        const bool compact = gsort_compact(minmax, ck.seg, kbase);
        // (the keys are not read again after a view's LAST pass -- pass 2 of a compact view, pass 3 otherwise --: it does not write them)
        if (!compact) {
            if constexpr (PASS == 3 && RECTS)
                scatter_chunk<u32, false, ShiftDigit, true>(keys_in, vals_in, (u32*)nullptr, vals_out, ck, ShiftDigit{8 * PASS}, offsets, sh, skey, sink);
            else
                scatter_chunk<u32, false, ShiftDigit>(keys_in, vals_in, PASS == 3 ? (u32*)nullptr : keys_out, vals_out, ck, ShiftDigit{8 * PASS}, offsets, sh, skey);
        } else if constexpr (PASS == 2) {
            if constexpr (RECTS)
                scatter_chunk<u32, false, CompactDigit, true>(keys_in, vals_in, (u32*)nullptr, vals_out, ck, CompactDigit{kbase}, offsets, sh, skey, sink);
            else
                scatter_chunk<u32, false, CompactDigit>(keys_in, vals_in, (u32*)nullptr, vals_out, ck, CompactDigit{kbase}, offsets, sh, skey);
        }
    }
}

// ---- the instances of every view (variable segments): persistent workgroups. Workgroup b belongs to XCD b % 8 and walks that
// XCD's list of (view, chunk) pairs -- views x, x + 8, ... chunk by chunk -- with stride gridDim.x / 8.
// vstart / bglob / xprefix from the prefix sum of tiles_touched in (view, depth) order; all zero chunks on overflow.
__global__ void __launch_bounds__(512)
view_segments_kernel(int V, int P, const u32* __restrict__ offsets_sorted, const F3dgHeader* __restrict__ hdr, u32* __restrict__ vstart,
                     u32* __restrict__ bglob, u32* __restrict__ xprefix, u32 xstride)
{
    __shared__ u32 carry;
    __shared__ u32 wtot[8];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool dead = hdr->overflow != 0;
    auto start_of = [&](u32 v) -> u32 {
        const u64 i = (u64)v * (u64)P;
        return (dead || i == 0) ? 0u : offsets_sorted[i - 1];
    };
    auto chunks_of = [&](u32 v) -> u32 {
        const u32 n = start_of(v + 1) - start_of(v);
        return (n + F3DG_SORT_CHUNK - 1) / F3DG_SORT_CHUNK;
    };
    // global chunk prefix, view order
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 v0 = 0; v0 < (u32)V; v0 += 512) {
        const u32 v = v0 + threadIdx.x;
        const u32 c = v < (u32)V ? chunks_of(v) : 0u;
        u32 x = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off, 64);
            if (lane >= (u32)off) x += y;
        }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        u32 woff = carry;
        for (u32 w = 0; w < wave; w++) woff += wtot[w];
        if (v < (u32)V) { vstart[v] = start_of(v); bglob[v] = woff + x - c; }
        __syncthreads();
        if (threadIdx.x == 511) carry = woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) { vstart[V] = start_of((u32)V); bglob[V] = carry; }
    // per-XCD prefix: wave x scans the chunks of the views x, x + 8, ...
    const u32 nk = ((u32)V > wave) ? ((u32)V - wave + 7u) / 8u : 0u;
    u32 run = 0;
    for (u32 k0 = 0; k0 < nk; k0 += 64) {
        const u32 k = k0 + lane;
        const u32 c = k < nk ? chunks_of(wave + 8u * k) : 0u;
        u32 x = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off, 64);
            if (lane >= (u32)off) x += y;
        }
        if (k < nk) xprefix[wave * xstride + k] = run + x - c;
        run += __shfl(x, 63, 64);
    }
    if (lane == 0) xprefix[wave * xstride + nk] = run;
}

// identifyTileRanges when ONE pass sorted the tile bits: the scanned histogram already holds the first position of every
// (view, tile) group -- hist[256 bglob[v] + digit * chunks_v] -- and the next entry in scan order is where it ends
__global__ void __launch_bounds__(F3DG_BLOCK)
ranges_from_offsets_kernel(u32 V, u32 T, int tile_bits, const u32* __restrict__ bglob, const u32* __restrict__ offsets,
                           const F3dgHeader* __restrict__ hdr, uint2* __restrict__ ranges)
{
    const u32 i = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    if (i >= V * T) return;
    const u32 v = i / T, t = i % T;
    const u32 b0 = bglob[v], cpv = bglob[v + 1] - b0;
    uint2 r = make_uint2(0u, 0u);
    if (cpv != 0u && !hdr->overflow) {
        const u32 d = ((v << tile_bits) | t) & 255u;           // the digit of the group: below 8 tile bits it holds view bits too
        const u32 lo = offsets[256ull * b0 + (u64)d * cpv], hi = offsets[256ull * b0 + (u64)(d + 1u) * cpv];
        if (hi > lo) r = make_uint2(lo, hi);
    }
    ranges[i] = r;
}

// next (view, chunk) of this workgroup's XCD list; false when the list is exhausted. k is the cursor into the XCD's views.
__device__ __forceinline__ bool var_chunk(const SegTable& st, u32 f, u32& k, SegChunk& ck)
{
    const u32 x = blockIdx.x & 7u;
    const u32* xp = st.xprefix + x * st.xstride;
    const u32 nk = (st.V > x) ? (st.V - x + 7u) / 8u : 0u;
    while (k < nk && f >= xp[k + 1]) k++;
    if (k >= nk) return false;
    const u32 v = x + 8u * k;
    const u32 s0 = st.vstart[v], b0 = st.bglob[v];
    ck.seg_base = s0; ck.n = st.vstart[v + 1] - s0; ck.cps = st.bglob[v + 1] - b0; ck.obase = 256ull * b0; ck.c = f - xp[k];
    return true;
}

template <typename K>
__global__ void __launch_bounds__(F3DG_BLOCK)
radix2_hist_var_kernel(const K* __restrict__ keys, SegTable st, int shift, u32 mask, u32* __restrict__ hist)
{
    __shared__ u32 h[F3DG_BLOCK / 64][256];
    u32 k = 0;
    SegChunk ck;
    for (u32 f = blockIdx.x >> 3; var_chunk(st, f, k, ck); f += gridDim.x >> 3)
        hist_chunk<K, ShiftDigit>(keys, ck, ShiftDigit{shift, mask}, hist, h);
}

template <typename K>
__global__ void __launch_bounds__(F3DG_BLOCK)
radix2_scatter_var_kernel(const K* __restrict__ keys_in, const u32* __restrict__ vals_in, K* __restrict__ keys_out,
                          u32* __restrict__ vals_out, SegTable st, int shift, u32 mask, const u32* __restrict__ offsets)
{
    __shared__ ScatterShared sh;
    __shared__ K skey[F3DG_SORT_CHUNK];
    u32 k = 0;
    SegChunk ck;
    for (u32 f = blockIdx.x >> 3; var_chunk(st, f, k, ck); f += gridDim.x >> 3) {
        scatter_chunk<K, false, ShiftDigit>(keys_in, vals_in, keys_out, vals_out, ck, ShiftDigit{shift, mask}, offsets, sh, skey);
        __syncthreads();          // the LDS staging area is reused by the next chunk
    }
}

// tile rectangles (written by the projection kernel: x = rminx | rmaxx << 16, y = rminy | rmaxy << 16) gathered into sorted order,
// with their areas = tiles_touched as the input of the prefix sum that places the instances: the one random gather of the path
__global__ void __launch_bounds__(F3DG_BLOCK)
gsort_gather_rects_kernel(int P, u32* __restrict__ gv0, u32* __restrict__ gv1, const u32* __restrict__ minmax,
                          const uint2* __restrict__ rects, u32* __restrict__ tiles_sorted, u32* __restrict__ rx)
{
    constexpr int ITEMS = 4;                  // four independent gathers in flight per thread
    unsigned view, chunk;                     // a view's rectangles (8 P bytes) are gathered through one XCD's L2
    const unsigned cpv = (unsigned)((P + F3DG_BLOCK * ITEMS - 1) / (F3DG_BLOCK * ITEMS));
    f3dg_xcd_map(blockIdx.x, gridDim.x / cpv, cpv, view, chunk);
    const size_t vb = (size_t)view * P;
    // the view's order is in gv1 after three passes (compact key range) or in gv0 after four; the other one takes ry
    u32 kbase;
    const bool pass2 = gsort_compact(minmax, view, kbase);
    const u32* perm = pass2 ? gv1 : gv0;
    u32* ry = pass2 ? gv0 : gv1;
    const int k0 = (int)(chunk * F3DG_BLOCK * ITEMS + threadIdx.x);
    u32 id[ITEMS];
    uint2 r[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) { const int k = k0 + i * F3DG_BLOCK; id[i] = k < P ? perm[vb + k] : 0u; }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) { const int k = k0 + i * F3DG_BLOCK; r[i] = k < P ? rects[vb + id[i]] : make_uint2(0u, 0u); }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int k = k0 + i * F3DG_BLOCK;
        if (k < P) {
            tiles_sorted[vb + k] = (((r[i].x >> 16) & F3DG_RECT_COORD) - (r[i].x & F3DG_RECT_COORD)) *
                                   (((r[i].y >> 16) & F3DG_RECT_COORD) - (r[i].y & F3DG_RECT_COORD));
            rx[vb + k] = r[i].x;
            ry[vb + k] = r[i].y;
        }
    }
}

// duplicateWithKeys in (view, depth, id) order: sorted position k of view v emits the tiles of Gaussian perm[v][k]. The runs of a
// workgroup's 256 Gaussians are consecutive in the output (a few instances each), so they are assembled in LDS and written out as
// whole lines, DUP_CAP instances per window.
#define F3DG_DUP_CAP 3072
template <typename G>
__global__ void __launch_bounds__(F3DG_BLOCK)
duplicate_sorted_kernel(int P, int tile_bits, int grid_x, const u32* __restrict__ gv0, const u32* __restrict__ gv1,
                        const u32* __restrict__ minmax, const u32* __restrict__ rx, const u32* __restrict__ ry_fused /* null: the view's spare order buffer */,
                        const u32* __restrict__ offsets_sorted,
                        const F3dgHeader* __restrict__ hdr, G* __restrict__ kgrp, u32* __restrict__ vals)
{
    __shared__ G sk[F3DG_DUP_CAP];
    __shared__ u32 sv[F3DG_DUP_CAP];
    if (hdr->overflow) return;
    const int k = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    const int v = blockIdx.y;
    u32 kbase;
    const bool pass2 = gsort_compact(minmax, (u32)v, kbase);      // as in gsort_gather_rects_kernel
    const u32* perm = pass2 ? gv1 : gv0;
    const u32* ry = ry_fused ? ry_fused : pass2 ? gv0 : gv1;
    // output range of the workgroup
    const size_t first = (size_t)v * P + (size_t)blockIdx.x * F3DG_BLOCK;
    const int in_block = min(F3DG_BLOCK, P - (int)(blockIdx.x * F3DG_BLOCK));
    const u32 base = first == 0 ? 0u : offsets_sorted[first - 1];
    const u32 n = offsets_sorted[first + in_block - 1] - base;
    // the workgroup's output range [base, base + n) is assembled in LDS one window of F3DG_DUP_CAP instances at a time and written out
    // as whole lines: a thread emits the part of its Gaussian's run that falls into the window. (Until round 3 a workgroup whose
    // Gaussians cover more than one window wrote its runs directly -- a few bytes per lane, lines apart: with large splats, 24 tiles
    // per Gaussian at sigma0 = 0.05, that was every workgroup and the kernel ran at 1 TB/s.)
    u32 off0 = 0, cnt = 0, x = 0, y = 0, g = 0;
    if (k < P) {
        const size_t pos = (size_t)v * P + k;
        x = rx[pos]; y = ry[pos];
        const u32 rminx = x & F3DG_RECT_COORD, rmaxx = (x >> 16) & F3DG_RECT_COORD, rminy = y & F3DG_RECT_COORD, rmaxy = (y >> 16) & F3DG_RECT_COORD;
        if (rmaxx > rminx && rmaxy > rminy) {
            g = perm[pos];
            off0 = ((pos == 0) ? 0 : offsets_sorted[pos - 1]) - base;
            cnt = (rmaxx - rminx) * (rmaxy - rminy);
        }
    }
    const u32 rminx = x & F3DG_RECT_COORD, rmaxx = (x >> 16) & F3DG_RECT_COORD, rminy = y & F3DG_RECT_COORD, rmaxy = (y >> 16) & F3DG_RECT_COORD;
    const u32 view_base = (u32)v << tile_bits;
    // quadrant mask of an instance (f3dg_common.h: F3DG_ID_BITS): the halves of the first / last tile column and row that the
    // conservative box misses are cleared
    auto halves = [](u32 t, u32 tmin, u32 tmax, u32 word) -> u32 {
        u32 m = 3u;
        if (t == tmin && (word & F3DG_RECT_SKIP_LO)) m &= ~1u;
        if (t + 1u == tmax && (word & F3DG_RECT_SKIP_HI)) m &= ~2u;
        return m;
    };
    for (u32 wb = 0; wb < n; wb += (u32)F3DG_DUP_CAP) {
        const u32 lo = off0 > wb ? off0 : wb;
        const u32 hi = off0 + cnt < wb + (u32)F3DG_DUP_CAP ? off0 + cnt : wb + (u32)F3DG_DUP_CAP;
        if (lo < hi) {
            const u32 w = rmaxx - rminx, j0 = lo - off0;
            u32 ty = rminy + j0 / w, tx = rminx + j0 % w;
            u32 my = halves(ty, rminy, rmaxy, y);
            for (u32 j = lo; j < hi; j++) {
                const u32 mx = halves(tx, rminx, rmaxx, x);
                sk[j - wb] = (G)(view_base | (ty * (u32)grid_x + tx));
                sv[j - wb] = g | ((((my & 1u) ? mx : 0u) | ((my & 2u) ? mx << 2 : 0u)) << F3DG_ID_BITS);
                if (++tx == rmaxx) { tx = rminx; ty++; my = halves(ty, rminy, rmaxy, y); }
            }
        }
        __syncthreads();
        const u32 in_win = n - wb < (u32)F3DG_DUP_CAP ? n - wb : (u32)F3DG_DUP_CAP;
        for (u32 i = threadIdx.x; i < in_win; i += F3DG_BLOCK) {
            kgrp[base + wb + i] = sk[i];
            vals[base + wb + i] = sv[i];
        }
        __syncthreads();
    }
}

// debug export only: the 64-bit sort keys of the reference, (view << tile_bits | tile) << 32 | depth bits, of the final list
__global__ void __launch_bounds__(F3DG_BLOCK)
export_keys_kernel(u32 nseg, int P, int tile_bits, int T, const uint2* __restrict__ ranges, const F3dgHeader* __restrict__ hdr,
                   const u32* __restrict__ point_list, const float* __restrict__ depths, u64* __restrict__ keys_out)
{
    if (hdr->overflow) return;
    for (u32 seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint2 r = ranges[seg];
        const u32 view = seg / (u32)T, tile = seg % (u32)T;
        const u64 hi = (u64)((view << tile_bits) | tile) << 32;
        for (u32 i = r.x + threadIdx.x; i < r.y; i += F3DG_BLOCK)
            keys_out[i] = hi | (u64)__float_as_uint(depths[(size_t)view * P + (point_list[i] & F3DG_ID_MASK)]);
    }
}

// identifyTileRanges: first / one-past-last index of every (view, tile) group of the final list, ranges[view * T + tile] (zeroed before)
template <typename G>
__global__ void __launch_bounds__(F3DG_BLOCK)
group_bounds_kernel(const G* __restrict__ kgrp, const F3dgHeader* __restrict__ hdr, int tile_bits, int T, u32* __restrict__ ranges)
{
    // every thread looks at 8 consecutive entries (one or two 16-byte loads) plus the entry on either side
    const u32 L = hdr->overflow ? 0u : hdr->num_rendered;
    const u32 tmask = (1u << tile_bits) - 1u;
    for (u64 base = ((u64)blockIdx.x * F3DG_BLOCK + threadIdx.x) * 8u; base < L; base += (u64)gridDim.x * F3DG_BLOCK * 8u) {
        G v[8];
        if (base + 8 <= L) {
            typedef G __attribute__((ext_vector_type(8))) G8;
            const G8 w = *reinterpret_cast<const G8*>(kgrp + base);          // base is a multiple of 8: aligned
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = w[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = base + i < L ? kgrp[base + i] : (G)0;
        }
        u32 prev = base > 0 ? (u32)kgrp[base - 1] : 0u;
        const u32 next = base + 8 < L ? (u32)kgrp[base + 8] : 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 idx = base + i;
            if (idx < L) {
                const u32 cur = (u32)v[i];
                const u32 nxt = i < 7 ? (u32)v[i + 1] : next;
                const u32 seg = (cur >> tile_bits) * (u32)T + (cur & tmask);
                if (idx == 0 || prev != cur) ranges[2u * seg] = (u32)idx;
                if (idx == L - 1 || nxt != cur) ranges[2u * seg + 1u] = (u32)idx + 1u;
                prev = cur;
            }
        }
    }
}

int bits_for(unsigned long long n_values)   // number of bits needed to represent values 0 .. n_values-1
{
    int b = 0;
    while (b < 64 && (n_values - 1) >> b) b++;
    return n_values <= 1 ? 0 : b;
}

} // namespace

int f3dg_launch_scan_inclusive(hipStream_t s, const unsigned* in, unsigned* out, unsigned long long n,
                               unsigned* tmp, unsigned tmp_elems, int exclusive, F3dgHeader* hdr_total)
{
    if (n == 0) return F3DG_OK;
    const unsigned nblocks = (unsigned)((n + F3DG_SCAN_CHUNK - 1) / F3DG_SCAN_CHUNK);
    if (nblocks > tmp_elems) return F3DG_ERR_WORKSPACE;
    F3DG_KLAUNCH(scan_reduce_kernel, dim3(nblocks), dim3(F3DG_BLOCK), 0, s, in, (u64)n, tmp);
    F3DG_KLAUNCH(scan_blocksums_kernel, dim3(1), dim3(1024), 0, s, tmp, nblocks, hdr_total);
    F3DG_KLAUNCH(scan_apply_kernel, dim3(nblocks), dim3(F3DG_BLOCK), 0, s, in, out, (u64)n, tmp, exclusive);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

int f3dg_tile_bits(int T) { return bits_for((unsigned long long)T); }

// Number of 8-bit passes over the instances: only the tile bits (instances are generated in (view, depth, id) order).
int g_f3dg_tile_split = 1;         // lab option tile_split: two tile passes split their bits evenly (1, default) or 8 + rest (0)

int f3dg_sort_passes(int V, int T)
{
    (void)V;
    const int bits = f3dg_tile_bits(T);
    return bits == 0 ? 0 : (bits + 7) / 8;
}

template <typename G>
static int binning_tail(hipStream_t s, int V, int P, int grid_x, int T, int tile_bits, const F3dgLayout& L, char* ws, F3dgHeader* hdr)
{
    int rc = F3DG_OK;
    const size_t VP = (size_t)V * P;
    u32* scan_tmp = reinterpret_cast<u32*>(ws + L.scan_tmp);
    u32* hist = reinterpret_cast<u32*>(ws + L.hist);
    u32* gk[2] = { reinterpret_cast<u32*>(ws + L.gsort), reinterpret_cast<u32*>(ws + L.gsort) + VP };
    u32* gv[2] = { reinterpret_cast<u32*>(ws + L.gsort) + 2 * VP, reinterpret_cast<u32*>(ws + L.gsort) + 3 * VP };
    G* kgrp[2] = { reinterpret_cast<G*>(ws + L.keys[0]), reinterpret_cast<G*>(ws + L.keys[1]) };   // (view << tile_bits | tile) per instance
    u32* vals[2] = { reinterpret_cast<u32*>(ws + L.vals[0]), reinterpret_cast<u32*>(ws + L.vals[1]) };
    const dim3 pgrid((P + F3DG_BLOCK - 1) / F3DG_BLOCK, V);

    // 1. per-view stable sort of the Gaussians by their sort keys (written by the projection kernel into gk[0]: the depth bits,
    //    ~0 for Gaussians that touch no tile); the first pass takes the Gaussian id from the position
    const u32 cps = (u32)((P + F3DG_SORT_CHUNK - 1) / F3DG_SORT_CHUNK);
    const u32 gblocks = (u32)V * cps;
    u32* minmax = reinterpret_cast<u32*>(ws + L.segtab) + L.segtab_minmax;
    u32* chunk_minmax = minmax + 2 * (size_t)V;
    const bool view_scan = cps <= 64;      // one workgroup per view (<= 4 rounds of 4096 entries) or the general three-kernel scan
    // (sort_fused_rects: the last pass of a view also delivers its rectangles and tile counts in sorted order; see RectSink)
    const bool fused = g_f3dg_sort_fused_rects != 0;
    u32* const gx = reinterpret_cast<u32*>(ws + L.gsort) + 4 * VP;           // [3][V P]: tiles / prefix sum, rx, ry of the fused path
    const RectSink sink{reinterpret_cast<const uint2*>(ws + L.rects), gx, gx + VP, gx + 2 * VP};
#define F3DG_GSORT_PASS(PASS, IN, OUT)                                                                                                     \
    F3DG_KLAUNCH((gsort_hist_kernel<PASS>), dim3(gblocks), dim3(F3DG_BLOCK), 0, s, gk[IN], (u32)P, cps, hist, minmax, chunk_minmax); \
    if (view_scan)                                                                                                                         \
        F3DG_KLAUNCH(gsort_scan_kernel, dim3(V), dim3(1024), 0, s, 256u * cps, (u32)P, hist);                                       \
    else {                                                                                                                                 \
        rc = f3dg_launch_scan_inclusive(s, hist, hist, (unsigned long long)256 * gblocks, scan_tmp, L.scan_tmp_elems, 1, nullptr);        \
        if (rc != F3DG_OK) return rc;                                                                                                      \
    }                                                                                                                                      \
    if (fused && PASS >= 2)                                                                                                                \
        F3DG_KLAUNCH((gsort_scatter_kernel<PASS, (PASS >= 2)>), dim3(gblocks), dim3(F3DG_BLOCK), 0, s, gk[IN], gv[IN], gk[OUT], gv[OUT], (u32)P, cps, \
                     hist, minmax, sink);                                                                                                  \
    else                                                                                                                                   \
        F3DG_KLAUNCH((gsort_scatter_kernel<PASS>), dim3(gblocks), dim3(F3DG_BLOCK), 0, s, gk[IN], gv[IN], gk[OUT], gv[OUT], (u32)P, cps, \
                     hist, minmax, sink)
    F3DG_GSORT_PASS(0, 0, 1);
    F3DG_GSORT_PASS(1, 1, 0);
    F3DG_KLAUNCH(gsort_range_kernel, dim3(V), dim3(64), 0, s, cps, chunk_minmax, minmax);
    F3DG_GSORT_PASS(2, 0, 1);
    F3DG_GSORT_PASS(3, 1, 0);
#undef F3DG_GSORT_PASS
    // a view's order (perm) is gv[1] after pass 2 if its key range is compact, gv[0] after pass 3 otherwise; the other one takes ry
    u32* offsets_sorted = fused ? sink.tiles : gk[1];          // tiles_touched in sorted order, then its inclusive prefix sum (in place)
    u32* rx = fused ? sink.rx : gk[0];

    // 2. instances in (view, depth, id) order
    if (!fused)
        F3DG_KLAUNCH(gsort_gather_rects_kernel, dim3((unsigned)V * (unsigned)((P + 4 * F3DG_BLOCK - 1) / (4 * F3DG_BLOCK))), dim3(F3DG_BLOCK), 0, s, P, gv[0], gv[1], minmax,
                           reinterpret_cast<const uint2*>(ws + L.rects), offsets_sorted, rx);
    rc = f3dg_launch_scan_inclusive(s, offsets_sorted, offsets_sorted, (unsigned long long)VP, scan_tmp, L.scan_tmp_elems, 0, hdr);
    if (rc != F3DG_OK) return rc;
    const int passes = f3dg_sort_passes(V, T);
    int src = passes & 1;                                                      // so that the tile pass(es) end in half 0
    F3DG_KLAUNCH((duplicate_sorted_kernel<G>), pgrid, dim3(F3DG_BLOCK), 0, s, P, tile_bits, grid_x, gv[0], gv[1], minmax, rx, fused ? sink.ry : (u32*)nullptr, offsets_sorted, hdr,
                       kgrp[src], vals[src]);

    // 3. stable pass(es) over the tile bits inside every view's segment of the instance arrays: (view, tile, depth, id) order
    if (passes > 0) {
        SegTable st;
        u32* segtab = reinterpret_cast<u32*>(ws + L.segtab);
        const u32 xstride = (u32)V / 8u + 2u;
        st.vstart = segtab; st.bglob = segtab + (V + 1); st.xprefix = segtab + 2 * (V + 1); st.xstride = xstride; st.V = (u32)V;
        F3DG_KLAUNCH(view_segments_kernel, dim3(1), dim3(512), 0, s, V, P, offsets_sorted, hdr, segtab, segtab + (V + 1),
                           segtab + 2 * (V + 1), xstride);
        const size_t nbmax = (size_t)L.sort_blocks + (size_t)V;                // >= sum over the views of ceil(instances / chunk)
        const u32 per_xcd = (u32)((nbmax + 7) / 8 < 160 ? (nbmax + 7) / 8 : 160);   // 5 workgroups per CU of the scatter's LDS
        const u32 per_xcd_h = (u32)((nbmax + 7) / 8 < 512 ? (nbmax + 7) / 8 : 512);
        // two passes (257..65,536 tiles: 512^2 images have 1,024): the tile bits are split EVENLY, 5 + 5 instead of 8 + 2 -- a pass costs
        // the same whatever its digit width, and a chunk's runs of equal digits, which leave LDS as coalesced stores, are 8 x longer
        // with 32 digits than with 256 (option tile_split 0: the 8-bit digits of rounds 1-5)
        const int split = (passes == 2 && g_f3dg_tile_split) ? (tile_bits + 1) / 2 : 8;
        for (int p = 0; p < passes; p++) {
            const int shift = split * p;
            const u32 mask = (passes == 2 && g_f3dg_tile_split) ? (1u << (p == 0 ? split : tile_bits - split)) - 1u : 255u;
            F3DG_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(u32) * 256 * nbmax, s));
            F3DG_KLAUNCH((radix2_hist_var_kernel<G>), dim3(8 * per_xcd_h), dim3(F3DG_BLOCK), 0, s, kgrp[src], st, shift, mask, hist);
            rc = f3dg_launch_scan_inclusive(s, hist, hist, (unsigned long long)256 * nbmax, scan_tmp, L.scan_tmp_elems, 1, nullptr);
            if (rc != F3DG_OK) return rc;
            // (a single pass: the group stream is not needed again, the ranges come from the scanned histogram)
            F3DG_KLAUNCH((radix2_scatter_var_kernel<G>), dim3(8 * per_xcd), dim3(F3DG_BLOCK), 0, s, kgrp[src], vals[src],
                               passes == 1 ? (G*)nullptr : kgrp[src ^ 1], vals[src ^ 1], st, shift, mask, hist);
            src ^= 1;
        }
        if (passes == 1) {
            const u32 nseg = (u32)V * (u32)T;
            F3DG_KLAUNCH(ranges_from_offsets_kernel, dim3((nseg + F3DG_BLOCK - 1) / F3DG_BLOCK), dim3(F3DG_BLOCK), 0, s, (u32)V, (u32)T, tile_bits,
                               segtab + (V + 1), hist, hdr, reinterpret_cast<uint2*>(ws + L.ranges));
            F3DG_HIP_CHECK(hipGetLastError());
            return F3DG_OK;
        }
    }
    // 4. identifyTileRanges on the final list
    F3DG_HIP_CHECK(hipMemsetAsync(ws + L.ranges, 0, sizeof(uint2) * (size_t)V * T, s));
    F3DG_KLAUNCH((group_bounds_kernel<G>), dim3(2048), dim3(F3DG_BLOCK), 0, s, kgrp[0], hdr, tile_bits, T, reinterpret_cast<u32*>(ws + L.ranges));
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

int f3dg_launch_binning(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws, int export_offsets)
{
    const int grid_x = (W + F3DG_TILE - 1) / F3DG_TILE, grid_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = grid_x * grid_y;
    const int tile_bits = f3dg_tile_bits(T);
    F3dgHeader* hdr = reinterpret_cast<F3dgHeader*>(ws + L.header);

    int rc = F3DG_OK;
    if (export_offsets) {
        // the (view, Gaussian)-ordered prefix sum of the reference (point_offsets) is only an exported intermediate here
        rc = f3dg_launch_scan_inclusive(s, reinterpret_cast<const u32*>(ws + L.tiles), reinterpret_cast<u32*>(ws + L.offsets),
                                        (unsigned long long)V * P, reinterpret_cast<u32*>(ws + L.scan_tmp), L.scan_tmp_elems, 0, nullptr);
        if (rc != F3DG_OK) return rc;
    }
    // group stream type that fits (view << tile_bits | tile)
    const bool small = !g_f3dg_sort_wide_groups &&
                       (((unsigned long long)(V > 0 ? V - 1 : 0) << tile_bits) | ((1ull << tile_bits) - 1ull)) <= 0xFFFFull;
    rc = small ? binning_tail<unsigned short>(s, V, P, grid_x, T, tile_bits, L, ws, hdr)
               : binning_tail<u32>(s, V, P, grid_x, T, tile_bits, L, ws, hdr);
    if (rc != F3DG_OK) return rc;
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// debug export: rebuild the reference's 64-bit sort keys of the final list into the workspace's key region (half 0)
int f3dg_launch_export_keys(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws)
{
    const int grid_x = (W + F3DG_TILE - 1) / F3DG_TILE, grid_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = grid_x * grid_y;
    const u32 nseg = (u32)V * (u32)T;
    const u32 rg = nseg < 65535u * 4u ? nseg : 65535u * 4u;
    F3DG_KLAUNCH(export_keys_kernel, dim3(rg), dim3(F3DG_BLOCK), 0, s, nseg, P, f3dg_tile_bits(T), T,
                       reinterpret_cast<const uint2*>(ws + L.ranges), reinterpret_cast<const F3dgHeader*>(ws + L.header),
                       reinterpret_cast<const u32*>(ws + L.vals[0]), reinterpret_cast<const float*>(ws + L.depths),
                       reinterpret_cast<u64*>(ws + L.keys[0]));
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
