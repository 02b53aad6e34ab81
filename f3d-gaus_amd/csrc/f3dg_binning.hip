// f3dg_binning.hip -- tile binning: prefix sum, key duplication, stable LSD radix sort, tile ranges.
//
// Replaces, for ALL views of a call at once (reference RAST/cuda_rasterizer/rasterizer_impl.cu):
//   cub::DeviceScan::InclusiveSum      :332      -> scan_* kernels (hand-written reduce / scan / propagate)
//   the blocking D2H of num_rendered   :336      -> count stays on the device (workspace header)
//   duplicateWithKeys                  :70-111   -> duplicate_keys_kernel
//   cub::DeviceRadixSort::SortPairs    :358-363  -> radix_hist_kernel + scan + radix_scatter_kernel (8 bits/pass)
//   cudaMemset(ranges) + identifyTileRanges :365, :149-171 -> group_bounds / group_counts / scan / group_ranges kernels
//
// Sort keys are ((view << tile_bits) | tile) << 32 | float_bits(depth), carried as separate streams until the final
// buffer (see duplicate_keys_kernel). Instances are GENERATED in (view, Gaussian) order,
// so only the tile bits need a global sort: after the stable tile-digit pass(es) every (tile, view) group is contiguous
// and internally in ascending Gaussian-id order; group bounds + a scan over the V*T groups in (view, tile) order give
// each group its final place, and the per-group depth sort (tile_sort_kernel) reads the group where the tile pass left
// it and writes it, depth-sorted, where the compositing kernel expects it. One global pass at 256^2 (T = 256) for any
// number of views, instead of the six a flat 47-bit sort needs. Within a view the final order is exactly the
// reference's (tile, depth, then ascending Gaussian id). Depths are > 0.2 so their IEEE bits order as unsigned ints.
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

// ------------------------------------------------------------------------------------------------ keys
// The unsorted and the tile-grouped instance buffers are three streams (structure of arrays), not u64 keys:
//   depth [cap] u32 : the IEEE bits of the view-space depth (only read again by the per-tile sort)
//   grp   [cap] G   : (view << tile_bits) | tile -- the only thing the histogram / group-bounds kernels read. G is u16
//                     whenever V << tile_bits fits (C2: 120 views x 256 tiles), so those kernels move 2 B instead of 8 B
//   val   [cap] u32 : Gaussian id
// depth and grp share the 8-byte-per-instance region that held the u64 keys. Only the FINAL, depth-sorted buffer (half 0)
// keeps u64 keys ((grp << 32) | depth), which is what the debug export and the tests read.
template <typename G>
__global__ void __launch_bounds__(F3DG_BLOCK)
duplicate_keys_kernel(int P, int tile_bits, int grid_x, int grid_y, const float2* __restrict__ means2D,
                      const float* __restrict__ depths, const u32* __restrict__ offsets,
                      const int* __restrict__ radii, const F3dgHeader* __restrict__ hdr,
                      u32* __restrict__ kdepth, G* __restrict__ kgrp, u32* __restrict__ vals)
{
    if (hdr->overflow) return;
    const int g = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    const int v = blockIdx.y;
    if (g >= P) return;
    const size_t idx = (size_t)v * P + g;
    const int radius = radii[idx];
    if (radius > 0) {
        u32 off = (idx == 0) ? 0 : offsets[idx - 1];
        const float2 p = means2D[idx];
        const int rminx = min(grid_x, max(0, (int)((p.x - radius) / F3DG_TILE)));
        const int rminy = min(grid_y, max(0, (int)((p.y - radius) / F3DG_TILE)));
        const int rmaxx = min(grid_x, max(0, (int)((p.x + radius + F3DG_TILE - 1) / F3DG_TILE)));
        const int rmaxy = min(grid_y, max(0, (int)((p.y + radius + F3DG_TILE - 1) / F3DG_TILE)));
        const u32 depth_bits = __float_as_uint(depths[idx]);
        const u32 view_base = (u32)v << tile_bits;
        for (int y = rminy; y < rmaxy; y++)
            for (int x = rminx; x < rmaxx; x++) {
                kdepth[off] = depth_bits;
                kgrp[off] = (G)(view_base | (u32)(y * grid_x + x));
                vals[off] = (u32)g;
                off++;
            }
    }
}

// ------------------------------------------------------------------------------------------------ sort
template <typename G>
__global__ void __launch_bounds__(F3DG_BLOCK)
radix_hist_kernel(const G* __restrict__ kgrp, const F3dgHeader* __restrict__ hdr, int shift, u32 nblocks,
                  u32* __restrict__ hist /* [256][nblocks] */)
{
    // per-wave counters (4x fewer same-address LDS conflicts), 8 consecutive entries per 16/32-byte load
    __shared__ u32 h[F3DG_BLOCK / 64][256];
#pragma unroll
    for (int w = 0; w < F3DG_BLOCK / 64; w++) h[w][threadIdx.x] = 0;
    __syncthreads();
    const u32 n = hdr->overflow ? 0u : hdr->num_rendered;
    const u64 base = (u64)blockIdx.x * F3DG_SORT_CHUNK;
    u32* hw = h[threadIdx.x >> 6];
    if (base < n) {
        typedef G __attribute__((ext_vector_type(8))) G8;
#pragma unroll
        for (int i = 0; i < F3DG_SORT_ITEMS / 8; i++) {
            const u64 k = base + ((u64)i * F3DG_BLOCK + threadIdx.x) * 8u;
            if (k + 8 <= n) {
                const G8 w = *reinterpret_cast<const G8*>(kgrp + k);
#pragma unroll
                for (int q = 0; q < 8; q++) atomicAdd(&hw[((u32)w[q] >> shift) & 255u], 1u);
            } else {
                for (int q = 0; q < 8; q++)
                    if (k + q < n) atomicAdd(&hw[((u32)kgrp[k + q] >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
    }
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}

template <typename G>
__global__ void __launch_bounds__(F3DG_BLOCK)
radix_scatter_kernel(const u32* __restrict__ kdepth_in, const G* __restrict__ kgrp_in, const u32* __restrict__ vals_in,
                     u32* __restrict__ kdepth_out, G* __restrict__ kgrp_out, u32* __restrict__ vals_out,
                     const F3dgHeader* __restrict__ hdr, int shift, u32 nblocks,
                     const u32* __restrict__ offsets /* exclusive scan of hist, [256][nblocks] */)
{
    // The chunk is first sorted by digit INSIDE LDS (stable), then written out: consecutive LDS slots of one digit
    // go to consecutive global addresses, so the global stores are coalesced runs instead of 4096 scattered 8+4-byte
    // writes (the first version measured 2.7x write amplification on the WRITE_SIZE counter).
    __shared__ u32 cnt[F3DG_BLOCK / 64][256];
    __shared__ u32 lbase[256];          // first LDS slot of each digit inside the chunk
    __shared__ u32 gdelta[256];         // global position of a digit's first element minus lbase
    __shared__ u32 wtot[F3DG_BLOCK / 64];
    __shared__ u32 sdep[F3DG_SORT_CHUNK];
    __shared__ G sgrp[F3DG_SORT_CHUNK];
    __shared__ u32 sval[F3DG_SORT_CHUNK];
    const u32 n = hdr->overflow ? 0u : hdr->num_rendered;
    const u64 block_base = (u64)blockIdx.x * F3DG_SORT_CHUNK;
    if (block_base >= n) return;
    const u32 in_block = (u32)((n - block_base) < (u64)F3DG_SORT_CHUNK ? (n - block_base) : (u64)F3DG_SORT_CHUNK);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < F3DG_BLOCK / 64; w++) cnt[w][threadIdx.x] = 0;
    __syncthreads();

    const u64 wave_base = block_base + (u64)wave * (64 * F3DG_SORT_ITEMS);
    u32 dep[F3DG_SORT_ITEMS];
    u32 grp[F3DG_SORT_ITEMS];
    u32 val[F3DG_SORT_ITEMS];
    u32 rank[F3DG_SORT_ITEMS];
    const u64 lane_lt = ((u64)1 << lane) - 1;
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {          // all loads first
        const u64 i = wave_base + (u64)r * 64 + lane;
        const bool valid = i < n;
        dep[r] = valid ? kdepth_in[i] : 0u;
        grp[r] = valid ? (u32)kgrp_in[i] : 0u;
        val[r] = valid ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
        const u64 i = wave_base + (u64)r * 64 + lane;
        const bool valid = i < n;
        const u32 d = (grp[r] >> shift) & 255u;
        u64 same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const u32 below = (u32)__popcll(same & lane_lt);
        const u32 prev = cnt[wave][d];
        rank[r] = prev + below;
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) cnt[wave][d] = prev + (u32)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        // thread d: chunk-wide count of digit d, exclusive scan over the 256 digits -> lbase; per-wave starts -> cnt
        const u32 d = threadIdx.x;
        const u32 c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        const u32 tot = c0 + c1 + c2 + c3;
        u32 x = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        u32 excl = x - tot;
        for (int w = 0; w < wave; w++) excl += wtot[w];
        lbase[d] = excl;
        gdelta[d] = offsets[(size_t)d * nblocks + blockIdx.x] - excl;
        cnt[0][d] = excl;
        cnt[1][d] = excl + c0;
        cnt[2][d] = excl + c0 + c1;
        cnt[3][d] = excl + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
        const u64 i = wave_base + (u64)r * 64 + lane;
        if (i < n) {
            const u32 d = (grp[r] >> shift) & 255u;
            const u32 slot = cnt[wave][d] + rank[r];
            sdep[slot] = dep[r];
            sgrp[slot] = (G)grp[r];
            sval[slot] = val[r];
        }
    }
    __syncthreads();
    for (u32 slot = threadIdx.x; slot < in_block; slot += F3DG_BLOCK) {
        const G k = sgrp[slot];
        const u32 d = ((u32)k >> shift) & 255u;
        const u32 pos = gdelta[d] + slot;
        kdepth_out[pos] = sdep[slot];
        kgrp_out[pos] = k;
        vals_out[pos] = sval[slot];
    }
}

// ------------------------------------------------------------------------------------------------ per-tile depth sort
// Level 2 of the two-level sort. After the global pass(es) have grouped the instances by (view, tile) -- stably, so each
// segment is in ascending Gaussian-id order -- one workgroup per segment sorts it by its depth bits, in three tiers:
//   * n <= 4032  (C2: ~2.5 k entries per tile)   tile_sort_lds_kernel<512, 8>, 32 KB of LDS, 4 workgroups = 32 waves per CU
//   * n <= 16320 (C5: 8-9 k entries per tile)     tile_sort_lds_kernel<1024, 16>, 128 KB of LDS, one 16-wave workgroup per CU
//   * longer                                        tile_sort_long_kernel: 8-bit LSD passes through a global scratch slice
// The LDS tiers first OR (key ^ first key) over the segment: only the depth bits that actually vary inside the tile are
// sorted, in ceil(bits/9) passes of <= 9 bits (three passes for the 24 varying bits of depths in [6.7, 8.7], not four);
// 12 B read + 12 B written per instance of global traffic.

// stable in-wave ranking of one digit per lane; returns the lane's rank among equal digits seen so far by this wave
template <typename CT>
__device__ __forceinline__ u32 wave_rank(u32 d, bool valid, CT* wave_cnt, u64 lane_lt, int digit_bits = 8)
{
    u64 same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 9; b++) {
        if (b < digit_bits) {                 // wave-uniform
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
    }
    const u32 below = (u32)__popcll(same & lane_lt);
    const u32 prev = wave_cnt[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) wave_cnt[d] = (CT)(prev + (u32)__popcll(same));
    __builtin_amdgcn_wave_barrier();
    return prev + below;
}

// (view, tile) groups of the tile-sorted buffer: first / one-past-last index of every group, indexed view * T + tile
template <typename G>
__global__ void __launch_bounds__(F3DG_BLOCK)
group_bounds_kernel(const G* __restrict__ kgrp, const F3dgHeader* __restrict__ hdr, int tile_bits, int T,
                    u32* __restrict__ gstart, u32* __restrict__ gend)
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
                if (idx == 0 || prev != cur) gstart[seg] = (u32)idx;
                if (idx == L - 1 || nxt != cur) gend[seg] = (u32)idx + 1u;
                prev = cur;
            }
        }
    }
}

__global__ void __launch_bounds__(F3DG_BLOCK)
group_counts_kernel(u32 nseg, const u32* __restrict__ gstart, const u32* __restrict__ gend, u32* __restrict__ gcount)
{
    const u32 i = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    if (i < nseg) gcount[i] = gend[i] - gstart[i];
}

__global__ void __launch_bounds__(F3DG_BLOCK)
group_ranges_kernel(u32 nseg, const u32* __restrict__ gcount, const u32* __restrict__ gcum /* inclusive scan */,
                    uint2* __restrict__ ranges, F3dgHeader* __restrict__ hdr, u32 small_cap, u32 mid_cap)
{
    const u32 i = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    if (i < nseg) {
        const u32 c = gcount[i];
        ranges[i] = c ? make_uint2(gcum[i] - c, gcum[i]) : make_uint2(0u, 0u);
        if (c > mid_cap) atomicAdd(&hdr->n_long_segments, 1u);
        else if (c > small_cap) atomicAdd(&hdr->n_mid_segments, 1u);
    }
}

// LDS-resident per-(view, tile) sort for segments of lo < n <= THREADS * ITEMS - 64 entries. The keys live in REGISTERS
// between the passes (ITEMS per thread, in (wave, row, lane) = segment order); one LDS buffer is only the exchange medium of
// a pass (scatter to the ranked slot, barrier, read the own rows back). The payload is the position inside the segment
// (u16; the Gaussian ids are gathered once, at the end). Two instantiations:
//   <512, 8>: n <= 4032, 32 KB of LDS, 64 VGPRs -> 4 workgroups = 32 waves per CU (the C2 regime: ~2.5 k entries per tile)
//   <1024, 16>: n <= 16320, 128 KB of LDS -> 1 workgroup of 16 waves per CU (the C5 regime: 8-9 k entries per tile;
//               measured 1.7 -> 0.9 ms against <512, 32>, which kept only 8 waves per CU)
#ifndef F3DG_SMALL_THREADS
#define F3DG_SMALL_THREADS 512
#define F3DG_SMALL_ITEMS 8
#endif
#ifndef F3DG_MID_THREADS
#define F3DG_MID_THREADS 1024
#define F3DG_MID_ITEMS 16
#endif
template <int THREADS, int ITEMS>
__global__ void __launch_bounds__(THREADS, ITEMS == 8 ? 8 : 4)
tile_sort_lds_kernel(const uint2* __restrict__ ranges, const u32* __restrict__ gstart, u32 n_segments,
                     const F3dgHeader* __restrict__ hdr, u32 n_lo /* exclusive */, int tile_bits, int T,
                     const u32* __restrict__ kdepth_src, const u32* __restrict__ vals_src,   // tile-grouped streams
                     u64* __restrict__ keys_dst /* may be null: the sorted keys are only kept for inspection */,
                     u32* __restrict__ vals_dst)                                             // final buffer (half 0)
{
    constexpr int WAVES = THREADS / 64;
    constexpr u32 CAP = (u32)THREADS * ITEMS - 64u;
    constexpr int DPT = THREADS >= 512 ? 1 : 512 / THREADS;   // digits of the 512-entry counter table owned by one thread
    const bool owner = (u32)threadIdx.x * DPT < 512u;          // with 1024 threads only the first 512 own a digit
    __shared__ unsigned short cnt[WAVES][512];          // per-wave digit counters (up to 9-bit digits); values <= CAP < 65536
    __shared__ u32 wtot[WAVES];
    __shared__ u32 sdepth[CAP];
    __shared__ unsigned short sidx[CAP];
    if (hdr->overflow) return;
    if (n_lo != 0u && hdr->n_mid_segments == 0u) return;          // the medium tier is launched unconditionally
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 lane_lt = ((u64)1 << lane) - 1;

    for (u32 seg = blockIdx.x; seg < n_segments; seg += gridDim.x) {
        const uint2 range = ranges[seg];
        const u32 n = range.y - range.x;
        if (n <= n_lo || n > CAP) continue;
        const u32 src0 = gstart[seg];
        __syncthreads();

        const u32 first = kdepth_src[src0];
        const u64 hi = (u64)(((seg / (u32)T) << tile_bits) | (seg % (u32)T)) << 32;   // (view, tile) bits of the final key
        const u32 wave_base = (u32)wave * (64 * ITEMS);
        u32 dk[ITEMS];                                              // depth bits
        u32 di[ITEMS];                                              // position in the segment (low 16) | rank << 16
        u32 diff = 0;
        // all of this thread's global loads are issued before the first one is consumed
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const u32 i = wave_base + (u32)r * 64 + lane;
            dk[r] = i < n ? kdepth_src[src0 + i] : 0xFFFFFFFFu;
            di[r] = i;
        }
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const u32 i = wave_base + (u32)r * 64 + lane;
            if (i < n) diff |= dk[r] ^ first;
        }
        // number of depth bits that actually vary inside this tile -> as few, as narrow (<= 9 bit) passes as possible
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) diff |= __shfl_xor(diff, m, 64);
        if (lane == 0) wtot[wave] = diff;
        __syncthreads();
        diff = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) diff |= wtot[w];
        const int vbits = diff ? 32 - __builtin_clz(diff) : 0;
        const int npass = (vbits + 8) / 9;
        const int dbits = npass ? (vbits + npass - 1) / npass : 0;   // <= 9
        const u32 dmask = (1u << dbits) - 1u;
        for (int pass = 0; pass < npass; pass++) {
            const int shift = dbits * pass;
            __syncthreads();
            if (owner) {
#pragma unroll
                for (int w = 0; w < WAVES; w++)
#pragma unroll
                    for (int q = 0; q < DPT; q++) cnt[w][DPT * threadIdx.x + q] = 0;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if (wave_base + (u32)r * 64 < n) {           // wave-uniform
                    const bool valid = wave_base + (u32)r * 64 + lane < n;
                    const u32 rk = wave_rank((dk[r] >> shift) & dmask, valid, cnt[wave], lane_lt, dbits);
                    di[r] = (di[r] & 0xFFFFu) | (rk << 16);
                }
            }
            __syncthreads();
            {
                // thread t owns the DPT consecutive digits DPT*t ..: chunk-wide exclusive scan over the (up to) 512 digits
                u32 tot[DPT], x = 0;
#pragma unroll
                for (int q = 0; q < DPT; q++) {
                    tot[q] = 0;
                    if (owner) {
#pragma unroll
                        for (int w = 0; w < WAVES; w++) tot[q] += cnt[w][DPT * threadIdx.x + q];
                    }
                    x += tot[q];
                }
                const u32 mine = x;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const u32 y = __shfl_up(x, off, 64);
                    if (lane >= off) x += y;
                }
                __syncthreads();
                if (lane == 63) wtot[wave] = x;
                __syncthreads();
                u32 e = x - mine;
                for (int w = 0; w < wave; w++) e += wtot[w];
#pragma unroll
                for (int q = 0; q < DPT; q++) {
                    u32 run = e;
                    if (owner) {
#pragma unroll
                        for (int w = 0; w < WAVES; w++) {
                            const u32 c = cnt[w][DPT * threadIdx.x + q];
                            cnt[w][DPT * threadIdx.x + q] = (unsigned short)run;
                            run += c;
                        }
                    }
                    e += tot[q];
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const u32 i = wave_base + (u32)r * 64 + lane;
                if (i < n) {
                    const u32 pos = cnt[wave][(dk[r] >> shift) & dmask] + (di[r] >> 16);
                    sdepth[pos] = dk[r];
                    sidx[pos] = (unsigned short)(di[r] & 0xFFFFu);
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                const u32 i = wave_base + (u32)r * 64 + lane;
                if (i < n) { dk[r] = sdepth[i]; di[r] = sidx[i]; }
            }
        }
        // registers hold the sorted segment in (wave, row, lane) order: coalesced stores, ids gathered from the source
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const u32 i = wave_base + (u32)r * 64 + lane;
            if (i < n) {
                vals_dst[range.x + i] = vals_src[src0 + (di[r] & 0xFFFFu)];
                if (keys_dst) keys_dst[range.x + i] = hi | dk[r];
            }
        }
    }
}

// Segments above the LDS capacities: 8-bit LSD passes through global memory (a scratch slice <-> the final slice, both
// L2 / Infinity-Cache resident), chunk by chunk with running per-digit cursors; one workgroup per segment.
__global__ void __launch_bounds__(F3DG_BLOCK, 2)
tile_sort_long_kernel(const uint2* __restrict__ ranges, const u32* __restrict__ gstart, u32 n_segments,
                      const F3dgHeader* __restrict__ hdr, u32 n_lo /* exclusive */, int tile_bits, int T,
                      const u32* __restrict__ kdepth_src, const u32* __restrict__ vals_src,
                      u64* __restrict__ keys_dst, u32* __restrict__ vals_dst,
                      u64* __restrict__ keys_tmp, u32* __restrict__ vals_tmp)
{
    __shared__ u32 cnt[F3DG_BLOCK / 64][256];
    __shared__ u32 cursor[256];
    __shared__ u32 wtot[F3DG_BLOCK / 64];
    __shared__ u32 skip_flag;
    if (hdr->overflow || hdr->n_long_segments == 0u) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 lane_lt = ((u64)1 << lane) - 1;

    for (u32 seg = blockIdx.x; seg < n_segments; seg += gridDim.x) {
        const uint2 range = ranges[seg];
        const u32 n = range.y - range.x;
        if (n <= n_lo) continue;
        const u32 src0 = gstart[seg];
        __syncthreads();

        // ---------------- long segment: copy to the scratch slice, then ping-pong scratch <-> final slice
        const u64 hi = (u64)(((seg / (u32)T) << tile_bits) | (seg % (u32)T)) << 32;
        for (u32 i = threadIdx.x; i < n; i += F3DG_BLOCK) {
            keys_tmp[range.x + i] = hi | kdepth_src[src0 + i];
            vals_tmp[range.x + i] = vals_src[src0 + i];
        }
        __threadfence_block();
        __syncthreads();
        u64* ksrc = keys_tmp + range.x; u32* vsrc = vals_tmp + range.x;
        u64* kdst = keys_dst + range.x; u32* vdst = vals_dst + range.x;
        bool in_dst = false;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 8 * pass;
            // digit histogram of the whole segment (wave-aggregated: one LDS add per distinct digit per wave round)
#pragma unroll
            for (int w = 0; w < F3DG_BLOCK / 64; w++) cnt[w][threadIdx.x] = 0;
            if (threadIdx.x == 0) skip_flag = 0;
            __syncthreads();
            for (u32 base = 0; base < n; base += F3DG_BLOCK) {
                const u32 i = base + threadIdx.x;
                const bool valid = i < n;
                const u32 d = valid ? ((u32)(ksrc[i] >> shift) & 255u) : 0u;
                (void)wave_rank(d, valid, cnt[wave], lane_lt);
            }
            __syncthreads();
            {
                const u32 d = threadIdx.x;
                const u32 tot = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
                if (tot == n) skip_flag = 1;
                u32 x = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const u32 y = __shfl_up(x, off, 64);
                    if (lane >= off) x += y;
                }
                if (lane == 63) wtot[wave] = x;
                __syncthreads();
                u32 excl = x - tot;
                for (int w = 0; w < wave; w++) excl += wtot[w];
                cursor[d] = excl;
            }
            __syncthreads();
            const bool skip_pass = skip_flag != 0;
            __syncthreads();
            if (skip_pass) continue;

            for (u32 chunk = 0; chunk < n; chunk += F3DG_SORT_CHUNK) {
#pragma unroll
                for (int w = 0; w < F3DG_BLOCK / 64; w++) cnt[w][threadIdx.x] = 0;
                __syncthreads();
                const u32 wave_base = chunk + (u32)wave * (64 * F3DG_SORT_ITEMS);
                u64 key[F3DG_SORT_ITEMS];
                u32 rank[F3DG_SORT_ITEMS];
#pragma unroll
                for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
                    const u32 i = wave_base + (u32)r * 64 + lane;
                    const bool valid = i < n;
                    key[r] = valid ? ksrc[i] : ~(u64)0;
                    rank[r] = 0;
                    if (wave_base + (u32)r * 64 < n)
                        rank[r] = wave_rank((u32)(key[r] >> shift) & 255u, valid, cnt[wave], lane_lt);
                }
                __syncthreads();
                {
                    const u32 d = threadIdx.x;
                    u32 run = cursor[d];
#pragma unroll
                    for (int w = 0; w < F3DG_BLOCK / 64; w++) {
                        const u32 c = cnt[w][d];
                        cnt[w][d] = run;
                        run += c;
                    }
                    cursor[d] = run;
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
                    const u32 i = wave_base + (u32)r * 64 + lane;
                    if (i < n) {
                        const u32 pos = cnt[wave][(u32)(key[r] >> shift) & 255u] + rank[r];
                        kdst[pos] = key[r];
                        vdst[pos] = vsrc[i];
                    }
                }
                __syncthreads();
            }
            { u64* tk = ksrc; ksrc = kdst; kdst = tk; u32* tv = vsrc; vsrc = vdst; vdst = tv; }
            in_dst = !in_dst;
            __threadfence_block();
            __syncthreads();
        }
        if (!in_dst) {                                // an even number of passes moved data: the result sits in the scratch
            for (u32 i = threadIdx.x; i < n; i += F3DG_BLOCK) { kdst[i] = ksrc[i]; vdst[i] = vsrc[i]; }
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
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(F3DG_BLOCK), 0, s, in, (u64)n, tmp);
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(1024), 0, s, tmp, nblocks, hdr_total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nblocks), dim3(F3DG_BLOCK), 0, s, in, out, (u64)n, tmp, exclusive);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

int f3dg_tile_bits(int T) { return bits_for((unsigned long long)T); }

// Number of 8-bit GLOBAL passes: only the tile bits are sorted globally (instances are generated in view order); the 32
// depth bits are sorted per (view, tile) group by tile_sort_kernel.
int f3dg_sort_passes(int V, int T)
{
    (void)V;
    const int bits = f3dg_tile_bits(T);
    return bits == 0 ? 0 : (bits + 7) / 8;
}

template <typename G>
static int binning_tail(hipStream_t s, int V, int P, int grid_x, int grid_y, int T, int tile_bits, const F3dgLayout& L, char* ws,
                        const int* radii, F3dgHeader* hdr, u32* offsets, u32* scan_tmp, u64** keys, u32** vals, u32* hist,
                        uint2* ranges, u32* gstart, u32* gend, u32* gcount, u32 nseg, int keep_keys)
{
    int rc = F3DG_OK;
    const size_t hdr_capacity = (L.keys[1] - L.keys[0]) / 8;      // >= the instance capacity (256-byte aligned carving)
    // 2. keys/values, generated in (view, Gaussian) order, into the half from which the tile pass(es) end in half 1
    const int passes = f3dg_sort_passes(V, T);
    int src = (passes & 1) ? 0 : 1;
    // stream views of a half: depth = first 4 bytes/entry of the key region, grp = the 4 bytes/entry behind all depths
    const size_t cap = (size_t)hdr_capacity;
    auto kdepth = [&](int h) { return reinterpret_cast<u32*>(keys[h]); };
    auto kgrp = [&](int h) { return reinterpret_cast<G*>(reinterpret_cast<u32*>(keys[h]) + cap); };
    hipLaunchKernelGGL((duplicate_keys_kernel<G>), dim3((P + F3DG_BLOCK - 1) / F3DG_BLOCK, V), dim3(F3DG_BLOCK), 0, s, P, tile_bits,
                       grid_x, grid_y, reinterpret_cast<const float2*>(ws + L.means2D),
                       reinterpret_cast<const float*>(ws + L.depths), offsets, radii, hdr, kdepth(src), kgrp(src), vals[src]);

    // 3. level 1: stable LSD radix pass(es) over the TILE bits only, 8 bits per pass (one pass up to 256 tiles)
    const u32 nb = L.sort_blocks;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        hipLaunchKernelGGL((radix_hist_kernel<G>), dim3(nb), dim3(F3DG_BLOCK), 0, s, kgrp(src), hdr, shift, nb, hist);
        rc = f3dg_launch_scan_inclusive(s, hist, hist, (unsigned long long)256 * nb, scan_tmp, L.scan_tmp_elems, 1, nullptr);
        if (rc != F3DG_OK) return rc;
        hipLaunchKernelGGL((radix_scatter_kernel<G>), dim3(nb), dim3(F3DG_BLOCK), 0, s, kdepth(src), kgrp(src), vals[src],
                           kdepth(src ^ 1), kgrp(src ^ 1), vals[src ^ 1], hdr, shift, nb, hist);
        src ^= 1;
    }
    // tile-grouped instances are now in half 1 (src == 1); every (tile, view) group is contiguous, in Gaussian-id order

    // 4. (view, tile) group bounds -> final ranges by a scan over the groups in (view, tile) order
    F3DG_HIP_CHECK(hipMemsetAsync(gstart, 0, sizeof(u32) * 2 * (size_t)nseg, s));      // gstart[nseg] + gend[nseg], adjacent
    hipLaunchKernelGGL((group_bounds_kernel<G>), dim3(2048), dim3(F3DG_BLOCK), 0, s, kgrp(1), hdr, tile_bits, T, gstart, gend);
    hipLaunchKernelGGL(group_counts_kernel, dim3((nseg + F3DG_BLOCK - 1) / F3DG_BLOCK), dim3(F3DG_BLOCK), 0, s, nseg, gstart, gend, gcount);
    rc = f3dg_launch_scan_inclusive(s, gcount, hist /* reuse as gcum */, nseg, scan_tmp, L.scan_tmp_elems, 0, nullptr);
    if (rc != F3DG_OK) return rc;
    hipLaunchKernelGGL(group_ranges_kernel, dim3((nseg + F3DG_BLOCK - 1) / F3DG_BLOCK), dim3(F3DG_BLOCK), 0, s, nseg, gcount, hist, ranges, hdr,
                       (u32)(F3DG_SMALL_THREADS * F3DG_SMALL_ITEMS - 64), (u32)(F3DG_MID_THREADS * F3DG_MID_ITEMS - 64));

    // 5. level 2: per-(view, tile) stable sort by the depth bits: gather the group from half 1, write it sorted to half 0
    //    three tiers by segment length (each kernel skips the segments of the others): <= 4032, <= 16320, longer
    const u32 sort_grid = nseg < 65535u * 16u ? nseg : 65535u * 16u;
    hipLaunchKernelGGL((tile_sort_lds_kernel<F3DG_SMALL_THREADS, F3DG_SMALL_ITEMS>), dim3(sort_grid), dim3(F3DG_SMALL_THREADS), 0, s, ranges, gstart, nseg, hdr, 0u,
                       tile_bits, T, kdepth(1), vals[1], keep_keys ? keys[0] : nullptr, vals[0]);
    const u32 mid_grid = nseg < 2048u ? nseg : 2048u;       // these two stride over all segments and skip most of them
    hipLaunchKernelGGL((tile_sort_lds_kernel<F3DG_MID_THREADS, F3DG_MID_ITEMS>), dim3(mid_grid), dim3(F3DG_MID_THREADS), 0, s, ranges, gstart, nseg, hdr,
                       (u32)(256 * 16 - 64), tile_bits, T, kdepth(1), vals[1], keep_keys ? keys[0] : nullptr, vals[0]);
    hipLaunchKernelGGL(tile_sort_long_kernel, dim3(mid_grid), dim3(F3DG_BLOCK), 0, s, ranges, gstart, nseg, hdr,
                       (u32)(512 * 32 - 64), tile_bits, T, kdepth(1), vals[1], keys[0], vals[0], keys[2], vals[2]);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

int f3dg_launch_binning(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws, const int* radii,
                        int keep_keys)
{
    const int grid_x = (W + F3DG_TILE - 1) / F3DG_TILE, grid_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = grid_x * grid_y;
    const int tile_bits = f3dg_tile_bits(T);
    F3dgHeader* hdr = reinterpret_cast<F3dgHeader*>(ws + L.header);
    u32* tiles = reinterpret_cast<u32*>(ws + L.tiles);
    u32* offsets = reinterpret_cast<u32*>(ws + L.offsets);
    u32* scan_tmp = reinterpret_cast<u32*>(ws + L.scan_tmp);
    u64* keys[3] = { reinterpret_cast<u64*>(ws + L.keys[0]), reinterpret_cast<u64*>(ws + L.keys[1]), reinterpret_cast<u64*>(ws + L.keys[2]) };
    u32* vals[3] = { reinterpret_cast<u32*>(ws + L.vals[0]), reinterpret_cast<u32*>(ws + L.vals[1]), reinterpret_cast<u32*>(ws + L.vals[2]) };
    u32* hist = reinterpret_cast<u32*>(ws + L.hist);
    uint2* ranges = reinterpret_cast<uint2*>(ws + L.ranges);
    u32* gstart = reinterpret_cast<u32*>(ws + L.gstart);
    u32* gend = reinterpret_cast<u32*>(ws + L.gend);
    u32* gcount = reinterpret_cast<u32*>(ws + L.gcount);
    const u32 nseg = (u32)V * (u32)T;

    // 1. inclusive prefix sum of tiles_touched over all (view, Gaussian); total -> header (+ overflow flag)
    int rc = f3dg_launch_scan_inclusive(s, tiles, offsets, (unsigned long long)V * P, scan_tmp, L.scan_tmp_elems, 0, hdr);
    if (rc != F3DG_OK) return rc;

    // 2.-5. with the group stream type that fits (view << tile_bits | tile)
    const bool small = !g_f3dg_sort_wide_groups &&
                       (((unsigned long long)(V > 0 ? V - 1 : 0) << tile_bits) | ((1ull << tile_bits) - 1ull)) <= 0xFFFFull;
    rc = small ? binning_tail<unsigned short>(s, V, P, grid_x, grid_y, T, tile_bits, L, ws, radii, hdr, offsets, scan_tmp, keys, vals,
                                              hist, ranges, gstart, gend, gcount, nseg, keep_keys)
               : binning_tail<u32>(s, V, P, grid_x, grid_y, T, tile_bits, L, ws, radii, hdr, offsets, scan_tmp, keys, vals, hist,
                                   ranges, gstart, gend, gcount, nseg, keep_keys);
    if (rc != F3DG_OK) return rc;
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
