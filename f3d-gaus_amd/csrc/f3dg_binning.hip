// f3dg_binning.hip -- tile binning: prefix sum, key duplication, stable LSD radix sort, tile ranges.
//
// Replaces, for ALL views of a call at once (reference RAST/cuda_rasterizer/rasterizer_impl.cu):
//   cub::DeviceScan::InclusiveSum      :332      -> scan_* kernels (hand-written reduce / scan / propagate)
//   the blocking D2H of num_rendered   :336      -> count stays on the device (workspace header)
//   duplicateWithKeys                  :70-111   -> duplicate_sorted_kernel
//   cub::DeviceRadixSort::SortPairs    :358-363  -> radix2_hist_kernel + scan + radix2_scatter_kernel (8 bits/pass)
//   cudaMemset(ranges) + identifyTileRanges :365, :149-171 -> group_bounds / group_counts / scan / group_ranges kernels
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
// largest binning kernel (1.0 of 2.3 ms at C2, 4.9 of 12 ms at C5: latency-bound), and it sorted every INSTANCE (R = 3-7 per Gaussian). A Gaussian's depth is the same in all its tiles, so the depth order is established once per
// (view, Gaussian) instead:
//   1. gsort: stable LSD radix sort of each view's P Gaussians by their depth bits (4 passes of 8 bits over V*P keys; culled
//      Gaussians get the key ~0 and produce no instances) -> perm[V][P], ascending (depth, Gaussian id);
//   2. tiles_touched gathered in that order, prefix sum, and the instances are GENERATED in (view, depth, id) order;
//   3. the stable tile-digit pass(es) of the old path (two streams now: group u16/u32 + Gaussian id, no depth stream);
//      afterwards every (tile, view) group is contiguous and already in its final (depth, id) order;
//   4. group bounds -> ranges as before, and a copy of every group to its place in the (view, tile)-ordered list.
// Same final list as a stable 64-bit sort of (tile << 32 | depth) keys, i.e. the reference's.

// two-stream stable radix pass over `nseg` segments of seg_len keys each (hdr != null: ONE segment of hdr->num_rendered keys).
// Block b = (segment, chunk): counts of its digit d go to hist[(segment * 256 + d) * cps + chunk], so that one exclusive scan of
// the whole array yields global output positions (every segment owns exactly seg_len consecutive outputs).
template <typename K>
__global__ void __launch_bounds__(F3DG_BLOCK)
radix2_hist_kernel(const K* __restrict__ keys, const F3dgHeader* __restrict__ hdr, u32 seg_len, u32 cps, int shift,
                   u32* __restrict__ hist)
{
    __shared__ u32 h[F3DG_BLOCK / 64][256];
#pragma unroll
    for (int w = 0; w < F3DG_BLOCK / 64; w++) h[w][threadIdx.x] = 0;
    __syncthreads();
    u32 seg = 0, c = blockIdx.x;
    if (!hdr) f3dg_xcd_map(blockIdx.x, gridDim.x / cps, cps, seg, c);      // a view's chunks on one XCD: its key segment stays in that L2
    const u32 n = hdr ? (hdr->overflow ? 0u : hdr->num_rendered) : seg_len;
    const u64 seg_base = (u64)seg * seg_len;
    const u64 base = (u64)c * F3DG_SORT_CHUNK;
    u32* hw = h[threadIdx.x >> 6];
    if (base < n) {
        constexpr int VEC = 16 / (int)sizeof(K);                   // keys per 16-byte load
        typedef K __attribute__((ext_vector_type(VEC))) KV;
        const bool aligned = ((seg_base + base) % VEC) == 0;       // (chunks are multiples of VEC keys; segments need not be)
#pragma unroll
        for (int i = 0; i < F3DG_SORT_ITEMS / VEC; i++) {
            const u64 k = base + ((u64)i * F3DG_BLOCK + threadIdx.x) * VEC;
            if (aligned && k + VEC <= n) {
                const KV w = *reinterpret_cast<const KV*>(keys + seg_base + k);
#pragma unroll
                for (int q = 0; q < VEC; q++) atomicAdd(&hw[((u32)w[q] >> shift) & 255u], 1u);
            } else {
                for (int q = 0; q < VEC; q++)
                    if (k + q < n) atomicAdd(&hw[((u32)keys[seg_base + k + q] >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
    }
    hist[((size_t)seg * 256 + threadIdx.x) * cps + c] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}

template <typename K, bool IOTA = false>      // IOTA: the payload of the input is its position inside the segment (first pass)
__global__ void __launch_bounds__(F3DG_BLOCK)
radix2_scatter_kernel(const K* __restrict__ keys_in, const u32* __restrict__ vals_in, K* __restrict__ keys_out,
                      u32* __restrict__ vals_out, const F3dgHeader* __restrict__ hdr, u32 seg_len, u32 cps, int shift,
                      const u32* __restrict__ offsets /* exclusive scan of hist */)
{
    // as radix_scatter_kernel: the chunk is digit-sorted inside LDS (stable), then written out in coalesced runs
    __shared__ u32 cnt[F3DG_BLOCK / 64][256];
    __shared__ u32 lbase[256];
    __shared__ u32 gdelta[256];
    __shared__ u32 wtot[F3DG_BLOCK / 64];
    __shared__ K skey[F3DG_SORT_CHUNK];
    __shared__ u32 sval[F3DG_SORT_CHUNK];
    u32 seg = 0, c = blockIdx.x;
    if (!hdr) f3dg_xcd_map(blockIdx.x, gridDim.x / cps, cps, seg, c);      // the scattered writes of a view merge in one XCD's L2
    const u32 n = hdr ? (hdr->overflow ? 0u : hdr->num_rendered) : seg_len;
    const u64 seg_base = (u64)seg * seg_len;
    const u64 block_base = (u64)c * F3DG_SORT_CHUNK;
    if (block_base >= n) return;
    const u32 in_block = (u32)((n - block_base) < (u64)F3DG_SORT_CHUNK ? (n - block_base) : (u64)F3DG_SORT_CHUNK);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < F3DG_BLOCK / 64; w++) cnt[w][threadIdx.x] = 0;
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
        key[r] = valid ? (u32)keys_in[seg_base + i] : 0u;
        val[r] = IOTA ? (u32)i : (valid ? vals_in[seg_base + i] : 0u);
    }
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
        const u64 i = wave_base + (u64)r * 64 + lane;
        const bool valid = i < n;
        const u32 d = (key[r] >> shift) & 255u;
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
        gdelta[d] = offsets[((size_t)seg * 256 + d) * cps + c] - excl;
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
            const u32 d = (key[r] >> shift) & 255u;
            const u32 slot = cnt[wave][d] + rank[r];
            skey[slot] = (K)key[r];
            sval[slot] = val[r];
        }
    }
    __syncthreads();
    for (u32 slot = threadIdx.x; slot < in_block; slot += F3DG_BLOCK) {
        const K k = skey[slot];
        const u32 d = ((u32)k >> shift) & 255u;
        const u32 pos = gdelta[d] + slot;          // global position (the scan spans all segments)
        keys_out[pos] = k;
        vals_out[pos] = sval[slot];
    }
}

// tile rectangles (written by the projection kernel: x = rminx | rmaxx << 16, y = rminy | rmaxy << 16) gathered into sorted order,
// with their areas = tiles_touched as the input of the prefix sum that places the instances: the one random gather of the path
__global__ void __launch_bounds__(F3DG_BLOCK)
gsort_gather_rects_kernel(int P, const u32* __restrict__ perm, const uint2* __restrict__ rects, u32* __restrict__ tiles_sorted,
                          u32* __restrict__ rx, u32* __restrict__ ry)
{
    unsigned view, chunk;                     // a view's rectangles (8 P bytes) are gathered through one XCD's L2
    const unsigned cpv = (unsigned)((P + F3DG_BLOCK - 1) / F3DG_BLOCK);
    f3dg_xcd_map(blockIdx.x, gridDim.x / cpv, cpv, view, chunk);
    const int k = (int)(chunk * F3DG_BLOCK + threadIdx.x);
    if (k >= P) return;
    const size_t vb = (size_t)view * P;
    const uint2 r = rects[vb + perm[vb + k]];
    tiles_sorted[vb + k] = ((r.x >> 16) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.y & 0xFFFFu));
    rx[vb + k] = r.x;
    ry[vb + k] = r.y;
}

// duplicateWithKeys in (view, depth, id) order: sorted position k of view v emits the tiles of Gaussian perm[v][k]
template <typename G>
__global__ void __launch_bounds__(F3DG_BLOCK)
duplicate_sorted_kernel(int P, int tile_bits, int grid_x, const u32* __restrict__ perm, const u32* __restrict__ rx,
                        const u32* __restrict__ ry, const u32* __restrict__ offsets_sorted, const F3dgHeader* __restrict__ hdr,
                        G* __restrict__ kgrp, u32* __restrict__ vals)
{
    if (hdr->overflow) return;
    const int k = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    const int v = blockIdx.y;
    if (k >= P) return;
    const size_t pos = (size_t)v * P + k;
    const u32 x = rx[pos], y = ry[pos];
    const u32 rminx = x & 0xFFFFu, rmaxx = x >> 16, rminy = y & 0xFFFFu, rmaxy = y >> 16;
    if (rmaxx > rminx && rmaxy > rminy) {
        const u32 g = perm[pos];
        u32 off = (pos == 0) ? 0 : offsets_sorted[pos - 1];
        const u32 view_base = (u32)v << tile_bits;
        for (u32 ty = rminy; ty < rmaxy; ty++)
            for (u32 tx = rminx; tx < rmaxx; tx++) {
                kgrp[off] = (G)(view_base | (ty * (u32)grid_x + tx));
                vals[off] = g;
                off++;
            }
    }
}

// every (view, tile) group from where the tile pass left it to its place in the (view, tile)-ordered list
__global__ void __launch_bounds__(F3DG_BLOCK)
regroup_kernel(u32 nseg, const uint2* __restrict__ ranges, const u32* __restrict__ gstart, const F3dgHeader* __restrict__ hdr,
               const u32* __restrict__ vals_in, u32* __restrict__ vals_out)
{
    if (hdr->overflow) return;
    for (u32 seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint2 r = ranges[seg];
        const u32 n = r.y - r.x;
        const u32 src = gstart[seg];
        for (u32 i = threadIdx.x; i < n; i += F3DG_BLOCK)
            vals_out[r.x + i] = vals_in[src + i];
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
            keys_out[i] = hi | (u64)__float_as_uint(depths[(size_t)view * P + point_list[i]]);
    }
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
                    uint2* __restrict__ ranges)
{
    const u32 i = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    if (i < nseg) {
        const u32 c = gcount[i];
        ranges[i] = c ? make_uint2(gcum[i] - c, gcum[i]) : make_uint2(0u, 0u);
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

// Number of 8-bit passes over the instances: only the tile bits (instances are generated in (view, depth, id) order).
int f3dg_sort_passes(int V, int T)
{
    (void)V;
    const int bits = f3dg_tile_bits(T);
    return bits == 0 ? 0 : (bits + 7) / 8;
}

template <typename G>
static int binning_tail(hipStream_t s, int V, int P, int grid_x, int T, int tile_bits, const F3dgLayout& L, char* ws, F3dgHeader* hdr,
                        u32* scan_tmp, u64** keys, u32** vals, u32* hist, uint2* ranges, u32* gstart, u32* gend, u32* gcount, u32 nseg)
{
    int rc = F3DG_OK;
    const size_t VP = (size_t)V * P;
    u32* gk[2] = { reinterpret_cast<u32*>(ws + L.gsort), reinterpret_cast<u32*>(ws + L.gsort) + VP };
    u32* gv[2] = { reinterpret_cast<u32*>(ws + L.gsort) + 2 * VP, reinterpret_cast<u32*>(ws + L.gsort) + 3 * VP };
    const dim3 pgrid((P + F3DG_BLOCK - 1) / F3DG_BLOCK, V);

    // 1. per-view stable sort of the Gaussians by their sort keys (written by the projection kernel into gk[0]: the depth bits,
    //    ~0 for Gaussians that touch no tile); the first pass takes the Gaussian id from the position
    const u32 cps = (u32)((P + F3DG_SORT_CHUNK - 1) / F3DG_SORT_CHUNK);
    const u32 gblocks = (u32)V * cps;
    int cur = 0;
    for (int pass = 0; pass < 4; pass++) {
        hipLaunchKernelGGL((radix2_hist_kernel<u32>), dim3(gblocks), dim3(F3DG_BLOCK), 0, s, gk[cur], (const F3dgHeader*)nullptr, (u32)P, cps,
                           8 * pass, hist);
        rc = f3dg_launch_scan_inclusive(s, hist, hist, (unsigned long long)256 * gblocks, scan_tmp, L.scan_tmp_elems, 1, nullptr);
        if (rc != F3DG_OK) return rc;
        if (pass == 0)
            hipLaunchKernelGGL((radix2_scatter_kernel<u32, true>), dim3(gblocks), dim3(F3DG_BLOCK), 0, s, gk[cur], (const u32*)nullptr, gk[cur ^ 1],
                               gv[cur ^ 1], (const F3dgHeader*)nullptr, (u32)P, cps, 8 * pass, hist);
        else
            hipLaunchKernelGGL((radix2_scatter_kernel<u32, false>), dim3(gblocks), dim3(F3DG_BLOCK), 0, s, gk[cur], gv[cur], gk[cur ^ 1], gv[cur ^ 1],
                               (const F3dgHeader*)nullptr, (u32)P, cps, 8 * pass, hist);
        cur ^= 1;
    }
    const u32* perm = gv[cur];            // cur == 0 after four passes; the three other buffers are free now
    u32* offsets_sorted = gk[cur ^ 1];    // tiles_touched in sorted order, then its inclusive prefix sum (in place)
    u32* rx = gk[cur];
    u32* ry = gv[cur ^ 1];

    // 2. instances in (view, depth, id) order
    hipLaunchKernelGGL(gsort_gather_rects_kernel, dim3(pgrid.x * pgrid.y), dim3(F3DG_BLOCK), 0, s, P, perm, reinterpret_cast<const uint2*>(ws + L.rects),
                       offsets_sorted, rx, ry);
    rc = f3dg_launch_scan_inclusive(s, offsets_sorted, offsets_sorted, (unsigned long long)VP, scan_tmp, L.scan_tmp_elems, 0, hdr);
    if (rc != F3DG_OK) return rc;
    auto kgrp = [&](int h) { return reinterpret_cast<G*>(keys[h]); };         // group stream of a half (the old depth stream's place)
    const int passes = f3dg_sort_passes(V, T);
    int src = (passes & 1) ? 0 : 1;                                            // so that the tile pass(es) end in half 1
    hipLaunchKernelGGL((duplicate_sorted_kernel<G>), pgrid, dim3(F3DG_BLOCK), 0, s, P, tile_bits, grid_x, perm, rx, ry, offsets_sorted, hdr,
                       kgrp(src), vals[src]);

    // 3. stable pass(es) over the tile bits
    const u32 nb = L.sort_blocks;
    for (int p = 0; p < passes; p++) {
        hipLaunchKernelGGL((radix2_hist_kernel<G>), dim3(nb), dim3(F3DG_BLOCK), 0, s, kgrp(src), hdr, 0u, nb, 8 * p, hist);
        rc = f3dg_launch_scan_inclusive(s, hist, hist, (unsigned long long)256 * nb, scan_tmp, L.scan_tmp_elems, 1, nullptr);
        if (rc != F3DG_OK) return rc;
        hipLaunchKernelGGL((radix2_scatter_kernel<G, false>), dim3(nb), dim3(F3DG_BLOCK), 0, s, kgrp(src), vals[src], kgrp(src ^ 1), vals[src ^ 1],
                           hdr, 0u, nb, 8 * p, hist);
        src ^= 1;
    }
    if (passes == 0) src = 1;
    // 4. group bounds -> ranges; copy every group to its final place
    F3DG_HIP_CHECK(hipMemsetAsync(gstart, 0, sizeof(u32) * 2 * (size_t)nseg, s));
    hipLaunchKernelGGL((group_bounds_kernel<G>), dim3(2048), dim3(F3DG_BLOCK), 0, s, kgrp(1), hdr, tile_bits, T, gstart, gend);
    hipLaunchKernelGGL(group_counts_kernel, dim3((nseg + F3DG_BLOCK - 1) / F3DG_BLOCK), dim3(F3DG_BLOCK), 0, s, nseg, gstart, gend, gcount);
    rc = f3dg_launch_scan_inclusive(s, gcount, hist /* reuse as gcum */, nseg, scan_tmp, L.scan_tmp_elems, 0, nullptr);
    if (rc != F3DG_OK) return rc;
    hipLaunchKernelGGL(group_ranges_kernel, dim3((nseg + F3DG_BLOCK - 1) / F3DG_BLOCK), dim3(F3DG_BLOCK), 0, s, nseg, gcount, hist, ranges);
    const u32 rg = nseg < 65535u * 4u ? nseg : 65535u * 4u;
    hipLaunchKernelGGL(regroup_kernel, dim3(rg), dim3(F3DG_BLOCK), 0, s, nseg, ranges, gstart, hdr, vals[1], vals[0]);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

int f3dg_launch_binning(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws, int export_offsets)
{
    const int grid_x = (W + F3DG_TILE - 1) / F3DG_TILE, grid_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = grid_x * grid_y;
    const int tile_bits = f3dg_tile_bits(T);
    F3dgHeader* hdr = reinterpret_cast<F3dgHeader*>(ws + L.header);
    u32* scan_tmp = reinterpret_cast<u32*>(ws + L.scan_tmp);
    u64* keys[2] = { reinterpret_cast<u64*>(ws + L.keys[0]), reinterpret_cast<u64*>(ws + L.keys[1]) };
    u32* vals[2] = { reinterpret_cast<u32*>(ws + L.vals[0]), reinterpret_cast<u32*>(ws + L.vals[1]) };
    u32* hist = reinterpret_cast<u32*>(ws + L.hist);
    uint2* ranges = reinterpret_cast<uint2*>(ws + L.ranges);
    u32* gstart = reinterpret_cast<u32*>(ws + L.gstart);
    u32* gend = reinterpret_cast<u32*>(ws + L.gend);
    u32* gcount = reinterpret_cast<u32*>(ws + L.gcount);
    const u32 nseg = (u32)V * (u32)T;

    int rc = F3DG_OK;
    if (export_offsets) {
        // the (view, Gaussian)-ordered prefix sum of the reference (point_offsets) is only an exported intermediate here
        rc = f3dg_launch_scan_inclusive(s, reinterpret_cast<const u32*>(ws + L.tiles), reinterpret_cast<u32*>(ws + L.offsets),
                                        (unsigned long long)V * P, scan_tmp, L.scan_tmp_elems, 0, nullptr);
        if (rc != F3DG_OK) return rc;
    }
    // group stream type that fits (view << tile_bits | tile)
    const bool small = !g_f3dg_sort_wide_groups &&
                       (((unsigned long long)(V > 0 ? V - 1 : 0) << tile_bits) | ((1ull << tile_bits) - 1ull)) <= 0xFFFFull;
    rc = small ? binning_tail<unsigned short>(s, V, P, grid_x, T, tile_bits, L, ws, hdr, scan_tmp, keys, vals, hist, ranges, gstart, gend, gcount, nseg)
               : binning_tail<u32>(s, V, P, grid_x, T, tile_bits, L, ws, hdr, scan_tmp, keys, vals, hist, ranges, gstart, gend, gcount, nseg);
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
    hipLaunchKernelGGL(export_keys_kernel, dim3(rg), dim3(F3DG_BLOCK), 0, s, nseg, P, f3dg_tile_bits(T), T,
                       reinterpret_cast<const uint2*>(ws + L.ranges), reinterpret_cast<const F3dgHeader*>(ws + L.header),
                       reinterpret_cast<const u32*>(ws + L.vals[0]), reinterpret_cast<const float*>(ws + L.depths),
                       reinterpret_cast<u64*>(ws + L.keys[0]));
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
