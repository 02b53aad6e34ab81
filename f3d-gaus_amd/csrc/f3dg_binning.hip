// f3dg_binning.hip -- tile binning: prefix sum, key duplication, stable LSD radix sort, tile ranges.
//
// Replaces, for ALL views of a call at once (reference RAST/cuda_rasterizer/rasterizer_impl.cu):
//   cub::DeviceScan::InclusiveSum      :332      -> scan_* kernels (hand-written reduce / scan / propagate)
//   the blocking D2H of num_rendered   :336      -> count stays on the device (workspace header)
//   duplicateWithKeys                  :70-111   -> duplicate_keys_kernel
//   cub::DeviceRadixSort::SortPairs    :358-363  -> radix_hist_kernel + scan + radix_scatter_kernel (8 bits/pass)
//   cudaMemset(ranges) + identifyTileRanges :365, :149-171 -> tile_ranges_kernel
//
// Keys are (view * T + tile) << 32 | float_bits(depth): the view index rides in the high bits so that one sort
// orders every view of the batch; within a view the order is exactly the reference's (tile, depth, then input
// order = ascending Gaussian id, because the sort is stable). Depths are > 0.2 so their IEEE bits order as
// unsigned integers.
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
    if (threadIdx.x == 0) carry_s = 0;
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
        __syncthreads();
        u32 wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += wtot[w];
        const u32 carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + wave_off + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + x;
        __syncthreads();
    }
    if (hdr && threadIdx.x == 0) {
        const u32 total = carry_s;
        hdr->num_rendered = total;
        hdr->overflow = total > hdr->capacity ? 1u : 0u;
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
#pragma unroll
    for (int i = 0; i < F3DG_SCAN_ITEMS; i++) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
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
#pragma unroll
    for (int i = 0; i < F3DG_SCAN_ITEMS; i++) {
        const u32 before = run;
        run += v[i];
        if (base + i < n) out[base + i] = exclusive ? before : run;
    }
}

// ------------------------------------------------------------------------------------------------ keys
__global__ void __launch_bounds__(F3DG_BLOCK)
duplicate_keys_kernel(int P, int T, int grid_x, int grid_y, const float2* __restrict__ means2D,
                      const F3dgRec* __restrict__ rec, const u32* __restrict__ offsets,
                      const int* __restrict__ radii, const F3dgHeader* __restrict__ hdr,
                      u64* __restrict__ keys, u32* __restrict__ vals)
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
        const u32 depth_bits = __float_as_uint(rec[idx].f[14]);
        const u64 view_base = (u64)v * (u64)T;
        for (int y = rminy; y < rmaxy; y++)
            for (int x = rminx; x < rmaxx; x++) {
                u64 key = view_base + (u64)(y * grid_x + x);
                key <<= 32;
                key |= depth_bits;
                keys[off] = key;
                vals[off] = (u32)g;
                off++;
            }
    }
}

// ------------------------------------------------------------------------------------------------ sort
__global__ void __launch_bounds__(F3DG_BLOCK)
radix_hist_kernel(const u64* __restrict__ keys, const F3dgHeader* __restrict__ hdr, int shift, u32 nblocks,
                  u32* __restrict__ hist /* [256][nblocks] */)
{
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u32 n = hdr->overflow ? 0u : hdr->num_rendered;
    const u64 base = (u64)blockIdx.x * F3DG_SORT_CHUNK;
    if (base < n) {
#pragma unroll
        for (int i = 0; i < F3DG_SORT_ITEMS; i++) {
            const u64 k = base + (u64)i * F3DG_BLOCK + threadIdx.x;
            if (k < n) atomicAdd(&h[(u32)(keys[k] >> shift) & 255u], 1u);
        }
        __syncthreads();
    }
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(F3DG_BLOCK)
radix_scatter_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ vals_in, u64* __restrict__ keys_out,
                     u32* __restrict__ vals_out, const F3dgHeader* __restrict__ hdr, int shift, u32 nblocks,
                     const u32* __restrict__ offsets /* exclusive scan of hist, [256][nblocks] */)
{
    __shared__ u32 cnt[F3DG_BLOCK / 64][256];
    const u32 n = hdr->overflow ? 0u : hdr->num_rendered;
    const u64 block_base = (u64)blockIdx.x * F3DG_SORT_CHUNK;
    if (block_base >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < F3DG_BLOCK / 64; w++) cnt[w][threadIdx.x] = 0;
    __syncthreads();

    const u64 wave_base = block_base + (u64)wave * (64 * F3DG_SORT_ITEMS);
    u64 key[F3DG_SORT_ITEMS];
    u32 rank[F3DG_SORT_ITEMS];
    const u64 lane_lt = ((u64)1 << lane) - 1;
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
        const u64 i = wave_base + (u64)r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : ~(u64)0;
        const u32 d = (u32)(key[r] >> shift) & 255u;
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
        u32 run = offsets[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < F3DG_BLOCK / 64; w++) {
            const u32 c = cnt[w][d];
            cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < F3DG_SORT_ITEMS; r++) {
        const u64 i = wave_base + (u64)r * 64 + lane;
        if (i < n) {
            const u32 d = (u32)(key[r] >> shift) & 255u;
            const u32 pos = cnt[wave][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = vals_in[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ ranges
__global__ void __launch_bounds__(F3DG_BLOCK)
tile_ranges_kernel(const u64* __restrict__ keys, const F3dgHeader* __restrict__ hdr, uint2* __restrict__ ranges)
{
    const u32 L = hdr->overflow ? 0u : hdr->num_rendered;
    for (u64 idx = (u64)blockIdx.x * F3DG_BLOCK + threadIdx.x; idx < L; idx += (u64)gridDim.x * F3DG_BLOCK) {
        const u32 currtile = (u32)(keys[idx] >> 32);
        if (idx == 0)
            ranges[currtile].x = 0;
        else {
            const u32 prevtile = (u32)(keys[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[prevtile].y = (u32)idx;
                ranges[currtile].x = (u32)idx;
            }
        }
        if (idx == L - 1)
            ranges[currtile].y = L;
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

// Number of 8-bit passes needed for V views of T tiles: key bits = 32 (depth) + bits(V*T).
int f3dg_sort_passes(int V, int T)
{
    const int bits = 32 + bits_for((unsigned long long)V * (unsigned long long)T);
    return (bits + 7) / 8;
}

int f3dg_launch_binning(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws, const int* radii)
{
    const int grid_x = (W + F3DG_TILE - 1) / F3DG_TILE, grid_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = grid_x * grid_y;
    F3dgHeader* hdr = reinterpret_cast<F3dgHeader*>(ws + L.header);
    u32* tiles = reinterpret_cast<u32*>(ws + L.tiles);
    u32* offsets = reinterpret_cast<u32*>(ws + L.offsets);
    u32* scan_tmp = reinterpret_cast<u32*>(ws + L.scan_tmp);
    u64* keys[2] = { reinterpret_cast<u64*>(ws + L.keys[0]), reinterpret_cast<u64*>(ws + L.keys[1]) };
    u32* vals[2] = { reinterpret_cast<u32*>(ws + L.vals[0]), reinterpret_cast<u32*>(ws + L.vals[1]) };
    u32* hist = reinterpret_cast<u32*>(ws + L.hist);
    uint2* ranges = reinterpret_cast<uint2*>(ws + L.ranges);

    // 1. inclusive prefix sum of tiles_touched over all (view, Gaussian); total -> header (+ overflow flag)
    int rc = f3dg_launch_scan_inclusive(s, tiles, offsets, (unsigned long long)V * P, scan_tmp, L.scan_tmp_elems, 0, hdr);
    if (rc != F3DG_OK) return rc;

    // 2. keys/values into ping-pong half `src`; chosen so that the final pass lands in half 0
    const int passes = f3dg_sort_passes(V, T);
    int src = passes & 1;
    hipLaunchKernelGGL(duplicate_keys_kernel, dim3((P + F3DG_BLOCK - 1) / F3DG_BLOCK, V), dim3(F3DG_BLOCK), 0, s, P, T,
                       grid_x, grid_y, reinterpret_cast<const float2*>(ws + L.means2D),
                       reinterpret_cast<const F3dgRec*>(ws + L.rec), offsets, radii, hdr, keys[src], vals[src]);

    // 3. stable LSD radix sort, 8 bits per pass
    const u32 nb = L.sort_blocks;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(F3DG_BLOCK), 0, s, keys[src], hdr, shift, nb, hist);
        rc = f3dg_launch_scan_inclusive(s, hist, hist, (unsigned long long)256 * nb, scan_tmp, L.scan_tmp_elems, 1, nullptr);
        if (rc != F3DG_OK) return rc;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(F3DG_BLOCK), 0, s, keys[src], vals[src], keys[src ^ 1],
                           vals[src ^ 1], hdr, shift, nb, hist);
        src ^= 1;
    }
    // sorted result is now in half 0 (src == 0)

    // 4. tile ranges
    F3DG_HIP_CHECK(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)V * T, s));
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(2048), dim3(F3DG_BLOCK), 0, s, keys[0], hdr, ranges);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
