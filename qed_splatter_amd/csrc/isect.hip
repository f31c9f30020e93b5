// K3 + K5: tile intersection (scan + emit) and per-tile offsets of the sorted list.
// Replaces gsplat isect_tiles / isect_offset_encode behind model.py:267-288
// (SURVEY.md Appendix A.4-A.5).  Integer / byte work, HBM-bound.
#include "qed_common.h"

namespace qed {

// ---- exclusive scan of block_sums (one workgroup; n_blocks is N/256, i.e. thousands) -----------
__global__ void __launch_bounds__(1024)
isect_scan_kernel(const int* __restrict__ block_sums, int n_blocks, int* __restrict__ block_offsets,
                  int* __restrict__ n_isect, long long capacity, int* __restrict__ status) {
    __shared__ long long wave_tot[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + tid;
        const long long v = i < n_blocks ? (long long)block_sums[i] : 0;
        // inclusive scan within the wave
        long long x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        __syncthreads();
        long long wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += wave_tot[w];
        const long long carry = carry_s;
        const long long excl = carry + wbase + x - v;
        if (i < n_blocks) block_offsets[i] = (int)min(excl, (long long)0x7fffffff);
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + x;
        __syncthreads();
    }
    if (tid == 0) {
        const long long total = carry_s;
        if (total > capacity || total > 0x7fffffffll) {
            status[0] = (int)min(total, (long long)0x7fffffff);
            n_isect[0] = 0;  // downstream kernels then do nothing
        } else {
            n_isect[0] = (int)total;
        }
    }
}

// ---- emit (key, value) pairs -------------------------------------------------------------------
// One wave owns 64 consecutive (camera,Gaussian) slots and writes their intersections
// cooperatively: output slot j of the wave's range belongs to the Gaussian found by binary search
// over the wave's prefix sums, so consecutive lanes write consecutive addresses.
__global__ void __launch_bounds__(256)
isect_emit_kernel(int N, int C, const float* __restrict__ means2d, const int* __restrict__ radii,
                  const float* __restrict__ depths, const int* __restrict__ tiles_per_gauss,
                  const int* __restrict__ block_offsets, int tile_w, int tile_h, int tile_bits,
                  const int* __restrict__ n_isect, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    __shared__ int s_pref[4][65];   // per wave: exclusive prefix of counts (+ total)
    __shared__ int s_x0[4][64], s_y0[4][64], s_w[4][64];
    __shared__ unsigned s_depth[4][64];
    __shared__ int s_wave_tot[4];
    if (n_isect[0] == 0) return;    // nothing to do (or capacity exceeded)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long total = (long long)C * N;
    const long long slot = (long long)blockIdx.x * 256 + tid;
    int cnt = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    unsigned dbits = 0;
    if (slot < total) {
        cnt = tiles_per_gauss[slot];
        if (cnt > 0) {
            tile_rect(means2d[2 * slot], means2d[2 * slot + 1], (float)radii[slot], tile_w, tile_h, x0, y0, x1, y1);
            dbits = __float_as_uint(depths[slot]);
        }
    }
    // wave-inclusive scan of the counts
    int x = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    s_pref[wid][lane] = x - cnt;
    if (lane == 63) { s_pref[wid][64] = x; s_wave_tot[wid] = x; }
    s_x0[wid][lane] = x0; s_y0[wid][lane] = y0; s_w[wid][lane] = x1 - x0; s_depth[wid][lane] = dbits;
    __syncthreads();
    int wave_base = block_offsets[blockIdx.x];
    for (int w = 0; w < wid; ++w) wave_base += s_wave_tot[w];
    const int wtot = s_pref[wid][64];
    const long long slot0 = (long long)blockIdx.x * 256 + wid * 64;
    for (int j = lane; j < wtot; j += 64) {
        // largest g with pref[g] <= j
        int lo = 0, hi = 63;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_pref[wid][mid] <= j) lo = mid; else hi = mid - 1;
        }
        const int g = lo;
        const int local = j - s_pref[wid][g];
        const int w = s_w[wid][g];
        const int ty = s_y0[wid][g] + local / w;
        const int tx = s_x0[wid][g] + local % w;
        const long long sl = slot0 + g;
        const unsigned long long cam = (unsigned long long)(sl / N);
        const unsigned long long tile = (unsigned long long)(ty * tile_w + tx);
        keys[wave_base + j] = (((cam << tile_bits) | tile) << 32) | (unsigned long long)s_depth[wid][g];
        vals[wave_base + j] = (int)sl;
    }
}

// ---- tile offsets ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
tile_offsets_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_dev, int n_tiles_total,
                    int n_tiles, int tile_bits, int* __restrict__ offsets) {
    const int n = n_dev[0];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n == 0) {
        if (i <= n_tiles_total) offsets[i] = 0;
        return;
    }
    if (i >= n) return;
    auto lin = [&](unsigned long long k) -> int {
        const unsigned long long ct = k >> 32;
        return (int)((ct >> tile_bits) * (unsigned long long)n_tiles + (ct & ((1ull << tile_bits) - 1ull)));
    };
    const int cur = lin(keys[i]);
    if (i == 0) {
        for (int t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
        const int prev = lin(keys[i - 1]);
        for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int)i;
    }
    if (i == n - 1) {
        for (int t = cur + 1; t <= n_tiles_total; ++t) offsets[t] = n;
    }
}

}  // namespace qed

using namespace qed;

extern "C" int qed_isect_scan(const int32_t* block_sums, int32_t n_blocks, int32_t* block_offsets, int32_t* n_isect,
                              int64_t capacity, int32_t* status, void* stream) {
    QED_REQUIRE(n_blocks >= 0 && n_isect && status, "bad arguments");
    QED_REQUIRE(n_blocks == 0 || (block_sums && block_offsets), "null buffers");
    hipLaunchKernelGGL(isect_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, block_sums, n_blocks,
                       block_offsets, n_isect, (long long)capacity, status);
    return check_launch("qed_isect_scan");
}

extern "C" int qed_isect_emit(int32_t N, int32_t C, const float* means2d, const int32_t* radii, const float* depths,
                              const int32_t* tiles_per_gauss, const int32_t* block_offsets, int32_t tile_w,
                              int32_t tile_h, int32_t tile_bits, const int32_t* n_isect, int64_t capacity,
                              uint64_t* keys, int32_t* vals, void* stream) {
    QED_REQUIRE(N >= 0 && C >= 1 && tile_bits > 0 && tile_bits < 31, "bad arguments");
    (void)capacity;
    if (N == 0) return QED_OK;
    QED_REQUIRE(means2d && radii && depths && tiles_per_gauss && block_offsets && n_isect && keys && vals,
                "null buffers");
    const long long total = (long long)C * N;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(isect_emit_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, C, means2d, radii, depths,
                       tiles_per_gauss, block_offsets, tile_w, tile_h, tile_bits, n_isect,
                       (unsigned long long*)keys, vals);
    return check_launch("qed_isect_emit");
}

extern "C" int qed_tile_offsets(const uint64_t* sorted_keys, const int32_t* n_dev, int64_t capacity, int32_t C,
                                int32_t n_tiles, int32_t tile_bits, int32_t* offsets, void* stream) {
    QED_REQUIRE(n_dev && offsets && C >= 1 && n_tiles >= 1, "bad arguments");
    QED_REQUIRE(capacity >= 0 && capacity < (1ll << 31), "capacity out of range");
    const long long n_tot = (long long)C * n_tiles;
    const long long work = capacity > n_tot + 1 ? capacity : n_tot + 1;
    const unsigned grid = (unsigned)((work + 255) / 256);
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)sorted_keys, n_dev, (int)n_tot, n_tiles, tile_bits, offsets);
    return check_launch("qed_tile_offsets");
}
