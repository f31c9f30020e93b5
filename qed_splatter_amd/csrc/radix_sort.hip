// K4: device LSD radix sort of (key, i32 value) pairs for gfx950 -- stable, 8-bit digits.
//
// Replaces the cub::DeviceRadixSort::SortPairs call inside gsplat's isect_tiles (behind
// model.py:267-288).  Written for 64-wide waves: the in-block stable rank
// uses 64-bit ballots ("match" on the digit), one LDS counter row per wave, and the scatter goes
// through an LDS-staged reorder so that every digit run is written with consecutive lanes on
// consecutive addresses.  HBM-bound: per pass each pair is read twice (histogram + scatter) and
// written once.
//
// The element count is read from device memory (n_dev) so the whole pipeline can be enqueued
// without a host round trip (and captured into a hipGraph); grids are sized by `capacity`.
#include "qed_common.h"

namespace qed {

constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

template <typename KeyT, int KPT>
struct SortCfg {
    static constexpr int kItems = kSortThreads * KPT;
};

template <typename KeyT>
__device__ __forceinline__ unsigned digit_of(KeyT k, int shift, unsigned mask) {
    return (unsigned)(k >> shift) & mask;
}

// ---- (1) per-block digit histogram ----------------------------------------------------------------
template <typename KeyT, int KPT>
__global__ void __launch_bounds__(kSortThreads)
sort_hist_kernel(const KeyT* __restrict__ keys, const int* __restrict__ n_dev, int shift, unsigned mask,
                 int nblocks_max, int* __restrict__ hist) {
    constexpr int kItems = SortCfg<KeyT, KPT>::kItems;
    const int n = n_dev[0];
    const long long base = (long long)blockIdx.x * kItems;
    if (base >= n) return;
    __shared__ int s_hist[kSortWaves][kRadix];
    const int tid = threadIdx.x, wid = tid >> 6;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) s_hist[w][tid] = 0;
    __syncthreads();
    // all KPT keys requested together (unconditional, clamped index), then counted: a load under `if (i < n)`
    // followed by its use is one dependent round trip per key
    KeyT key[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const long long i = base + k * kSortThreads + tid;
        key[k] = keys[i < n ? i : (long long)n - 1];
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const long long i = base + k * kSortThreads + tid;
        if (i < n) atomicAdd(&s_hist[wid][digit_of<KeyT>(key[k], shift, mask)], 1);
    }
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) tot += s_hist[w][tid];
    hist[(long long)tid * nblocks_max + blockIdx.x] = tot;
}

// ---- (2) scan: one workgroup per digit row -> row-exclusive prefix in place + digit totals ------------
template <int ITEMS>
__global__ void __launch_bounds__(kSortThreads)
sort_scan_kernel(const int* __restrict__ n_dev, int nblocks_max, int* __restrict__ hist, int* __restrict__ digit_tot) {
    const int n = n_dev[0];
    const int nb = (int)(((long long)n + ITEMS - 1) / ITEMS);
    int* row = hist + (long long)blockIdx.x * nblocks_max;
    __shared__ int s_wave[kSortWaves];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    // eight 256-wide slices per outer step: their loads are issued together (unconditional, clamped index), so a
    // row of 1 640 block counts costs one memory round trip instead of seven dependent ones
    constexpr int kSlices = 8;
    for (int b0 = 0; b0 < nb; b0 += kSlices * kSortThreads) {
        int vals[kSlices];
#pragma unroll
        for (int j = 0; j < kSlices; ++j) {
            const int i = b0 + j * kSortThreads + tid;
            vals[j] = row[i < nb ? i : nb - 1];
        }
#pragma unroll
        for (int j = 0; j < kSlices; ++j) {
            const int i = b0 + j * kSortThreads + tid;
            if (b0 + j * kSortThreads >= nb) break;              // block-uniform
            const int v = i < nb ? vals[j] : 0;
            int x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(x, o, 64);
                if (lane >= o) x += y;
            }
            if (lane == 63) s_wave[wid] = x;
            __syncthreads();
            int wb = 0;
            for (int w = 0; w < wid; ++w) wb += s_wave[w];
            const int carry = s_carry;
            if (i < nb) row[i] = carry + wb + x - v;
            __syncthreads();
            if (tid == kSortThreads - 1) s_carry = carry + wb + x;
            __syncthreads();
        }
    }
    if (tid == 0) digit_tot[blockIdx.x] = s_carry;
}

// ---- (3) stable rank + scatter --------------------------------------------------------------------------
// LOOKBACK = false: per-block digit bases come from the histogram + scan kernels above.
// LOOKBACK = true ("onesweep"): one kernel per pass.  Workgroups take tiles in ticket order; each
// publishes its per-digit count in a 32-bit word {2-bit state, 30-bit value} with a relaxed
// agent-scope store (payload and flag travel in ONE word, so no other ordering is needed -- the
// granule hand-off of the CDNA4 guide) and sums its predecessors' words (decoupled look-back).  A
// tile only ever waits for tiles with smaller tickets, which are already running; every spin is
// bounded and reports through status[1] instead of hanging.
constexpr unsigned kLbPartial = 1u << 30, kLbInclusive = 2u << 30, kLbValue = (1u << 30) - 1u;
constexpr int kLbSpinLimit = 1 << 18;

template <typename KeyT, int KPT, bool LOOKBACK>
__global__ void __launch_bounds__(kSortThreads)
sort_scatter_kernel(const KeyT* __restrict__ keys_in, const int* __restrict__ vals_in, KeyT* __restrict__ keys_out,
                    int* __restrict__ vals_out, const int* __restrict__ n_dev, int shift, unsigned mask,
                    int nblocks_max, const int* __restrict__ hist, const int* __restrict__ digit_tot,
                    unsigned* __restrict__ lookback, int* __restrict__ ticket, int* __restrict__ status) {
    constexpr int kItems = SortCfg<KeyT, KPT>::kItems;
    const int n = n_dev[0];
    int tile = blockIdx.x;
    if constexpr (LOOKBACK) {
        __shared__ int s_tile;
        if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1);
        __syncthreads();
        tile = s_tile;
    }
    const long long base = (long long)tile * kItems;
    if (base >= n) return;
    const int block_n = (int)min((long long)kItems, (long long)n - base);

    __shared__ KeyT s_keys[kItems];
    __shared__ int s_vals[kItems];
    __shared__ int s_cnt[kSortWaves][kRadix];   // per-wave digit counters -> per-wave digit bases
    __shared__ int s_lbase[kRadix];             // block-local exclusive digit base
    __shared__ int s_gofs[kRadix];              // global offset - local base
    __shared__ int s_wsum[kSortWaves];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) s_cnt[w][tid] = 0;

    // wave-striped load: wave w owns [w*64*KPT, (w+1)*64*KPT), item k of lane l = k*64 + l
    KeyT key[KPT];
    int val[KPT];
    int rank[KPT];
    const int wbase = wid * 64 * KPT;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int li = wbase + k * 64 + lane;
        if (li < block_n) {
            key[k] = keys_in[base + li];
            val[k] = vals_in[base + li];
        } else {
            key[k] = (KeyT)0;
            val[k] = 0;
        }
    }
    __syncthreads();

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // ballots for the bits the digit really has (the tile sort's second pass has five: the other three would all be
    // "every lane agrees")
    const int nbits = 32 - __builtin_clz(mask | 1u);
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int li = wbase + k * 64 + lane;
        const bool valid = li < block_n;
        const unsigned d = digit_of<KeyT>(key[k], shift, mask);
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        int r = 0;
        if (valid) {
            r = s_cnt[wid][d] + __popcll(peers & lt_mask);
            // highest peer lane publishes the new count (after every peer has read the old one:
            // LDS operations of one wave complete in program order)
            if ((peers >> lane) == 1ull) s_cnt[wid][d] = r + 1;
        }
        rank[k] = r;
    }
    __syncthreads();

    // thread tid owns digit tid: wave bases, block count, block-local scan, global offset
    {
        int c[kSortWaves];
        int tot = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) { c[w] = s_cnt[w][tid]; }
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) { s_cnt[w][tid] = tot; tot += c[w]; }
        // exclusive scan of `tot` over the 256 digits, and of digit_tot (global)
        int x = tot;
        int g = digit_tot[tid];
        int gx = g;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            const int gy = __shfl_up(gx, o, 64);
            if (lane >= o) { x += y; gx += gy; }
        }
        __shared__ int s_gw[kSortWaves];
        if (lane == 63) { s_wsum[wid] = x; s_gw[wid] = gx; }
        __syncthreads();
        int wb = 0, gwb = 0;
        for (int w = 0; w < wid; ++w) { wb += s_wsum[w]; gwb += s_gw[w]; }
        const int lbase = wb + x - tot;
        int block_prefix;
        if constexpr (LOOKBACK) {
            unsigned* mine = lookback + (long long)tile * kRadix + tid;
            __hip_atomic_store(mine, (unsigned)tot | (tile == 0 ? kLbInclusive : kLbPartial), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            unsigned excl = 0;
            for (int j = tile - 1; j >= 0; --j) {
                const unsigned* theirs = lookback + (long long)j * kRadix + tid;
                unsigned v;
                int spins = 0;
                while (((v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 30) == 0u) {
                    if (++spins > kLbSpinLimit) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if ((v >> 30) == 0u) { status[1] = 1; break; }        // watchdog: never hang the GPU
                excl += v & kLbValue;
                if ((v >> 30) == 2u) break;
            }
            if (tile != 0)
                __hip_atomic_store(mine, (excl + (unsigned)tot) | kLbInclusive, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            block_prefix = (int)excl;
        } else {
            block_prefix = hist[(long long)tid * nblocks_max + blockIdx.x];
        }
        const int gbase = gwb + gx - g + block_prefix;
        s_lbase[tid] = lbase;
        s_gofs[tid] = gbase - lbase;
    }
    __syncthreads();

    // reorder through LDS
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int li = wbase + k * 64 + lane;
        if (li < block_n) {
            const unsigned d = digit_of<KeyT>(key[k], shift, mask);
            const int pos = s_lbase[d] + s_cnt[wid][d] + rank[k];
            s_keys[pos] = key[k];
            s_vals[pos] = val[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int pos = k * kSortThreads + tid;
        if (pos < block_n) {
            const KeyT kk = s_keys[pos];
            const unsigned d = digit_of<KeyT>(kk, shift, mask);
            const long long dst = (long long)s_gofs[d] + pos;
            keys_out[dst] = kk;
            vals_out[dst] = s_vals[pos];
        }
    }
}

// ---- (0) onesweep: global digit histograms of ALL passes in one read of the keys -----------------------
template <typename KeyT, int PASSES>
__global__ void __launch_bounds__(kSortThreads)
sort_global_hist_kernel(const KeyT* __restrict__ keys, const int* __restrict__ n_dev, int end_bit,
                        int* __restrict__ ghist) {
    __shared__ int s_h[PASSES][kRadix];
    const int n = n_dev[0];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) s_h[p][threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kSortThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kSortThreads) {
        const KeyT k = keys[i];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int shift = p * kRadixBits;
            if (shift < end_bit) {
                const int bits = min(kRadixBits, end_bit - shift);
                atomicAdd(&s_h[p][digit_of<KeyT>(k, shift, (1u << bits) - 1u)], 1);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int c = s_h[p][threadIdx.x];
        if (c != 0) atomicAdd(&ghist[p * kRadix + threadIdx.x], c);
    }
}

// Measured on MI355X at M = 4.7e6 (round 1): the decoupled look-back pass takes 33 us against 28 us
// for histogram + scan + scatter (its per-digit look-back walks predecessors one ~1.5 us global load
// at a time), so the three-kernel pass is what the library runs.  Round 4, config D (27.5 M pairs): 121 us per look-back
// pass against 70 for the three kernels, and requesting eight predecessors per round trip + walking only after the LDS
// reorder changed nothing -- with ~1 000 tiles in flight that all started together, nearly every predecessor is itself
// still looking back, so the walk is as long as the number of tiles in flight whatever a hop costs; a build with -DQED_SORT_LOOKBACK selects the
// look-back pass (a compile-time choice: the library reads no environment and keeps no state).
static constexpr bool sort_use_lookback() {
#ifdef QED_SORT_LOOKBACK
    return true;
#else
    return false;
#endif
}

template <typename KeyT>
constexpr int max_passes() { return (int)sizeof(KeyT); }

// workspace: [reduce-then-scan] hist[256][nb] + digit_tot[256]  |  [onesweep] ghist[P][256] + tickets[P(+pad)]
// + lookback[P][nb][256]; sized for whichever is larger
template <typename KeyT, int KPT>
static long long sort_workspace_need(long long capacity) {
    constexpr int kItems = SortCfg<KeyT, KPT>::kItems;
    const long long nb = (capacity + kItems - 1) / kItems;
    const long long a = ((long long)kRadix * nb + kRadix) * 4;
    const long long P = max_passes<KeyT>();
    const long long b = (P * kRadix + 64 + P * nb * kRadix) * 4;
    return (a > b ? a : b) + 256;
}

template <typename KeyT, int KPT>
static int sort_pairs_impl(KeyT* keys, int* vals, KeyT* keys_alt, int* vals_alt, const int* n_dev, long long capacity,
                           int end_bit, void* workspace, long long workspace_bytes, int* status, hipStream_t st) {
    constexpr int kItems = SortCfg<KeyT, KPT>::kItems;
    const int nblocks_max = (int)((capacity + kItems - 1) / kItems);
    if (nblocks_max == 0) return 0;
    const long long need = sort_workspace_need<KeyT, KPT>(capacity);
    if (workspace_bytes < need) {
        set_error("qed_sort_pairs: workspace too small (%lld < %lld)", workspace_bytes, need);
        return QED_E_WORKSPACE;
    }
    int* hist = (int*)workspace;
    int* digit_tot = hist + (long long)kRadix * nblocks_max;
    const int passes = (end_bit + kRadixBits - 1) / kRadixBits;
    KeyT* kin = keys; int* vin = vals; KeyT* kout = keys_alt; int* vout = vals_alt;
    if (sort_use_lookback()) {
        constexpr int P = max_passes<KeyT>();
        int* ghist = (int*)workspace;                        // [P][256]
        int* tickets = ghist + P * kRadix;                   // [64]
        unsigned* lookback = (unsigned*)(tickets + 64);      // [passes][nblocks_max][256]
        const size_t zero_bytes = ((size_t)P * kRadix + 64 + (size_t)passes * nblocks_max * kRadix) * 4;
        if (hipMemsetAsync(workspace, 0, zero_bytes, st) != hipSuccess) {
            set_error("qed_sort_pairs: memset failed");
            return QED_E_LAUNCH;
        }
        const int hist_grid = nblocks_max < 256 ? nblocks_max : 256;
        hipLaunchKernelGGL((sort_global_hist_kernel<KeyT, P>), dim3(hist_grid), dim3(kSortThreads), 0, st, kin, n_dev,
                           end_bit, ghist);
        for (int p = 0; p < passes; ++p) {
            const int shift = p * kRadixBits;
            const int bits = min(kRadixBits, end_bit - shift);
            const unsigned mask = (1u << bits) - 1u;
            hipLaunchKernelGGL((sort_scatter_kernel<KeyT, KPT, true>), dim3(nblocks_max), dim3(kSortThreads), 0, st, kin,
                               vin, kout, vout, n_dev, shift, mask, nblocks_max, (const int*)nullptr,
                               (const int*)(ghist + p * kRadix), lookback + (size_t)p * nblocks_max * kRadix,
                               tickets + p, status);
            KeyT* tk = kin; kin = kout; kout = tk;
            int* tv = vin; vin = vout; vout = tv;
        }
        const int rc = check_launch("qed_sort_pairs");
        if (rc != QED_OK) return rc;
        return passes & 1;
    }
    // the key bits are dealt EVENLY over the passes (13 tile bits: 7 + 6, not 8 + 5): a pass's write runs are
    // (pairs per workgroup / buckets) long, and the pass with the most buckets has the shortest.  Measured: config D's
    // binning 786 -> 776 us, config B's 112 -> 110.5
    const int even_bits = (end_bit + passes - 1) / passes;
    for (int p = 0, shift = 0; p < passes; ++p) {
        const int bits = min(even_bits, end_bit - shift);
        const unsigned mask = (1u << bits) - 1u;
        hipLaunchKernelGGL((sort_hist_kernel<KeyT, KPT>), dim3(nblocks_max), dim3(kSortThreads), 0, st, kin, n_dev,
                           shift, mask, nblocks_max, hist);
        hipLaunchKernelGGL((sort_scan_kernel<kItems>), dim3(kRadix), dim3(kSortThreads), 0, st, n_dev, nblocks_max,
                           hist, digit_tot);
        hipLaunchKernelGGL((sort_scatter_kernel<KeyT, KPT, false>), dim3(nblocks_max), dim3(kSortThreads), 0, st, kin,
                           vin, kout, vout, n_dev, shift, mask, nblocks_max, (const int*)hist, (const int*)digit_tot,
                           (unsigned*)nullptr, (int*)nullptr, (int*)nullptr);
        KeyT* tk = kin; kin = kout; kout = tk;
        int* tv = vin; vin = vout; vout = tv;
        shift += bits;
    }
    const int rc = check_launch("qed_sort_pairs");
    if (rc != QED_OK) return rc;
    return passes & 1;
}

constexpr int kKpt64 = 8;   // 2048 pairs per workgroup: >= 1400 workgroups at M = 3e6
// 32-bit keys: 2048 pairs per workgroup for short lists, 4096 for long ones.  Measured on the tile sort of config B
// (3.35 M keys, two passes; the whole of qed_bin_tiles): 1792 keys per workgroup 151 us, 2048 147, 2560 144, 3072 141,
// 3584 / 4096 139, 5120 140, 6144 157 -- fewer, longer-lived workgroups (one generation instead of 1.07) until the
// 38 KB of LDS per workgroup leaves too few of them per CU.  Short lists keep the small workgroups: 4096 pairs would
// leave half the CUs idle below ~1 M keys.
constexpr int kKpt32 = 8;
constexpr int kKpt32Long = 16;
constexpr long long kSortLongList = 3000000;

// 32-bit-key flavour used by the tile binning (qed_bin_tiles, isect.hip); sized for the smaller workgroups (more of them)
long long sort32_workspace_bytes(long long capacity) { return sort_workspace_need<unsigned, kKpt32>(capacity); }

int sort_pairs_u32(unsigned* keys, int* vals, unsigned* keys_alt, int* vals_alt, const int* n_dev, long long capacity,
                   int end_bit, void* workspace, long long workspace_bytes, int* status, hipStream_t st) {
    if (capacity >= kSortLongList)
        return sort_pairs_impl<unsigned, kKpt32Long>(keys, vals, keys_alt, vals_alt, n_dev, capacity, end_bit, workspace,
                                                     workspace_bytes, status, st);
    return sort_pairs_impl<unsigned, kKpt32>(keys, vals, keys_alt, vals_alt, n_dev, capacity, end_bit, workspace,
                                             workspace_bytes, status, st);
}

}  // namespace qed

using namespace qed;

extern "C" int64_t qed_sort_workspace_bytes(int64_t capacity) {
    if (capacity < 0) return QED_E_INVALID_ARG;
    return sort_workspace_need<unsigned long long, kKpt64>(capacity);
}

extern "C" int qed_sort_pairs(uint64_t* keys, int32_t* vals, uint64_t* keys_alt, int32_t* vals_alt,
                              const int32_t* n_dev, int64_t capacity, int32_t end_bit, void* workspace,
                              int64_t workspace_bytes, int32_t* status, void* stream) {
    QED_REQUIRE(capacity >= 0 && capacity < (1ll << 31), "capacity out of range");
    QED_REQUIRE(end_bit >= 1 && end_bit <= 64, "end_bit must be in [1, 64]");
    QED_REQUIRE(capacity < (1ll << 30), "capacity must be below 2^30 (30-bit look-back words)");
    if (capacity == 0) return 0;
    QED_REQUIRE(keys && vals && keys_alt && vals_alt && n_dev && workspace && status, "null buffers");
    return sort_pairs_impl<unsigned long long, kKpt64>((unsigned long long*)keys, vals, (unsigned long long*)keys_alt,
                                                      vals_alt, n_dev, capacity, end_bit, workspace, workspace_bytes,
                                                      status, (hipStream_t)stream);
}
