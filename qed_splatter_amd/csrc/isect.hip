// K3 + K5: tile intersection (scan + emit) and per-tile offsets of the sorted list.
// Replaces gsplat isect_tiles / isect_offset_encode behind model.py:267-288
// (SURVEY.md Appendix A.4-A.5).  Integer / byte work, HBM-bound.
#include "qed_common.h"

namespace qed {

// ---- exclusive scan of block_sums (one workgroup; n_blocks is N/256, i.e. thousands) -----------
// Every thread owns kScanPer CONSECUTIVE sums, requested together (one memory round trip per 32 768 sums: the first
// version walked the array in 1 024-wide slices, one dependent round trip and three barriers each -- 31 us for the
// 19 532 sums of 5 M slots, 7 us now), scans them in registers, and the 1 024 thread totals are scanned across the
// workgroup.
constexpr int kScanPer = 32;
__global__ void __launch_bounds__(1024)
isect_scan_kernel(const int* __restrict__ block_sums, int n_blocks, int* __restrict__ block_offsets,
                  int* __restrict__ n_isect, long long capacity, int* __restrict__ status) {
    __shared__ long long wave_tot[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024 * kScanPer) {
        const int i0 = base + tid * kScanPer;
        int v[kScanPer];
        const bool vec = (((uintptr_t)block_sums | (uintptr_t)block_offsets) & 15) == 0;
        if (vec && i0 + kScanPer <= n_blocks) {                    // eight 16-byte loads instead of thirty-two scalar ones
#pragma unroll
            for (int j = 0; j < kScanPer / 4; ++j) {
                const int4 q = reinterpret_cast<const int4*>(block_sums + i0)[j];
                v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kScanPer; ++j) v[j] = block_sums[i0 + j < n_blocks ? i0 + j : n_blocks - 1];
        }
        long long mine = 0;
#pragma unroll
        for (int j = 0; j < kScanPer; ++j) mine += i0 + j < n_blocks ? v[j] : 0;
        // inclusive scan of the thread totals within the wave, then across the 16 waves
        long long x = mine;
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
        long long run = carry + wbase + x - mine;             // exclusive prefix of this thread's first sum
        if (vec && i0 + kScanPer <= n_blocks) {
#pragma unroll
            for (int j = 0; j < kScanPer / 4; ++j) {
                int4 q;
                q.x = (int)min(run, (long long)0x7fffffff); run += v[4 * j];
                q.y = (int)min(run, (long long)0x7fffffff); run += v[4 * j + 1];
                q.z = (int)min(run, (long long)0x7fffffff); run += v[4 * j + 2];
                q.w = (int)min(run, (long long)0x7fffffff); run += v[4 * j + 3];
                reinterpret_cast<int4*>(block_offsets + i0)[j] = q;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kScanPer; ++j) {
                if (i0 + j < n_blocks) {
                    block_offsets[i0 + j] = (int)min(run, (long long)0x7fffffff);
                    run += v[j];
                }
            }
        }
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
// KeyT = u64: key = (cam|tile) << 32 | depth bits (one-stage sort).  KeyT = u32: key = cam|tile only;
// then `order` lists the slots in depth order, so emission order already carries the depth order
// (two-stage binning, qed_bin_tiles).
// SELF_SCAN (qed_bin_tiles): block_offsets holds the per-block SUMS and every workgroup adds up its predecessors'
// itself (a few thousand L2-resident ints) -- one launch and one single-workgroup serial scan less; the last workgroup
// publishes M (or the overflow) in n_isect / status, which only later kernels read.
template <typename KeyT, bool SELF_SCAN>
__global__ void __launch_bounds__(256)
isect_emit_kernel(int N, int C, const float* __restrict__ means2d, const int* __restrict__ radii,
                  const float* __restrict__ depths, const int* __restrict__ tiles_per_gauss,
                  const int* __restrict__ block_offsets, int tile_w, int tile_h, int tile_bits,
                  int* __restrict__ n_isect, const int* __restrict__ order, KeyT* __restrict__ keys,
                  int* __restrict__ vals, const float* __restrict__ splats, long long capacity,
                  int* __restrict__ status, const unsigned long long* __restrict__ tile_masks = nullptr,
                  unsigned slot_mask = 0xFFFFFFFFu) {
    __shared__ int s_pref[4][65];   // per wave: exclusive prefix of counts (+ total)
    __shared__ int s_w[4][64], s_tile0[4][64], s_nz[4][64];
    __shared__ float s_invw[4][64];
    __shared__ unsigned long long s_ctb[4][64];    // camera bits of the key
    __shared__ unsigned long long s_mask[4][64];   // qed_project_fwd's tile_masks: which tiles of the rectangle are listed
    __shared__ unsigned long long s_start[4][64];  // bit j: a Gaussian's entries start at entry j of the wave
    __shared__ unsigned s_depth[4][64];
    __shared__ int s_slot[4][64];
    __shared__ int s_wave_tot[4];
    __shared__ long long s_scan[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int block_base = 0;
    if constexpr (SELF_SCAN) {
        long long before = 0, all = 0;
        for (int i0 = tid; i0 < (int)gridDim.x; i0 += 8 * 256) {          // eight sums in flight per trip
            int v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = block_offsets[i0 + 256 * j < (int)gridDim.x ? i0 + 256 * j : 0];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + 256 * j;
                if (i < (int)gridDim.x) {
                    all += v[j];
                    if (i < (int)blockIdx.x) before += v[j];
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); all += __shfl_xor(all, o, 64); }
        if (lane == 0) { s_scan[0][wid] = before; s_scan[1][wid] = all; }
        __syncthreads();
        before = s_scan[0][0] + s_scan[0][1] + s_scan[0][2] + s_scan[0][3];
        all = s_scan[1][0] + s_scan[1][1] + s_scan[1][2] + s_scan[1][3];
        const bool overflow = all > capacity || all > 0x7fffffffll;
        if (blockIdx.x == gridDim.x - 1 && tid == 0) {
            if (overflow) { status[0] = (int)min(all, (long long)0x7fffffff); n_isect[0] = 0; }
            else n_isect[0] = (int)all;
        }
        if (overflow || all == 0) return;
        block_base = (int)before;
    } else {
        if (n_isect[0] == 0) return;    // nothing to do (or capacity exceeded)
        block_base = block_offsets[blockIdx.x];
    }
    const long long total = (long long)C * N;
    const long long pos = (long long)blockIdx.x * 256 + tid;
    long long slot = pos;
    // (two-stage binning: the depth-ordered values carry the tile count above the slot bits, see depth_keys_kernel)
    if (order != nullptr && pos < total) slot = (long long)((unsigned)order[pos] & slot_mask);
    int cnt = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    unsigned dbits = 0;
    unsigned long long tmask = ~0ull;
    if (pos < total) {
        if (tile_masks != nullptr) {
            // qed_project_fwd's 16-byte descriptor: count, packed rectangle, mask of the listed tiles -- one gather
            const uint4 d = reinterpret_cast<const uint4*>(tile_masks)[slot];
            cnt = (int)d.x;
            x0 = (int)(d.y & 2047u); y0 = (int)((d.y >> 11) & 2047u); x1 = x0 + (int)(d.y >> 22);
            tmask = ((unsigned long long)d.w << 32) | d.z;
            if constexpr (sizeof(KeyT) == 8) dbits = __float_as_uint(depths[slot]);
        } else if (splats != nullptr) {
            // the rectangle project_fwd counted (QED_F_TIGHT_TILES or not), packed in record slot 11.  Fetched
            // beside the count, not behind it: both gathers depend on `slot` only (a rectangle read under
            // `cnt > 0` is a third serialised random access per Gaussian)
            const unsigned r = __float_as_uint(splats[(size_t)slot * QED_SPLAT_FLOATS + 11]);
            cnt = tiles_per_gauss[slot];
            x0 = (int)(r & 2047u); y0 = (int)((r >> 11) & 2047u); x1 = x0 + (int)(r >> 22);
            if constexpr (sizeof(KeyT) == 8) dbits = __float_as_uint(depths[slot]);
        } else {
            cnt = tiles_per_gauss[slot];
            if (cnt > 0) {
                tile_rect(means2d[2 * slot], means2d[2 * slot + 1], (float)radii[slot], tile_w, tile_h, x0, y0, x1, y1);
                dbits = __float_as_uint(depths[slot]);
            }
        }
    }
    // wave-inclusive scan of the counts
    int x = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    const int pref = x - cnt;
    s_pref[wid][lane] = pref;
    if (lane == 63) { s_pref[wid][64] = x; s_wave_tot[wid] = x; }
    // per Gaussian, once: everything of an entry's key but the tile's place inside the rectangle (the per-entry integer
    // divisions by the rectangle's width and by N used to be two thirds of this kernel's vector instructions)
    const int rw = x1 - x0;
    s_w[wid][lane] = rw;
    s_invw[wid][lane] = __builtin_amdgcn_rcpf((float)max(rw, 1));
    s_tile0[wid][lane] = y0 * tile_w + x0;
    s_ctb[wid][lane] = (unsigned long long)(C == 1 ? 0 : slot / N) << tile_bits;
    s_depth[wid][lane] = dbits;
    s_slot[wid][lane] = (int)slot;
    s_mask[wid][lane] = tmask;
    s_start[wid][lane] = 0ull;
    const unsigned long long nz = __ballot(cnt > 0);
    if (cnt > 0) s_nz[wid][__popcll(nz & ((1ull << lane) - 1ull))] = lane;
    __syncthreads();
    int wave_base = block_base;
    for (int w = 0; w < wid; ++w) wave_base += s_wave_tot[w];
    const int wtot = s_pref[wid][64];
    // Entry j of the wave belongs to the Gaussian whose range started last at or before j.  Up to 4096 entries per wave
    // (nearly always) that is a popcount over one LDS word of range-start bits per trip (as in project_fwd_kernel's
    // mask pass); beyond, a binary search over the prefix sums.
    const bool by_bits = wtot <= 64 * 64;
    if (by_bits && cnt > 0) atomicOr(&s_start[wid][pref >> 6], 1ull << (pref & 63));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int before = 0;
    for (int j0 = 0; j0 < wtot; j0 += 64) {
        const int j = j0 + lane;
        int g;
        if (by_bits) {
            const unsigned long long word = s_start[wid][j0 >> 6];
            const int k = before + __popcll(word & ((2ull << lane) - 1ull)) - 1;
            before += __popcll(word);
            if (j >= wtot) continue;
            g = s_nz[wid][k];
        } else {
            if (j >= wtot) continue;
            int lo = 0, hi = 63;                              // largest g with pref[g] <= j
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_pref[wid][mid] <= j) lo = mid; else hi = mid - 1;
            }
            g = lo;
        }
        int local = j - s_pref[wid][g];
        // exact tile lists: the entry is the local-th LISTED tile of the rectangle (mask ~0: every tile is listed, and
        // the rectangle may hold more than 64)
        const unsigned long long tm = s_mask[wid][g];
        if (tm != ~0ull) local = nth_set_bit(tm, local);
        // row = local / w through the reciprocal, put right by one step either way (local < 2^21, w <= 1023)
        const int w = s_w[wid][g];
        int row = (int)(((float)local + 0.5f) * s_invw[wid][g]);
        int col = local - row * w;
        if (col < 0) { --row; col += w; } else if (col >= w) { ++row; col -= w; }
        const unsigned long long ct = s_ctb[wid][g] | (unsigned long long)(s_tile0[wid][g] + row * tile_w + col);
        if constexpr (sizeof(KeyT) == 8) keys[wave_base + j] = (KeyT)((ct << 32) | (unsigned long long)s_depth[wid][g]);
        else keys[wave_base + j] = (KeyT)ct;
        vals[wave_base + j] = s_slot[wid][g];
    }
}

// ---- tile offsets ------------------------------------------------------------------------------
// Four consecutive keys per thread and trip, grid-stride: one thread per key meant 15 600 workgroups of one dependent
// load each at config B -- eight generations of pure latency, 9 us for 13 MB.
template <typename KeyT>
__global__ void __launch_bounds__(256)
tile_offsets_kernel(const KeyT* __restrict__ keys, const int* __restrict__ n_dev, int n_tiles_total,
                    int n_tiles, int tile_bits, int* __restrict__ offsets, const int* __restrict__ status = nullptr,
                    int* __restrict__ host_words = nullptr) {
    const int n = n_dev[0];
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, nt = (long long)gridDim.x * 256;
    // qed_bin_tiles' host_words: {M, overflow word, sort watchdog word, 0} straight into host-mapped memory, as ONE
    // 16-byte store (a reader that sees word 0 change sees all four) -- every kernel that sets one of them has retired
    // by now, and the caller is spared a copy and an event in the stream (a blit kernel and two barriers: ~10 us
    // between the binning and the compositing pass)
    if (host_words != nullptr && t0 == 0) {
        *reinterpret_cast<int4*>(host_words) = make_int4(n, status[0], status[1], 0);
        __threadfence_system();              // (pushed out to the host now, not when the kernel ends)
    }
    if (n == 0) {
        for (long long i = t0; i <= n_tiles_total; i += nt) offsets[i] = 0;
        return;
    }
    auto lin = [&](KeyT k) -> int {
        const unsigned long long ct = sizeof(KeyT) == 8 ? ((unsigned long long)k >> 32) : (unsigned long long)k;
        return (int)((ct >> tile_bits) * (unsigned long long)n_tiles + (ct & ((1ull << tile_bits) - 1ull)));
    };
    for (long long base = 4 * t0; base < n; base += 4 * nt) {
        KeyT k[4];
        const KeyT kp = keys[base > 0 ? base - 1 : 0];
        if (sizeof(KeyT) == 4 && base + 3 < n) {
            // one 16-byte load for the four keys (base is a multiple of 4, the array 16-byte aligned)
            const uint4 q = *reinterpret_cast<const uint4*>(keys + base);
            k[0] = (KeyT)q.x; k[1] = (KeyT)q.y; k[2] = (KeyT)q.z; k[3] = (KeyT)q.w;
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) k[v] = keys[base + v < n ? base + v : n - 1];
        }
        int prev = base > 0 ? lin(kp) : -1;                 // (tile -1: everything up to the first key's tile starts at 0)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const long long i = base + v;
            if (i >= n) break;
            const int cur = lin(k[v]);
            for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int)i;
            if (i == n - 1)
                for (int t = cur + 1; t <= n_tiles_total; ++t) offsets[t] = n;
            prev = cur;
        }
    }
}

// ---- two-stage binning helpers (qed_bin_tiles) ----------------------------------------------------
// stage A input: key = depth bits of visible slots (0xFFFFFFFF sorts culled slots last), value = slot -- with the
// slot's tile count riding in the bits above it (slot_bits = bits of the slot index; counts that do not fit are stored as
// the all-ones marker and looked up): the count then arrives in depth order WITH the sorted values, and summing it per
// block is a coalesced read instead of a random 4-byte gather per slot (71 -> 12 us at 5 M slots).  slot_bits = 32: no
// room, plain slots.
__global__ void __launch_bounds__(256)
depth_keys_kernel(int n_slots, const int* __restrict__ radii, const float* __restrict__ depths,
                  unsigned* __restrict__ keys, int* __restrict__ vals, int* __restrict__ n_slots_dev,
                  const int* __restrict__ tiles_per_gauss, int slot_bits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) n_slots_dev[0] = n_slots;
    if (i < n_slots) {
        keys[i] = radii[i] > 0 ? __float_as_uint(depths[i]) : 0xFFFFFFFFu;
        unsigned v = (unsigned)i;
        if (slot_bits < 32) {
            const unsigned marker = (1u << (32 - slot_bits)) - 1u;
            v |= min((unsigned)tiles_per_gauss[i], marker) << slot_bits;
        }
        vals[i] = (int)v;
    }
}

// tile counts of the depth-ordered slots, summed per 256 (input of the intersection scan)
__global__ void __launch_bounds__(256)
count_sorted_kernel(int n_slots, const int* __restrict__ order, const int* __restrict__ tiles_per_gauss,
                    int* __restrict__ block_sums, int slot_bits = 32) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int v = 0;
    if (i < n_slots) {
        if (order == nullptr) {
            v = tiles_per_gauss[i];
        } else if (slot_bits >= 32) {
            v = tiles_per_gauss[order[i]];
        } else {
            const unsigned o = (unsigned)order[i], marker = (1u << (32 - slot_bits)) - 1u;
            const unsigned c = o >> slot_bits;
            v = c == marker ? tiles_per_gauss[o & ((1u << slot_bits) - 1u)] : (int)c;
        }
    }
    __shared__ int wsum[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}


// ---- per-tile depth sort (qed_bin_tiles, mode QED_BIN_TILE_SORT) ------------------------------------------------
// After a STABLE sort of the list on the (camera|tile) bits alone -- entries emitted in slot order -- every tile's run
// holds its Gaussians in slot order.  One workgroup per tile then sorts its run by the 32 depth bits with a stable LSD
// radix sort (ties stay in slot order): the same list as sorting 64-bit (tile, depth) keys, without ever sorting the
// C N slots globally (12 launch-latency-bound launches at config B).  Runs of up to kTileSortItems entries are sorted
// entirely in LDS with the ranking scheme of sort_scatter_kernel (wave-striped keys in registers, 64-bit ballot
// match, per-wave counters); longer runs go through global scratch, 256 entries at a time -- slow, but any length is
// handled.  A pass whose digit is the same for the whole run (typical of the high depth byte) is skipped.
constexpr int kTileSortKpt = 8;
constexpr int kTileSortItems = 256 * kTileSortKpt;

// stable rank of `valid` lanes' digit d within the wave, on top of the wave's running counters cnt[256] (LDS)
__device__ __forceinline__ int wave_digit_rank(bool valid, unsigned d, int* cnt, int lane) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    int r = 0;
    if (valid) {
        r = cnt[d] + __popcll(peers & ((1ull << lane) - 1ull));
        if ((peers >> lane) == 1ull) cnt[d] = r + 1;      // highest peer publishes (LDS ops of a wave complete in order)
    }
    return r;
}


// Short runs (<= kTileWaveItems entries: nearly every tile at config B): ONE WAVE per tile, four tiles per workgroup,
// no workgroup barrier anywhere, a whole generation of tiles resident at once.  The keys stay in registers (8 rows of
// 64).  A run is a few hundred depths, so a full 4-pass radix sort is overkill:
//   (1) ONE counting pass on a 9-bit bucket  b = floor((key - kmin) * 511.99 / (kmax - kmin))  -- monotone in the key,
//       places inside the bucket handed out by returning LDS atomics, the 512 bucket totals scanned by the wave itself
//       (8 per lane);
//   (2) the run is now ordered by bucket and two keys can only be out of order inside one bucket, so every entry
//       counts the members of its own bucket that sort before it (smaller depth, or equal depth and earlier slot:
//       stable) and that count is its place -- a few LDS reads per entry for a spread of depths.
// Runs with a crowded bucket (> kTileBucketMax) take four stable 8-bit radix passes instead, like the long runs that
// are left to tile_depth_sort_kernel.
constexpr int kTileWaveKpt = 8;
constexpr int kTileWaveItems = 64 * kTileWaveKpt;
constexpr int kTileBuckets = 512;
constexpr int kTileBucketMax = 12;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one wave sorts the run [start, start + n), 1 <= n <= kTileWaveItems; kv / cnt: this wave's LDS
__device__ __forceinline__ void tile_sort_wave(int start, int n, int lane, uint2* __restrict__ kv, int* __restrict__ cnt,
                                               const int* __restrict__ vals_in, const float* __restrict__ depths,
                                               int* __restrict__ vals_out) {
    if (n == 1) {
        if (lane == 0) vals_out[start] = vals_in[start];
        return;
    }
    const int rows = (n + 63) >> 6;                        // wave-uniform
    unsigned key[kTileWaveKpt];
    int val[kTileWaveKpt], rank[kTileWaveKpt];
#pragma unroll
    for (int k = 0; k < kTileWaveKpt; ++k) {
        if (k >= rows) break;
        const int li = k * 64 + lane;
        val[k] = vals_in[start + (li < n ? li : n - 1)];
    }
    unsigned kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < kTileWaveKpt; ++k) {
        if (k >= rows) break;
        key[k] = __float_as_uint(depths[val[k]]);          // (lanes past the end hold a copy of the last entry)
        kmin = min(kmin, key[k]);
        kmax = max(kmax, key[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
    }
    if (kmin == kmax) {                                    // one depth: the slot order is the answer
#pragma unroll
        for (int k = 0; k < kTileWaveKpt; ++k) {
            if (k >= rows) break;
            if (k * 64 + lane < n) vals_out[start + k * 64 + lane] = val[k];
        }
        return;
    }
    // ---- (1) stable counting pass on the bucket ----
    const float scale = 511.99f / (float)(kmax - kmin);
    unsigned bkt[kTileWaveKpt];
    *reinterpret_cast<int4*>(&cnt[8 * lane]) = make_int4(0, 0, 0, 0);
    *reinterpret_cast<int4*>(&cnt[8 * lane + 4]) = make_int4(0, 0, 0, 0);
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < kTileWaveKpt; ++k) {
        if (k >= rows) break;
        bkt[k] = min((unsigned)((float)(key[k] - kmin) * scale), (unsigned)(kTileBuckets - 1));
        // the entry's place among its bucket's members as they ARRIVE (one returning LDS atomic; the 9-ballot match that
        // gave the stable rank was ~60 instructions per row): step (2) orders a bucket's members by (depth, slot) whatever
        // order they were stored in
        rank[k] = k * 64 + lane < n ? atomicAdd(&cnt[bkt[k]], 1) : 0;
    }
    wave_lds_fence();
    int largest;
    {   // lane l owns buckets 8 l .. 8 l + 7: totals -> exclusive bases
        const int4 c0 = *reinterpret_cast<const int4*>(&cnt[8 * lane]);
        const int4 c1 = *reinterpret_cast<const int4*>(&cnt[8 * lane + 4]);
        largest = max(max(max(c0.x, c0.y), max(c0.z, c0.w)), max(max(c1.x, c1.y), max(c1.z, c1.w)));
        const int tot = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
        int x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) largest = max(largest, __shfl_xor(largest, o, 64));
        int b = x - tot;
        int4 e0, e1;
        e0.x = b; b += c0.x; e0.y = b; b += c0.y; e0.z = b; b += c0.z; e0.w = b; b += c0.w;
        e1.x = b; b += c1.x; e1.y = b; b += c1.y; e1.z = b; b += c1.z; e1.w = b;
        *reinterpret_cast<int4*>(&cnt[8 * lane]) = e0;
        *reinterpret_cast<int4*>(&cnt[8 * lane + 4]) = e1;
    }
    wave_lds_fence();
    if (largest <= kTileBucketMax) {
#pragma unroll
        for (int k = 0; k < kTileWaveKpt; ++k) {
            if (k >= rows) break;
            if (k * 64 + lane < n) kv[cnt[bkt[k]] + rank[k]] = make_uint2(key[k], (unsigned)val[k]);
        }
        wave_lds_fence();
        // ---- (2) place every entry inside its bucket: its offset there is the number of bucket members that sort
        // before it (smaller depth, or the same depth and a smaller slot: the run arrives in slot order, so that IS the
        // stable order).  Buckets hold a handful
        // of entries, so this is a few LDS reads per entry and ONE pass (an odd-even transposition took 3-5 rounds of
        // two fenced phases each) ----
#pragma unroll
        for (int k = 0; k < kTileWaveKpt; ++k) {
            if (k >= rows) break;
            const int li = k * 64 + lane;
            if (li < n) {
                const int b0 = cnt[bkt[k]];
                const int b1 = bkt[k] + 1 < kTileBuckets ? cnt[bkt[k] + 1] : n;
                int before = 0;
                for (int j = b0; j < b1; ++j) {
                    const unsigned other = kv[j].x;
                    before += (other < key[k]) | ((other == key[k]) & ((int)kv[j].y < val[k]));
                }
                vals_out[start + b0 + before] = val[k];
            }
        }
        return;
    }
    // ---- crowded bucket: four stable 8-bit radix passes (counters: the first 256 of cnt) ----
    unsigned* skeys = reinterpret_cast<unsigned*>(kv);
    int* svals = reinterpret_cast<int*>(kv) + kTileWaveItems;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        *reinterpret_cast<int4*>(&cnt[4 * lane]) = make_int4(0, 0, 0, 0);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < kTileWaveKpt; ++k) {
            if (k >= rows) break;
            rank[k] = wave_digit_rank(k * 64 + lane < n, (key[k] >> shift) & 255u, cnt, lane);
        }
        wave_lds_fence();
        // lane l owns digits 4 l .. 4 l + 3: totals -> exclusive bases
        const int4 c = *reinterpret_cast<const int4*>(&cnt[4 * lane]);
        const bool skip = __ballot(c.x == n || c.y == n || c.z == n || c.w == n) != 0ull;   // one digit holds the run
        const bool last = pass == 3;
        if (skip && !last) continue;
        const int tot = c.x + c.y + c.z + c.w;
        int x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        const int b = x - tot;
        *reinterpret_cast<int4*>(&cnt[4 * lane]) = make_int4(b, b + c.x, b + c.x + c.y, b + c.x + c.y + c.z);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < kTileWaveKpt; ++k) {
            if (k >= rows) break;
            const int li = k * 64 + lane;
            if (li < n) {
                const int pos = skip ? li : cnt[(key[k] >> shift) & 255u] + rank[k];
                if (last) vals_out[start + pos] = val[k];
                else { skeys[pos] = key[k]; svals[pos] = val[k]; }
            }
        }
        if (last) break;
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < kTileWaveKpt; ++k) {
            if (k >= rows) break;
            const int li = k * 64 + lane;
            if (li < n) { key[k] = skeys[li]; val[k] = svals[li]; }
        }
        wave_lds_fence();
    }
}

// a 256-thread workgroup sorts the run [start, start + n), n > kTileWaveItems: in LDS up to kTileSortItems entries, through
// global scratch beyond
__device__ __forceinline__ void tile_sort_long_run(int start, int n, const int* __restrict__ vals_in,
                                                   const float* __restrict__ depths, int* __restrict__ vals_out,
                                                   unsigned* __restrict__ gk0, unsigned* __restrict__ gk1,
                                                   int* __restrict__ gv0, int* __restrict__ gv1, unsigned* s_keys,
                                                   int* s_vals, int (*s_cnt)[256], int* s_base, int* s_wsum, int* s_skip) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // exclusive scan over the 256 digits of per-digit totals held one per thread; returns this digit's base
    auto digit_scan = [&](int tot) {
        int x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) s_wsum[wid] = x;
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wid; ++w) wb += s_wsum[w];
        return wb + x - tot;
    };
    if (n <= kTileSortItems) {
        // ---- LDS path: the run is dealt to the four waves as contiguous quarters of Q entries (a multiple of 64):
        // wave w owns [w Q, (w + 1) Q), item k of lane l = w Q + 64 k + l; every wave works, whatever the run length ----
        unsigned key[kTileSortKpt];
        int val[kTileSortKpt], rank[kTileSortKpt];
        const int Q = ((n + 255) >> 8) << 6;
        const int rows = Q >> 6;                            // rows of 64 per wave (block-uniform, <= kTileSortKpt)
        const int wbase = wid * Q;
        auto mine = [&](int k) { return wbase + 64 * k + lane < n; };
        if (tid < 4) s_skip[tid] = 0;
#pragma unroll
        for (int k = 0; k < kTileSortKpt; ++k) {
            if (k >= rows) break;
            const int li = wbase + k * 64 + lane;
            val[k] = vals_in[start + (li < n ? li : n - 1)];
        }
#pragma unroll
        for (int k = 0; k < kTileSortKpt; ++k) {
            if (k >= rows) break;
            key[k] = __float_as_uint(depths[val[k]]);
        }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 8 * pass;
#pragma unroll
            for (int w = 0; w < 4; ++w) s_cnt[w][tid] = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kTileSortKpt; ++k) {
                if (k >= rows) break;
                rank[k] = wave_digit_rank(mine(k), (key[k] >> shift) & 255u, s_cnt[wid], lane);
            }
            __syncthreads();
            {   // thread tid owns digit tid: per-wave bases, run-local exclusive scan
                int c[4], tot = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) c[w] = s_cnt[w][tid];
#pragma unroll
                for (int w = 0; w < 4; ++w) { s_cnt[w][tid] = tot; tot += c[w]; }
                if (tot == n) s_skip[pass] = 1;            // one digit holds the whole run: this pass is the identity
                s_base[tid] = digit_scan(tot);
            }
            __syncthreads();
            const bool skip = s_skip[pass] != 0;           // (block-uniform)
            const bool last = pass == 3;
            if (skip && !last) continue;
#pragma unroll
            for (int k = 0; k < kTileSortKpt; ++k) {
                if (k >= rows) break;
                if (mine(k)) {
                    const unsigned d = (key[k] >> shift) & 255u;
                    const int pos = skip ? wbase + k * 64 + lane : s_base[d] + s_cnt[wid][d] + rank[k];
                    if (last) vals_out[start + pos] = val[k];
                    else { s_keys[pos] = key[k]; s_vals[pos] = val[k]; }
                }
            }
            if (last) break;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kTileSortKpt; ++k) {
                if (k >= rows) break;
                if (mine(k)) { key[k] = s_keys[wbase + k * 64 + lane]; val[k] = s_vals[wbase + k * 64 + lane]; }
            }
        }
        return;
    }
    // ---- long runs: the same passes through global scratch, one row of 256 entries at a time ----
    const unsigned* ksrc = nullptr;        // pass 0 gathers the depth bits
    const int* vsrc = vals_in;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        unsigned* kdst = (pass & 1) ? gk1 : gk0;
        int* vdst = pass == 3 ? vals_out : ((pass & 1) ? gv1 : gv0);
        s_base[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const unsigned kk = ksrc ? ksrc[start + i] : __float_as_uint(depths[vsrc[start + i]]);
            atomicAdd(&s_base[(kk >> shift) & 255u], 1);
        }
        __syncthreads();
        const int tot = s_base[tid];
        const int base = digit_scan(tot);
        __syncthreads();
        s_base[tid] = base;                                 // running cursor of digit tid
        __syncthreads();
        for (int row = 0; row < n; row += 256) {
            const int i = row + tid;
            const bool valid = i < n;
            const int vv = valid ? vsrc[start + i] : 0;
            const unsigned kk = valid ? (ksrc ? ksrc[start + i] : __float_as_uint(depths[vv])) : 0u;
            const unsigned d = (kk >> shift) & 255u;
#pragma unroll
            for (int w = 0; w < 4; ++w) s_cnt[w][tid] = 0;
            __syncthreads();
            const int r = wave_digit_rank(valid, d, s_cnt[wid], lane);
            __syncthreads();
            {   // digit tid: counts of the four waves -> exclusive per-wave bases; advance the cursor
                int c[4], t2 = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) c[w] = s_cnt[w][tid];
                const int cur = s_base[tid];
#pragma unroll
                for (int w = 0; w < 4; ++w) { s_cnt[w][tid] = cur + t2; t2 += c[w]; }
                __syncthreads();
                s_base[tid] = cur + t2;
            }
            __syncthreads();
            if (valid) {
                const int pos = s_cnt[wid][d] + r;
                vdst[start + pos] = vv;
                if (pass < 3) kdst[start + pos] = kk;
            }
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        ksrc = kdst;
        vsrc = vdst;
    }
}

// One workgroup = four tiles.  Runs of <= kTileWaveItems entries (nearly all at config B): one wave each, no workgroup
// barrier.  Longer runs among the four: afterwards, by all 256 threads together, in the same LDS.
__global__ void __launch_bounds__(256)
tile_depth_sort_kernel(const int* __restrict__ offsets, const int* __restrict__ vals_in, const float* __restrict__ depths,
                       int* __restrict__ vals_out, unsigned* __restrict__ gk0, unsigned* __restrict__ gk1,
                       int* __restrict__ gv0, int* __restrict__ gv1, int n_tiles_total) {
    __shared__ __attribute__((aligned(16))) uint2 s_kv[4][kTileWaveItems];      // (key, value) pairs of a run
    __shared__ __attribute__((aligned(16))) int s_cnt[4][kTileBuckets];
    __shared__ int s_wsum[4], s_skip[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wid;
    int start = 0, n = 0;
    if (tile < n_tiles_total) { start = offsets[tile]; n = offsets[tile + 1] - start; }
    if (n >= 1 && n <= kTileWaveItems) tile_sort_wave(start, n, lane, s_kv[wid], s_cnt[wid], vals_in, depths, vals_out);
    if (!__syncthreads_or(n > kTileWaveItems)) return;
    static_assert(sizeof(s_kv) >= kTileSortItems * 8 && sizeof(s_cnt) >= (4 * 256 + 256) * 4, "LDS aliasing");
    unsigned* keys = reinterpret_cast<unsigned*>(&s_kv[0][0]);
    int* vals = reinterpret_cast<int*>(keys + kTileSortItems);
    int (*cnt)[256] = reinterpret_cast<int (*)[256]>(&s_cnt[0][0]);
    int* base = &s_cnt[0][0] + 4 * 256;
    for (int j = 0; j < 4; ++j) {
        const int t = blockIdx.x * 4 + j;
        if (t >= n_tiles_total) break;
        const int s0 = offsets[t], m = offsets[t + 1] - s0;
        if (m <= kTileWaveItems) continue;
        __syncthreads();
        tile_sort_long_run(s0, m, vals_in, depths, vals_out, gk0, gk1, gv0, gv1, keys, vals, cnt, base, s_wsum, s_skip);
    }
}

// gsplat-style 64-bit keys of the sorted list (info["isect_ids"]), rebuilt on demand
__global__ void __launch_bounds__(256)
isect_ids_kernel(const unsigned* __restrict__ tile_keys, const int* __restrict__ flatten_ids,
                 const float* __restrict__ depths, const int* __restrict__ n_dev, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_dev[0])
        out[i] = ((unsigned long long)tile_keys[i] << 32) | (unsigned long long)__float_as_uint(depths[flatten_ids[i]]);
}

}  // namespace qed

using namespace qed;

static long long align256(long long x) { return (x + 255) & ~255ll; }

// four keys per thread, at most 2048 workgroups (one generation at eight per CU), grid-stride beyond
static unsigned tile_offsets_grid(long long work) {
    long long g = (work / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    return (unsigned)(g < 1 ? 1 : g);
}

struct BinLayout {
    long long n_slots_dev, keysA0, keysA1, valsA0, valsA1, block_sums, block_offsets, keysB0, keysB1, valsB, sort_ws,
        tk1, tv0, tv1, total;
    long long sort_ws_bytes;
};

static BinLayout bin_layout(long long S, long long cap) {
    BinLayout L;
    long long o = 0;
    const long long nblk = (S + 255) / 256 + 1;
    L.n_slots_dev = o; o += 256;
    L.keysA0 = o; o += align256(4 * S);
    L.keysA1 = o; o += align256(4 * S);
    L.valsA0 = o; o += align256(4 * S);
    L.valsA1 = o; o += align256(4 * S);
    L.block_sums = o; o += align256(4 * nblk);
    L.block_offsets = o; o += align256(4 * nblk);
    L.keysB0 = o; o += align256(4 * cap);
    L.keysB1 = o; o += align256(4 * cap);
    L.valsB = o; o += align256(4 * cap);
    L.sort_ws_bytes = sort32_workspace_bytes(S > cap ? S : cap);
    L.sort_ws = o; o += align256(L.sort_ws_bytes);
    // tile-sort mode: scratch of the long-run path of tile_depth_sort_kernel (its other key buffer is keysB's spare)
    L.tk1 = o; o += align256(4 * cap);
    L.tv0 = o; o += align256(4 * cap);
    L.tv1 = o; o += align256(4 * cap);
    L.total = o;
    return L;
}

extern "C" int64_t qed_bin_workspace_bytes(int64_t n_slots, int64_t capacity) {
    if (n_slots < 0 || capacity < 0) return QED_E_INVALID_ARG;
    return bin_layout(n_slots, capacity).total;
}

extern "C" int qed_bin_tiles(int32_t N, int32_t C, const float* means2d, const int32_t* radii, const float* depths,
                             const int32_t* tiles_per_gauss, const float* splats, const uint64_t* tile_masks,
                             const int32_t* block_sums_in, int32_t tile_w, int32_t tile_h, int64_t capacity,
                             int32_t mode, int32_t* flatten_ids,
                             int32_t* offsets, int32_t* n_isect, uint64_t* isect_ids, void* workspace,
                             int64_t workspace_bytes, int32_t* status, int32_t* host_words, void* stream) {
    QED_REQUIRE(N >= 0 && C >= 1 && tile_w > 0 && tile_h > 0, "bad extents");
    QED_REQUIRE(splats == nullptr || (tile_w <= 1023 && tile_h <= 2047), "packed tile rectangles need tile_w <= 1023");
    QED_REQUIRE(tile_masks == nullptr || splats != nullptr, "tile_masks index the rectangles of the splat records");
    QED_REQUIRE(((uintptr_t)tile_masks & 15) == 0, "tile_masks must be 16-byte aligned");
    QED_REQUIRE(capacity >= 0 && capacity < (1ll << 30), "capacity out of range");
    QED_REQUIRE(offsets && n_isect && status && workspace, "null buffers");
    QED_REQUIRE(((uintptr_t)host_words & 15) == 0, "host_words must be 16-byte aligned");
    const long long S = (long long)C * N;
    QED_REQUIRE(S < (1ll << 30), "too many (camera, Gaussian) slots");
    const long long n_tot = (long long)C * tile_w * tile_h;
    int tile_bits = 1;
    while ((1ll << tile_bits) <= (long long)tile_w * tile_h) ++tile_bits;      // floor(log2 T) + 1
    int cam_bits = 0;
    while ((1 << cam_bits) < C) ++cam_bits;
    QED_REQUIRE(tile_bits + cam_bits <= 32, "camera/tile key does not fit 32 bits");
    const BinLayout L = bin_layout(S, capacity);
    if (workspace_bytes < L.total) {
        set_error("qed_bin_tiles: workspace too small (%lld < %lld)", (long long)workspace_bytes, L.total);
        return QED_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    int* n_slots_dev = (int*)(w + L.n_slots_dev);
    unsigned* kA0 = (unsigned*)(w + L.keysA0); unsigned* kA1 = (unsigned*)(w + L.keysA1);
    int* vA0 = (int*)(w + L.valsA0); int* vA1 = (int*)(w + L.valsA1);
    int* block_sums = (int*)(w + L.block_sums);
    unsigned* kB0 = (unsigned*)(w + L.keysB0); unsigned* kB1 = (unsigned*)(w + L.keysB1);
    int* vB = (int*)(w + L.valsB);
    void* sort_ws = w + L.sort_ws;
    if (S == 0 || capacity == 0) {
        hipError_t e = hipMemsetAsync(offsets, 0, (size_t)(n_tot + 1) * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(n_isect, 0, 4, st);
        if (e == hipSuccess && host_words) e = hipMemsetAsync(host_words, 0, 16, st);
        if (e != hipSuccess) { set_error("qed_bin_tiles: memset failed"); return QED_E_LAUNCH; }
        return QED_OK;
    }
    QED_REQUIRE(means2d && radii && depths && tiles_per_gauss && flatten_ids, "null buffers");
    QED_REQUIRE(mode >= QED_BIN_AUTO && mode <= QED_BIN_TILE_SORT, "unknown binning mode");
    const unsigned gridS = (unsigned)((S + 255) / 256);
    // The emit kernel adds up its predecessors' block sums itself (one launch and one serial scan less): every one of the
    // G workgroups reads G ints, O(G^2) L2 traffic in all -- 15 MB at config B (G = 1 954), 1.5 GB at 5 M slots, and it would
    // be ~150 GB at 50 M.  Beyond kSelfScanMaxBlocks workgroups the single-workgroup scan launch (O(G)) runs instead.
    constexpr unsigned kSelfScanMaxBlocks = 16384;
    int* block_offsets = (int*)(w + L.block_offsets);
    auto launch_emit = [&](const int* bsums, const int* order, int* vals_out, unsigned slot_mask = 0xFFFFFFFFu) {
        if (gridS <= kSelfScanMaxBlocks) {
            hipLaunchKernelGGL((isect_emit_kernel<unsigned, true>), dim3(gridS), dim3(256), 0, st, N, C, means2d, radii,
                               depths, tiles_per_gauss, bsums, tile_w, tile_h, tile_bits, n_isect, order, kB0, vals_out,
                               splats, (long long)capacity, status, (const unsigned long long*)tile_masks, slot_mask);
        } else {
            hipLaunchKernelGGL(isect_scan_kernel, dim3(1), dim3(1024), 0, st, bsums, (int)gridS, block_offsets, n_isect,
                               (long long)capacity, status);
            hipLaunchKernelGGL((isect_emit_kernel<unsigned, false>), dim3(gridS), dim3(256), 0, st, N, C, means2d, radii,
                               depths, tiles_per_gauss, (const int*)block_offsets, tile_w, tile_h, tile_bits, n_isect, order,
                               kB0, vals_out, splats, (long long)capacity, status, (const unsigned long long*)tile_masks,
                               slot_mask);
        }
    };
    // Which pipeline: sorting every tile's run by depth costs time in proportion to the list and runs in LDS only
    // for runs of <= 2048 entries; the global depth sort of the slots costs ~12 launch latencies whatever the list.
    // The capacity (1.25 x the longest list seen) per tile decides: short lists per tile -> per-tile sort.
    if (mode == QED_BIN_AUTO) mode = capacity <= 1024 * n_tot ? QED_BIN_TILE_SORT : QED_BIN_TWO_STAGE;
    if (mode == QED_BIN_TILE_SORT) {
        // (1) list positions in SLOT order: block sums of the tile counts (project_fwd's, or counted here), scanned
        // by the emit kernel itself
        const int* bsums = block_sums_in;
        if (bsums == nullptr) {
            hipLaunchKernelGGL(count_sorted_kernel, dim3(gridS), dim3(256), 0, st, (int)S, (const int*)nullptr,
                               tiles_per_gauss, block_sums);
            bsums = block_sums;
        }
        // (2) emit (cam|tile, slot) in slot order, STABLE sort on the tile bits: every tile's run is in slot order.
        // The buffers are dealt so that the sorted values land in vB (the per-tile sort writes flatten_ids).
        const int end_bit = tile_bits + cam_bits;
        const int passes = (end_bit + 7) / 8;
        int* v_first = (passes & 1) ? flatten_ids : vB;
        int* v_alt = (passes & 1) ? vB : flatten_ids;
        launch_emit(bsums, (const int*)nullptr, v_first);
        const int which = sort_pairs_u32(kB0, v_first, kB1, v_alt, n_isect, capacity, end_bit, sort_ws, L.sort_ws_bytes,
                                         status, st);
        if (which < 0) return which;
        const unsigned* tile_keys = which ? kB1 : kB0;
        unsigned* k_spare = which ? kB0 : kB1;
        const long long work = capacity > n_tot + 1 ? capacity : n_tot + 1;
        hipLaunchKernelGGL(tile_offsets_kernel<unsigned>, dim3(tile_offsets_grid(work)), dim3(256), 0, st, tile_keys,
                           (const int*)n_isect, (int)n_tot, tile_w * tile_h, tile_bits, offsets, (const int*)status,
                       host_words);
        // (3) every tile's run into depth order (stable: ties stay in slot order)
        hipLaunchKernelGGL(tile_depth_sort_kernel, dim3((unsigned)((n_tot + 3) / 4)), dim3(256), 0, st,
                           (const int*)offsets, (const int*)vB, depths, flatten_ids, k_spare, (unsigned*)(w + L.tk1),
                           (int*)(w + L.tv0), (int*)(w + L.tv1), (int)n_tot);
        if (isect_ids != nullptr)
            hipLaunchKernelGGL(isect_ids_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, st, tile_keys,
                               (const int*)flatten_ids, depths, (const int*)n_isect, (unsigned long long*)isect_ids);
        return check_launch("qed_bin_tiles");
    }
    // stage A: (camera, Gaussian) slots into depth order; the values carry every slot's tile count above the slot bits
    // when at least four bits are free (up to 2^28 slots)
    int slot_bits = 1;
    while ((1ll << slot_bits) < S) ++slot_bits;
    if (slot_bits > 28) slot_bits = 32;
    const unsigned slot_mask = slot_bits >= 32 ? 0xFFFFFFFFu : (1u << slot_bits) - 1u;
    hipLaunchKernelGGL(depth_keys_kernel, dim3(gridS), dim3(256), 0, st, (int)S, radii, depths, kA0, vA0, n_slots_dev,
                       tiles_per_gauss, slot_bits);
    int which = sort_pairs_u32(kA0, vA0, kA1, vA1, n_slots_dev, S, 32, sort_ws, L.sort_ws_bytes, status, st);
    if (which < 0) return which;
    const int* order = which ? vA1 : vA0;
    // intersection counts in depth order -> offsets, M
    hipLaunchKernelGGL(count_sorted_kernel, dim3(gridS), dim3(256), 0, st, (int)S, order, tiles_per_gauss, block_sums,
                       slot_bits);
    // stage B: emit (cam|tile, slot) in depth order, then a STABLE sort on the tile bits only.  The pass
    // count decides which buffer to emit into so that the sorted values land in `flatten_ids`.
    const int end_bit = tile_bits + cam_bits;
    const int passes = (end_bit + 7) / 8;
    int* v_first = (passes & 1) ? vB : flatten_ids;
    int* v_alt = (passes & 1) ? flatten_ids : vB;
    launch_emit((const int*)block_sums, order, v_first, slot_mask);
    which = sort_pairs_u32(kB0, v_first, kB1, v_alt, n_isect, capacity, end_bit, sort_ws, L.sort_ws_bytes, status, st);
    if (which < 0) return which;
    const unsigned* tile_keys = which ? kB1 : kB0;
    const long long work = capacity > n_tot + 1 ? capacity : n_tot + 1;
    hipLaunchKernelGGL(tile_offsets_kernel<unsigned>, dim3(tile_offsets_grid(work)), dim3(256), 0, st, tile_keys,
                       (const int*)n_isect, (int)n_tot, tile_w * tile_h, tile_bits, offsets, (const int*)status,
                       host_words);
    if (isect_ids != nullptr)
        hipLaunchKernelGGL(isect_ids_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, st, tile_keys,
                           (const int*)flatten_ids, depths, (const int*)n_isect, (unsigned long long*)isect_ids);
    return check_launch("qed_bin_tiles");
}

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
    hipLaunchKernelGGL((isect_emit_kernel<unsigned long long, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, N, C,
                       means2d, radii, depths, tiles_per_gauss, block_offsets, tile_w, tile_h, tile_bits,
                       const_cast<int*>(n_isect), (const int*)nullptr, (unsigned long long*)keys, vals,
                       (const float*)nullptr, 0ll, (int*)nullptr);
    return check_launch("qed_isect_emit");
}

extern "C" int qed_tile_offsets(const uint64_t* sorted_keys, const int32_t* n_dev, int64_t capacity, int32_t C,
                                int32_t n_tiles, int32_t tile_bits, int32_t* offsets, void* stream) {
    QED_REQUIRE(n_dev && offsets && C >= 1 && n_tiles >= 1, "bad arguments");
    QED_REQUIRE(capacity >= 0 && capacity < (1ll << 31), "capacity out of range");
    const long long n_tot = (long long)C * n_tiles;
    const long long work = capacity > n_tot + 1 ? capacity : n_tot + 1;
    const unsigned grid = tile_offsets_grid(work);
    hipLaunchKernelGGL(tile_offsets_kernel<unsigned long long>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)sorted_keys, n_dev, (int)n_tot, n_tiles, tile_bits, offsets);
    return check_launch("qed_tile_offsets");
}
