// Shared helpers for the gfx950 kernels of libqed_splat.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qed_splat.h"

namespace qed {

void set_error(const char* fmt, ...);

// radix_sort.hip: 32-bit-key stable LSD sort of (key, value) pairs; same contract as qed_sort_pairs
long long sort32_workspace_bytes(long long capacity);
int sort_pairs_u32(unsigned* keys, int* vals, unsigned* keys_alt, int* vals_alt, const int* n_dev, long long capacity,
                   int end_bit, void* workspace, long long workspace_bytes, int* status, hipStream_t st);

// constants of the operator behind model.py:267-288
constexpr float kAlphaMax = 0.999f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTMin = 1e-4f;
constexpr float kJacMargin = 0.3f;

constexpr int kWave = 64;  // gfx950 wavefront

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return QED_E_LAUNCH;
    }
    return QED_OK;
}

#define QED_REQUIRE(cond, msg)                                 \
    do {                                                       \
        if (!(cond)) {                                         \
            qed::set_error("%s: %s", __func__, msg);           \
            return QED_E_INVALID_ARG;                          \
        }                                                      \
    } while (0)

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + __expf(-x)); }

// tile rectangle [x0,x1) x [y0,y1) of a projected Gaussian (gsplat isect_tiles)
__device__ __forceinline__ void tile_rect(float mx, float my, float radius, int tile_w, int tile_h, int& x0,
                                          int& y0, int& x1, int& y1) {
    const float ts = (float)QED_TILE;
    const float tr = radius / ts, tx = mx / ts, ty = my / ts;
    x0 = min(max(0, (int)floorf(tx - tr)), tile_w);
    y0 = min(max(0, (int)floorf(ty - tr)), tile_h);
    x1 = min(max(0, (int)ceilf(tx + tr)), tile_w);
    y1 = min(max(0, (int)ceilf(ty + tr)), tile_h);
}

// QED_F_TIGHT_TILES: the list only needs the tiles of gsplat's 3-sigma square in which SOME pixel can reach
// alpha >= 1/255, i.e. sigma <= tau = ln(255 o).  The ellipse {sigma <= tau} of the conic (a, b, c) has
// half extents sqrt(2 tau c / det) in x and sqrt(2 tau a / det) in y; tau and the extents are inflated well
// beyond fp32 rounding of the per-pixel evaluation, and never exceed the 3-sigma radius.  Tiles dropped
// this way are tiles every pixel would have skipped, so images and gradients do not change.
__device__ __forceinline__ void tight_tile_rect(float mx, float my, float radius, float a, float b, float c, float tau,
                                                int tile_w, int tile_h, int& x0, int& y0, int& x1, int& y1) {
    const float t = tau * (1.f + 1e-4f) + 1e-2f;
    if (!(t > 0.f)) { x0 = y0 = x1 = y1 = 0; return; }             // opacity <= 1/255: nothing can be drawn
    const float det = fmaxf(a * c - b * b, 1e-30f);
    const float rx = fminf(radius, sqrtf(2.f * t * c / det) * (1.f + 1e-5f) + 0.01f);
    const float ry = fminf(radius, sqrtf(2.f * t * a / det) * (1.f + 1e-5f) + 0.01f);
    const float ts = (float)QED_TILE;
    x0 = min(max(0, (int)floorf((mx - rx) / ts)), tile_w);
    y0 = min(max(0, (int)floorf((my - ry) / ts)), tile_h);
    x1 = min(max(0, (int)ceilf((mx + rx) / ts)), tile_w);
    y1 = min(max(0, (int)ceilf((my + ry) / ts)), tile_h);
}

// ---- exact-conservative culling of (Gaussian, rectangle of pixel centres) pairs ------------------------------------
// (the compositing kernels test the four 8x8 quadrants of a tile, composite.hip; the projection kernel whole tiles)
// alpha >= 1/255  <=>  sigma(d) = (a dx^2 + c dy^2)/2 + b dx dy <= tau = ln(255 o).  sigma is convex, so over the
// rectangle of a quadrant's (or a tile's) pixel centres its minimum is 0 if the mean lies inside and otherwise sits on one of the
// four edges; along an edge it is a 1-D quadratic whose minimiser is clamped to the edge (v_med3).  A pair whose
// minimum exceeds tau by more than the rounding margin is one every pixel would have skipped.
//
// sigma minimised over t in [lo, hi] on the line where the other coordinate is fixed:
//   h = (own diagonal term)/2 * fixed^2,  bb = b * fixed,  s = -bb / (other diagonal term),  ho = (other term)/2
__device__ __forceinline__ float edge_min(float h, float bb, float s, float ho, float lo, float hi) {
    const float t = __builtin_amdgcn_fmed3f(s, lo, hi);
    return __builtin_fmaf(t, __builtin_fmaf(ho, t, bb), h);
}

// Everything a lane needs about its staged Gaussian to test rectangles of the tile at pixel origin (ox, oy).
// ia, ic = 1/a, 1/c to within an ulp (v_rcp_f32): they only place the point on an edge at which sigma is
// evaluated, and a point off the minimiser by one part in 1e7 raises the value by one part in 1e14 -- the margin
// is eleven orders of magnitude wider.  The margin uses the tile's extent for all four quadrants.
struct CullGauss {
    float b, ha, hc, ia, ic, X0, Y0, thr;
};
__device__ __forceinline__ CullGauss cull_setup(const float4& r0, const float4& r1, float tau, float ox, float oy) {
    CullGauss g;
    const float a = r0.z, c = r1.x;
    g.b = r0.w; g.ha = 0.5f * a; g.hc = 0.5f * c;
    g.ia = __builtin_amdgcn_rcpf(a); g.ic = __builtin_amdgcn_rcpf(c);
    g.X0 = ox + 0.5f - r0.x; g.Y0 = oy + 0.5f - r0.y;
    const float ax = fmaxf(fabsf(g.X0), fabsf(g.X0 + 15.f)), ay = fmaxf(fabsf(g.Y0), fabsf(g.Y0 + 15.f));
    const float scale = a * ax * ax + c * ay * ay + fabsf(g.b) * ax * ay;
    g.thr = tau + 1e-3f + 8e-6f * scale;
    return g;
}
// the whole 16x16 tile at the origin cull_setup was given: can ANY of its pixels reach alpha >= 1/255?  (The same
// arithmetic as composite.hip's quadrant_may_touch on the rectangle X0 .. X0 + 15, Y0 .. Y0 + 15; a quadrant's
// rectangle is a subset, so a tile this test drops is one whose four quadrants the compositing kernels would have
// dropped too, up to roundings eleven orders of magnitude inside the margin.)
__device__ __forceinline__ bool tile_may_touch(const CullGauss& g) {
    const float xl = g.X0, xh = xl + 15.f, yl = g.Y0, yh = yl + 15.f;
    auto vline = [&](float X) { const float bb = g.b * X; return edge_min(g.ha * X * X, bb, -bb * g.ic, g.hc, yl, yh); };
    auto hline = [&](float Y) { const float bb = g.b * Y; return edge_min(g.hc * Y * Y, bb, -bb * g.ia, g.ha, xl, xh); };
    const float m = fminf(fminf(vline(xl), vline(xh)), fminf(hline(yl), hline(yh)));
    const bool inside = (xl <= 0.f) & (xh >= 0.f) & (yl <= 0.f) & (yh >= 0.f);
    return inside | !(m > g.thr);
}

// r-th set bit (r = 0 for the lowest) of a 64-bit mask that has more than r bits set
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int r) {
    unsigned x = (unsigned)m;
    int base = 0;
    int c = __popc(x);
    if (r >= c) { r -= c; x = (unsigned)(m >> 32); base = 32; }
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        c = __popc(x & ((1u << w) - 1u));
        if (r >= c) { r -= c; x >>= w; base += w; }
    }
    return base;
}

// tile rectangle of a splat record (slot 11): x0 | y0 << 11 | width << 22  (tile grids up to 2047 x 2047,
// rectangles up to 1023 tiles wide); the height follows from tiles_per_gauss
__device__ __forceinline__ unsigned pack_tile_rect(int x0, int y0, int x1) {
    return (unsigned)x0 | ((unsigned)y0 << 11) | ((unsigned)(x1 - x0) << 22);
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
// contiguous range of tiles (whole image rows): neighbouring tiles share splat records in one L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---- spherical harmonics (standard real SH basis, 3DGS constants) ------------------------------
constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;

// basis values b[0..K) for unit direction (x,y,z)
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b) {
    b[0] = SH_C0;
    if constexpr (DEG > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    }
    if constexpr (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = 1.0925484305920792f * xy;
        b[5] = -1.0925484305920792f * yz;
        b[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
        b[7] = -1.0925484305920792f * xz;
        b[8] = 0.5462742152960396f * (xx - yy);
        if constexpr (DEG > 2) {
            b[9] = -0.5900435899266435f * y * (3.f * xx - yy);
            b[10] = 2.890611442640554f * xy * z;
            b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
            b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
            b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
            b[14] = 1.445305721320277f * z * (xx - yy);
            b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
        }
    }
}


// ---- wave64 cross-lane helpers (DPP; no LDS traffic) --------------------------------------
// sum over the 16 lanes of each DPP row; result valid in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    // quad_perm / row_ror rotate within a row of 16 lanes
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    // row_ror:8 as ONE instruction.  Written with the builtin, the compiler sinks this last add into a following
    // `if (lane % 16 == 0)` and leaves v_mov 0 + v_mov_dpp outside it: three instructions instead of one, per value.
    // (s_nop 1: a DPP read needs two wait states after the VALU write of its source; the compiler cannot see
    // that hazard through inline asm.)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
    return v;
}

// full wave64 sum, result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- shared between loss.hip and ssim.hip: the sums workspace of the loss passes -------------------------------
// sums (QED_LOSS_SUMS_FLOATS floats): from [8] on four rows of kLossMaxGrid per-workgroup partials: n_valid, max depth
// (pass 1), sum |rgb - gt|, sum |depth - gt| (pass 2); loss_finalize_kernel folds them into sums[0..3] and the losses.
constexpr int kLossMaxGrid = (QED_LOSS_SUMS_FLOATS - 8) / 4;
__device__ __forceinline__ float* loss_part(float* sums, int row) { return sums + 8 + row * kLossMaxGrid; }

// Advancing the optimiser's device-resident step state (qed_adam_step_dev / qed_adam_step_sh): state = {step (as float),
// 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step)}; one thread.  A launch of its own (adam_tick_kernel) or, in the fused
// training step, a passenger of the loss pass's one-workgroup fold launch (qed_loss_grad_ssim's `tick`): that launch
// sits between the previous step's Adam launches and this step's, which is all the tick needs.
struct AdamTick {
    float* state;           // NULL: nothing to do
    float beta1, beta2;
    float* lr_slot;         // NULL, or the scheduled group's learning-rate slot (lr_exp_decay_kernel's formula)
    float log_init, log_final, inv_max_steps;
    const int* skip;        // NULL, or the word that makes the step a no-op when non-zero (see adam_skipped, loss.hip)
};

__device__ __forceinline__ void adam_tick(const AdamTick& t) {
    if (t.skip != nullptr && t.skip[0] != 0) return;
    if (t.lr_slot != nullptr) {         // scheduled rate of the step about to be taken
        const float u = fminf(fmaxf(t.state[0] * t.inv_max_steps, 0.f), 1.f);
        t.lr_slot[0] = expf(t.log_init * (1.f - u) + t.log_final * u);
    }
    const float n = t.state[0] + 1.f;
    t.state[0] = n;
    t.state[1] = 1.f / (1.f - powf(t.beta1, n));
    t.state[2] = 1.f / sqrtf(1.f - powf(t.beta2, n));
}

__global__ void loss_finalize_kernel(int n_pix, int n_blocks, int has_depth, float* __restrict__ sums, float rgb_weight,
                                     float depth_lambda, float* __restrict__ losses, const float* __restrict__ extra_sum,
                                     int extra_n, float extra_scale, float extra_offset, AdamTick tick);

// ---- costliest-first tile order of the compositing backward (composite.hip's tile_order_kernel; ssim.hip carries it as
// a passenger workgroup of the SSIM forward launch in the fused training step) -----------------------------------------
// Workgroups are dispatched in index order as wave slots free up, so handing the tiles out in order of decreasing cost is
// greedy longest-processing-time-first scheduling: the end of the launch is filled with the cheapest tiles instead of with
// whatever the image's corner holds.  The cost is the forward pass's own count of (Gaussian, quadrant) visits on the tile
// (qed_composite_fwd's tile_cost), which the backward pass repeats within a per cent; the sorted list's length per tile
// is NOT a usable predictor (culling and early termination decide).  Measured at config B: 336 -> 293 us.
// One workgroup of NT threads: counting sort on min(cost, 4095), descending (order inside a bucket is arbitrary: it only
// permutes the order of the float atomics, which is arbitrary anyway).  order[n_tiles] receives n_split = the number of
// leading tiles whose cost exceeds `split_factor` x (total cost / wave slots): a single wave on such a tile would set the
// length of the launch by itself, so the kernel deals them as four quadrant waves each.
#ifndef QED_K7_WAVES
#define QED_K7_WAVES 4                                    // waves per SIMD the compositing backward is built for
#endif
// XCD-aware (round 5): workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own, and a tile's neighbours
// read largely the same splat records.  A plain cost order scatters neighbours over all XCDs (both compositing kernels then
// fetch ~3x the bytes of a raster-order launch).  So the image is cut into 8 REGIONS of consecutive tiles, every region's
// tiles are sorted by cost on their own, and the regions are interleaved in step with the block index: position p holds
// the next costliest tile of region (p + 3 n_split) mod 8 -- the XCD that runs block p.  Each XCD's working set is then one
// region's records (3 MB of 24 at config B: it fits the 4 MB L2), and every XCD still works costliest-first.  The few
// tiles by which the regions' sizes differ go to the very end.  Measured at config B against the plain cost order
// (scripts/xcd_order_ab.py): K7 237 us against 245, K6 108 against 110.
// Cost resolution: 256 buckets of 16 per region.
constexpr int kCostShift = 4;                             // cost units per bucket = 16
constexpr int kCostBuckets = 256;
constexpr int kCostClip = (kCostBuckets << kCostShift) - 1;          // 4095
constexpr int kOrderRegions = 8;
constexpr int kOrderCells = kCostBuckets * kOrderRegions;            // (bucket, region) cells
constexpr int kOrderLdsInts = 2 * kOrderCells + 16 + 2 + 6 * kOrderRegions + 2;   // cell bases, cell tickets, wave totals, total, per-region words
struct TileOrderJob {
    const int* cost4;       // [n_tiles][4] (NULL: no job)
    int n_tiles;
    int* order;             // [n_tiles + 1]
    float split_factor;
    int slots;              // wave slots of the backward launch
    int max_split;
};
// at most an eighth of the tiles are split (the grid must be fixed before the count is known)
inline int max_split_tiles(long long grid) { return (int)(grid / 8); }

template <int NT>
__device__ __forceinline__ void tile_order_body(const TileOrderJob& job, int* __restrict__ lds) {
    static_assert(NT % 64 == 0 && NT <= 1024 && NT >= kCostBuckets, "workgroup size");
    constexpr int kRegs = 8;                              // tiles per thread held in registers between the passes: all of 1080p's
                                                          // 8 160 for the 1 024-thread launch, the first 2 048 for a 256-thread passenger
    constexpr int R = kOrderRegions;
    int* cnt = lds;                                       // [bucket][region]: tiles, then the cell's first rank inside its region
    int* tick = lds + kOrderCells;                        // [bucket][region] tickets handed out in the placement pass
    int* wave_tot = lds + 2 * kOrderCells;                // [16]
    long long* total_s = reinterpret_cast<long long*>(wave_tot + 16);
    int* reg_tot = wave_tot + 18;                         // [R] tiles of the region
    int* reg_heavy = reg_tot + R;                         // [R] of them above the split threshold
    int* reg_head = reg_heavy + R;                        // [R] first head position of the region's heavy tiles
    int* reg_i0 = reg_head + R;                           // [R] first interleaved position of the region
    int* reg_left = reg_i0 + R;                           // [R] first position of the region's left-over tiles
    int* misc = reg_left + R;                             // [0] n_split, [1] interleaved rounds (the smallest region's size)
    const int* __restrict__ cost4 = job.cost4;
    int* __restrict__ order = job.order;
    const int n_tiles = job.n_tiles;
    const int per = (n_tiles + R - 1) / R;                                       // tiles per region
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto cost_of = [&](int i) { const int4 c4 = reinterpret_cast<const int4*>(cost4)[i]; return c4.x + c4.y + c4.z + c4.w; };
    auto bucket_of = [](int c) { return kCostBuckets - 1 - (min(c, kCostClip) >> kCostShift); };   // bucket 0 = the costliest
    auto region_of = [&](int i) { return min(i / per, R - 1); };
    // the first 8 NT tiles stay in registers between the two passes: their loads are requested together, one memory round
    // trip; the rest are re-read from L2 in the second pass
    int creg[kRegs];
#pragma unroll
    for (int j = 0; j < kRegs; ++j) {
        const int i = tid + NT * j;
        creg[j] = cost_of(i < n_tiles ? i : 0);
    }
    for (int i = tid; i < 2 * kOrderCells; i += NT) lds[i] = 0;
    if (tid == 0) total_s[0] = 0;
    __syncthreads();
    long long mine = 0;
#pragma unroll
    for (int j = 0; j < kRegs; ++j)
        if (tid + NT * j < n_tiles) { mine += creg[j]; atomicAdd(&cnt[bucket_of(creg[j]) * R + region_of(tid + NT * j)], 1); }
    for (int i = tid + NT * kRegs; i < n_tiles; i += NT) {
        const int c = cost_of(i);
        mine += c;
        atomicAdd(&cnt[bucket_of(c) * R + region_of(i)], 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) atomicAdd((unsigned long long*)total_s, (unsigned long long)mine);
    __syncthreads();
    // threshold of the split (in cost units -> bucket index): buckets [0, first_light) hold cost > thr
    const float per_slot = (float)total_s[0] / (float)max(job.slots, 1);
    const int thr = (int)fminf(job.split_factor * per_slot, (float)kCostClip);
    const int first_light = kCostBuckets - 1 - (thr >> kCostShift);
    // per region: exclusive scan of its 256 bucket counts (thread b < 256 owns bucket b) -> the cell's first rank in the region
    for (int x = 0; x < R; ++x) {
        const int v = tid < kCostBuckets ? cnt[tid * R + x] : 0;
        int sc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(sc, o, 64); if (lane >= o) sc += y; }
        if (lane == 63) wave_tot[wid] = sc;
        __syncthreads();
        int base = sc - v;
        for (int w = 0; w < wid; ++w) base += wave_tot[w];
        if (tid < kCostBuckets) {
            cnt[tid * R + x] = base;
            if (tid == first_light) reg_heavy[x] = base;                         // tiles of the region in heavier buckets
            if (tid == kCostBuckets - 1) reg_tot[x] = base + v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        int heavy = 0;
        for (int x = 0; x < R; ++x) heavy += reg_heavy[x];
        // (more "heavy" tiles than the launch may split: a flat cost distribution -- nothing is split)
        const int ns = heavy <= job.max_split ? heavy : 0;
        if (ns == 0) for (int x = 0; x < R; ++x) reg_heavy[x] = 0;
        int head = 0, rounds = 0x7fffffff;
        for (int x = 0; x < R; ++x) {
            reg_head[x] = head;
            head += reg_heavy[x];
            rounds = min(rounds, reg_tot[x] - reg_heavy[x]);
        }
        int left = ns + R * rounds;
        for (int x = 0; x < R; ++x) {
            // block of position p = p + 3 ns, on XCD (p + 3 ns) mod 8: the region's first position at or behind ns
            reg_i0[x] = ns + ((x - 4 * ns) & (R - 1));
            reg_left[x] = left;
            left += reg_tot[x] - reg_heavy[x] - rounds;
        }
        misc[0] = ns;
        misc[1] = rounds;
        order[n_tiles] = ns;
    }
    __syncthreads();
    const int rounds = misc[1];
    auto place = [&](int i, int c) {
        const int cell = bucket_of(c) * R + region_of(i), x = region_of(i);
        const int k = cnt[cell] + atomicAdd(&tick[cell], 1);                     // rank inside the region, costliest first
        const int h = reg_heavy[x];
        int pos;
        if (k < h) pos = reg_head[x] + k;                                        // one of the split tiles: the head of the order
        else if (k - h < rounds) pos = reg_i0[x] + R * (k - h);                  // interleaved with the other regions
        else pos = reg_left[x] + (k - h - rounds);                               // the tiles by which this region is larger
        order[pos] = i;
    };
#pragma unroll
    for (int j = 0; j < kRegs; ++j)
        if (tid + NT * j < n_tiles) place(tid + NT * j, creg[j]);
    for (int i = tid + NT * kRegs; i < n_tiles; i += NT) place(i, cost_of(i));
}

// ---- pass 1 of the fused image loss (loss.hip's loss_reduce_kernel; ssim.hip carries it as passenger workgroups of the
// SSIM forward launch in the fused training step): workgroup `block` of `n_blocks` -------------------------------------
// Only what must be known BEFORE a gradient can be written: the number of valid depth pixels (its reciprocal scales every
// depth gradient) and the largest rendered depth (the value alpha == 0 pixels take, model.py:306).  It reads the depth
// channel, the ground-truth depth and the mask -- not the colours.  s: [2][NT / 64] floats of LDS.
template <int CH, int NT>
__device__ __forceinline__ void loss_reduce_body(int block, int n_blocks, int n_pix, const float* __restrict__ render,
                                                 const float* __restrict__ gt_depth, const float* __restrict__ mask,
                                                 float* __restrict__ sums, float (*s)[NT / 64]) {
    // rows 2 and 3 (the loss sums of pass 2): zeroed here for the fused SSIM-backward + gradient pass (ssim.hip), whose
    // workgroups -- more than kLossMaxGrid at 1080p -- ADD their partials to slot (index mod this grid); the plain pass 2
    // overwrites its slots
    if (threadIdx.x == 0) { loss_part(sums, 2)[block] = 0.f; loss_part(sums, 3)[block] = 0.f; }
    if constexpr (CH != 4) return;
    float nv = 0.f;
    float dmax = -3.0e38f;
    for (size_t i = (size_t)block * NT + threadIdx.x; i < (size_t)n_pix; i += (size_t)n_blocks * NT) {
        const float d = render[4 * i + 3];
        dmax = fmaxf(dmax, d);
        const float m = mask ? mask[i] : 1.f;
        const float dg = gt_depth[i] * m;
        // the predicted depth is finite whenever the render is; NaN renders fail isfinite below
        const float dp = d * m;
        if (isfinite(dp) && isfinite(dg) && dg > 0.f) nv += 1.f;
    }
    nv = wave_sum(nv);
    dmax = wave_max(dmax);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { s[0][wid] = nv; s[1][wid] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tn = 0.f, tm = -3.0e38f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { tn += s[0][w]; tm = fmaxf(tm, s[1][w]); }
        loss_part(sums, 0)[block] = tn;
        loss_part(sums, 1)[block] = tm;
    }
}

// the job of one backward launch over `grid` tiles on this device (host side; asked of the runtime at every call)
inline TileOrderJob tile_order_job(const int* tile_cost, long long grid, int* order_ws) {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    return TileOrderJob{tile_cost, (int)grid, order_ws, 1.0f, n_cu * 4 * QED_K7_WAVES, max_split_tiles(grid)};
}

// grid of the two streaming loss passes (pass 2 reads pass 1's per-workgroup partials by index)
inline unsigned loss_reduce_grid(long long n_pix) {
    long long g = (n_pix + 255) / 256;
    if (g > kLossMaxGrid) g = kLossMaxGrid;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace qed
