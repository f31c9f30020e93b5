// K6 / K7: per-tile alpha compositing, forward and backward, for gfx950 (wave64).
//
// Replaces gsplat rasterize_to_pixels fwd/bwd behind model.py:267-288.
//
// Design (CDNA4-first, not a warp-32 translation):
//   * ONE 64-lane wave per 16x16 tile.  Lane l owns four pixels, the same (l & 7, l >> 3) position
//     in each of the tile's four 8x8 quadrants.  A workgroup is a single wave, so the kernel is
//     wave-synchronous: no workgroup barriers, no inter-wave imbalance, early-out per tile.  The LAST
//     tiles of a launch are dealt as four one-quadrant waves each (one pixel per lane): finer work
//     items fill the end of the launch (see composite_fwd_kernel, big_tiles(), small_wave()).
//   * The tile's run of the depth-sorted list is streamed in batches of 64 through a two-deep
//     software pipeline: while batch b is composited, the 48-byte records (three 16-byte loads per
//     lane) of b+1 and the ids of b+2 are in flight, every prefetch an unconditional load from a
//     clamped index.  After culling each lane parks its (pre-scaled) record in LDS and the loop fetches
//     the current Gaussian with broadcast reads (every lane the same address): a v_readlane costs 11-12
//     issue cycles of the vector pipe, ten of them more than the forward pass's arithmetic per Gaussian.
//   * Exact-conservative cull by ballot, per 8x8 quadrant: each lane bounds the minimum of its
//     Gaussian's quadratic form sigma over the quadrant's pixel-centre rectangle (convex: 0 if the
//     mean is inside, else attained on an edge).  alpha >= 1/255 needs sigma <= ln(255 o), so a
//     Gaussian whose bound exceeds that (plus an fp32 rounding margin) cannot pass the alpha test at
//     any pixel of the quadrant.  64-bit ballots give one survivor mask per quadrant; the loop walks
//     the set bits with scalar instructions.  Results are identical to walking the whole list
//     (launch flag QED_CL_NO_CULL does that: tested bit-identical), at 1.5-1.65x the speed.
//   * The inner loop is branch-free: 4 independent pixel chains per lane; every predicate ("done",
//     "accepted", "valid") is a 64-bit scalar mask fed straight to v_cndmask.  The conic is
//     pre-scaled by -log2(e) at staging so the exponent is a bare v_exp_f32 of a 5-instruction
//     polynomial.
//   * Backward: the per-pixel gradients of one Gaussian are summed over the lane's four pixels; ONE
//     halving level runs in the wave (v_permlane16_swap), the 32 partials per value of up to three
//     Gaussians are parked in LDS and the flush -- one lane per (Gaussian, value) -- adds them in a
//     fixed order and issues ONE 64-byte-row atomic request per (tile, Gaussian) that contributed.
//     The launch hands its tiles out costliest first (tile_order_kernel), from the forward pass's own
//     per-tile visit counts.
#include <type_traits>

#include "qed_common.h"

// Development switches (A/B builds: python -m qed_splatter_amd.build --variant NAME -DQED_K7_FORM=1):
//   QED_K6_FORM / QED_K7_FORM  1 = one walk that tests the per-pixel form per Gaussian,
//                              2 = two copies of the walk inline, the test per BATCH (K7's product form),
//                              4 = (K6) the fast walk inline, the mixed walk OUT OF LINE on a copy of the state (K6's product
//                                  form: the hot loop keeps its scalar registers; 124-126 us against 128, r05_k6_k7_variants.txt)
#ifndef QED_K6_FORM
#define QED_K6_FORM 4
#endif
#ifndef QED_K7_FORM
#define QED_K7_FORM 2
#endif

namespace qed {

typedef unsigned long long u64;

constexpr int kBatch = 64;
constexpr float kLog2e = 1.4426950408889634f;

// How the current Gaussian's scalars reach all 64 lanes.  Measured on MI355X (scripts/ubench/xlane_cycles.hip): a
// v_readlane_b32 costs 11-12 issue cycles of the SIMD's vector pipe (a plain VALU op 2.2-2.5), so ten of them per
// Gaussian were 115 cycles -- more than the forward pass's arithmetic for that Gaussian.  Parking the batch's records
// in LDS once (three 16-byte stores per lane) and fetching the current one with broadcast reads (every lane the same
// address: two ds_read_b128 + one ds_read_b64) costs ~45 cycles of LDS time per Gaussian, which runs beside the
// vector pipe instead of on it.  (The v_readlane form and the fully in-wave reduction it was measured against live on
// in scripts/ubench/xlane_cycles.hip and DESIGN.md section 4, not in the product source.)
constexpr int kRecFloats = 12;      // LDS record stride (floats): 10 used + the list id (backward) + 1 pad

// Backward: how the 12 per-Gaussian gradient sums leave the wave.  Cross-lane instructions are the expensive ones
// (v_permlane*_swap 8.4, DPP 4.3 issue cycles against 2.2-2.5 for a plain op: scripts/ubench/xlane_cycles.hip), so there
// is NO reduction level in the wave: every lane parks its own 12 sums in LDS (stores are issued beside the vector pipe,
// not on it; conflict-free: lane = bank) and the flush -- two lanes per (Gaussian, value) -- adds the 64 partials in a
// fixed order, one DPP add joins the two halves, one lane issues the atomic.  (Rounds 2-4 ran one halving level in the
// wave -- six v_permlane16_swap + six adds per Gaussian -- and parked 32 partials of up to three Gaussians; measured
// against each other under the round-5 visit body the two forms tie (274.5 vs 276 us), this one without the 13 M
// bank-conflict cycles per launch of the other: profiles/r05_k6_k7_variants.txt.)
constexpr int kParkSlots = 2;                // Gaussians parked per flush: 24 rows, two lanes each
constexpr int kParkStride = 68;              // floats per (Gaussian, value) row: 64 partials + 4 pad: the 16-byte reads of the
                                             // flush (row = lane >> 1, half = lane & 1; ds_read_b128 lane groups
                                             // {0-3,12-15,20-27}, ...) hit 16 different 4-bank slots: slot = row + 8 half + j mod 16
constexpr int kParkSlot = 12 * kParkStride;  // floats per parked Gaussian

// Diagnostic build only (-DQED_COMPOSITE_STATS; scripts/composite_stats.py): how much work each stage of the
// compositing kernels really does.  Wave-uniform counts, added by lane 0.
#ifdef QED_COMPOSITE_STATS
__device__ unsigned long long g_stats[32];
#define QED_STAT(i, n) do { if (threadIdx.x == 0) atomicAdd(&g_stats[i], (unsigned long long)(n)); } while (0)
#else
#define QED_STAT(i, n) do { } while (0)
#endif

// per-lane select by a wave-uniform 64-bit mask held in an SGPR pair: bit set -> a, else b
__device__ __forceinline__ float sel(u64 m, float a, float b) {
    float d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m));
    return d;
}
__device__ __forceinline__ int sel(u64 m, int a, int b) {
    int d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m));
    return d;
}
// (exact-conservative culling of (Gaussian, rectangle) pairs: edge_min / CullGauss / cull_setup live in qed_common.h --
// the projection kernel applies the same test to whole tiles when it builds the tight lists)
// quadrant (i, j) = (right half, lower half).  The per-line terms of the four x- and four y-values of a tile are
// common subexpressions of the four quadrants' tests (two quadrants share each line).
__device__ __forceinline__ bool quadrant_may_touch(const CullGauss& g, int i, int j) {
    const float xl = g.X0 + (float)(8 * i), xh = xl + 7.f, yl = g.Y0 + (float)(8 * j), yh = yl + 7.f;
    auto vline = [&](float X) { const float bb = g.b * X; return edge_min(g.ha * X * X, bb, -bb * g.ic, g.hc, yl, yh); };
    auto hline = [&](float Y) { const float bb = g.b * Y; return edge_min(g.hc * Y * Y, bb, -bb * g.ia, g.ha, xl, xh); };
    const float m = fminf(fminf(vline(xl), vline(xh)), fminf(hline(yl), hline(yh)));
    // (& and |, not && and ||: branch-free, so the line terms are shared between quadrants instead of being
    // recomputed inside four separately predicated blocks)
    const bool inside = (xl <= 0.f) & (xh >= 0.f) & (yl <= 0.f) & (yh >= 0.f);
    return inside | !(m > g.thr);
}
// per-quadrant survivor masks of the 64 staged Gaussians (one per lane).  NQ = 4: the whole tile;
// NQ = 1: only quadrant q0 (a wave that owns one 8x8 quadrant of a tile, see composite_fwd_kernel)
// q0 carries the quadrant of a one-quadrant wave in bits 0-1 and, in bit 2, the test hook "do not cull" (every
// staged Gaussian reaches the per-pixel code; results must not change -- tests/test_gpu_parity.py)
template <int NQ>
__device__ __forceinline__ void quadrant_masks(bool present, const float4& r0, const float4& r1, float tau, float ox,
                                               float oy, int q0, u64* mq) {
    const CullGauss g = cull_setup(r0, r1, tau, ox, oy);
    const bool keep_all = (q0 >> 2) & 1;
    bool k[NQ];
    if constexpr (NQ == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) k[q] = present & (keep_all | quadrant_may_touch(g, q & 1, q >> 1));
    } else {
        k[0] = present & (keep_all | quadrant_may_touch(g, q0 & 1, (q0 >> 1) & 1));
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) mq[q] = __ballot(k[q]);
}

// ================================================================================================
// packed-math quadrant bodies
// ================================================================================================
// CDNA4's fp32 vector peak needs v_pk_* (two fp32 results per lane per issue), so the per-pixel
// arithmetic is written on 2-vectors: (dx, dy), colour pairs, gradient pairs.  Wave-uniform Gaussian
// data arrive as SGPR pairs.
typedef float f2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) Float3 { float a, b, c; };

__device__ __forceinline__ u64 uniform_u64(u64 v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// The conic reaches the per-pixel code as the two rows of M' = k [[a, b], [b, c]], k = -log2(e)/2 (scaled once per lane at
// staging): w = M' d, and p = d . w = -sigma log2(e).  The backward pass needs M d for the gradient of the mean as well:
// it is w / k, so the same two packed instructions serve both (the factor 1 / k is applied once per (tile, Gaussian), at
// the flush).
constexpr float kConicScale = -0.5f * kLog2e;
__device__ __forceinline__ f2 conic_times_d(f2 R0, f2 R1, f2 d) {
    // (k (a dx + b dy), k (b dx + c dy)) as fma(R0, dx, R1 dy) -- spelled out: left to the compiler's contraction of
    // `R0 dx + R1 dy` the product that is rounded first is whichever operand it has at hand, which differed between the
    // quadrants of ONE kernel (and could between the two kernels): alphas an ulp apart for the same (pixel, Gaussian)
    return __builtin_elementwise_fma(R0, (f2){d.x, d.x}, R1 * (f2){d.y, d.y});
}

// the third 16 bytes of a splat record without its last word (the packed tile rectangle, which only the binning reads):
// a 12-byte load.  As a 16-byte load the word's register is dead on arrival, the allocator hands it out at once, and the
// write-after-write hazard makes the wave wait for the load right behind its issue.
typedef float f3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ f3 load_rec_tail(const float4* __restrict__ splats, size_t g) {
    return *reinterpret_cast<const f3*>(&splats[3 * g + 2]);
}

// The pixel centres of a lane: ONE pair (its pixel in quadrant 0) instead of one pair per quadrant.  d = mean - centre is
// formed once per Gaussian against that pixel and a quadrant takes 8 off the component(s) it is displaced in -- six
// registers fewer per lane in both kernels: the forward kernel at 96 registers was moving a colour accumulator through
// scratch around every batch's cull, and the reload behind the next batch's gather made the walk wait for the gather
// (the hot loop is spill-free now; the kernel's time did not change: 123-125 us either way, r05_k6_k7_variants.txt).
// The forward and the backward kernel and both launch shapes form d with the SAME operations (a one-quadrant wave
// subtracts its offset pair, 0 or 8 per component: x - 0 is exact), so every alpha is bit-identical in all of them.
template <int NQ>
struct PixGrid {
    f2 p0;       // centre of the lane's pixel in quadrant 0 of the tile
    f2 off;      // NQ == 1: (8 or 0, 8 or 0) of the wave's quadrant
};
template <int NQ>
__device__ __forceinline__ f2 quadrant_delta(const PixGrid<NQ>& pg, f2 d0, int q) {
    if constexpr (NQ == 4) {
        f2 d = d0;
        if (q & 1) d.x -= 8.f;
        if (q >> 1) d.y -= 8.f;
        return d;
    } else {
        return d0 - pg.off;
    }
}

// get_outputs' post-processing folded into the compositing kernels (model.py:295-297, 304-306; the reference-shaped route):
//   forward:  rgb = clamp(render[:3] + (1 - alpha) background, 0, 1) written beside render, the depth channel copied to its own
//             image, and every wave's largest depth parked for the fix-up pass that follows (depth = alpha > 0 ? d : max d);
//   backward: v_render / v_alpha derived per pixel from v_rgb / v_depth in the tile prologue -- no separate pass over the image.
struct FwdPost {
    const float* bg;        // [3]; NULL: no post-processing outputs
    float* rgb;             // [C,H,W,3]
    float* depth;           // [C,H,W] (NULL with 3 channels)
    float* tile_dmax;       // [C*tiles][4]
};
struct BwdPost {
    const float* bg;        // [3]; NULL: v_render / v_alpha are given
    const float* render;    // [C,H,W,CH] the forward pass's render
    const float* v_rgb;     // [C,H,W,3] (may be NULL: no gradient)
    const float* v_depth;   // [C,H,W]   (may be NULL)
};

struct FwdPixel {
    float T;
    f2 out01, out23;
    int cur;
};

// Two forms of the per-pixel alpha, chosen per GAUSSIAN (wave-uniform, from the record alone, so the forward and the
// backward kernel take the same form for the same Gaussian):
//   general  a = min(0.999, o exp2(p)), skipped where p > 0 (sigma < 0) -- Appendix A.6 as written;
//   FAST     a = exp2(p + log2 o): for a Gaussian with o <= 0.998 and a conic that is positive definite by a margin
//            (b^2 <= 0.998 a c, so sigma >= 0.0005 (a dx^2 + c dy^2): no rounding of the fp32 evaluation can make it
//            negative) the clamp can never bite and sigma < 0 can never happen: the v_min, the multiply by the
//            opacity (folded into the exponent: the last add of the polynomial becomes the addend of an fma) and one
//            compare per pixel go.  Every Gaussian of an ordinary scene takes this form (sigmoid(4) = 0.982).
// `lo` carries the opacity (general form) or log2 of it (fast form).
__device__ __forceinline__ bool gaussian_is_fast(float ca, float cb, float cc, float op) {
    return (op <= 0.998f) & (cb * cb <= 0.998f * (ca * cc));       // (false for NaNs: they take the general form)
}
// MIXED = false: every Gaussian of the batch takes the fast form (no test at all); true: `slow` (wave-uniform) decides.
template <int CH, bool MIXED>
__device__ __forceinline__ void fwd_quadrant(f2 d, f2 R0, f2 R1, float lo, f2 col01, f2 col23, int idx_v,
                                             unsigned slow, u64& done, FwdPixel& s) {
    const f2 md = conic_times_d(R0, R1, d);
    float a;
    u64 m_ok;
    // (a scalar branch around the alpha, the rest is shared.  The empty asm keeps the flag a 32-bit scalar at every
    // use: hoisted into ONE lane-mask bool the compiler re-materialised it through a VGPR per quadrant)
    if constexpr (MIXED) asm volatile("" : "+s"(slow));
    if (!MIXED || __builtin_expect(slow == 0, 1)) {
        a = __builtin_amdgcn_exp2f(__builtin_fmaf(d.x, md.x, __builtin_fmaf(d.y, md.y, lo)));
        m_ok = __ballot(a >= kAlphaMin) & ~done;
    } else {
        const float p = __builtin_fmaf(d.x, md.x, d.y * md.y);
        a = fminf(kAlphaMax, lo * __builtin_amdgcn_exp2f(p));
        m_ok = __ballot(p <= 0.f) & __ballot(a >= kAlphaMin) & ~done;
    }
    const float at = a * s.T;
    const float nT = s.T - at;
    const u64 m_term = m_ok & __ballot(nT <= kTMin);
    done |= m_term;
    const u64 m_acc = m_ok & ~m_term;
    QED_STAT(5, __builtin_popcountll(m_acc)); QED_STAT(6, m_acc == 0 ? 1 : 0);
    const float w = sel(m_acc, at, 0.f);
    const f2 ww = {w, w};
    s.out01 += col01 * ww;
    if constexpr (CH == 4) s.out23 += col23 * ww;
    else s.out23.x += col23.x * w;
    s.T -= w;                                           // = nT where accepted, unchanged elsewhere
    s.cur = sel(m_acc, idx_v, s.cur);
}

// ---- the walk over one staged batch's surviving Gaussians (forward) -------------------------------------------------
struct FwdRec { float4 q0, q1, q2; };                   // {x, y, k a, k b | k b, k c, r, g | b, depth, opacity or its log2, -}

__device__ __forceinline__ void fwd_fetch(FwdRec& r, const float (*s_rec)[kRecFloats], int tt) {
    // broadcast Gaussian tt: every lane reads the same LDS record
    r.q0 = *reinterpret_cast<const float4*>(&s_rec[tt][0]);
    r.q1 = *reinterpret_cast<const float4*>(&s_rec[tt][4]);
    r.q2 = *reinterpret_cast<const float4*>(&s_rec[tt][8]);
}

// composite Gaussian t (record r): the quadrants whose mask holds it, in turn.  A quadrant that finishes is dropped from
// the masks, which can take later Gaussians out of km.
template <int CH, int NQ, bool MIXED>
__device__ __forceinline__ void fwd_gaussian(const FwdRec& r, int t, u64 bit, u64& km, u64 (&mq)[NQ], const PixGrid<NQ>& pg,
                                             FwdPixel (&px)[NQ], u64 (&done)[NQ], int batch_start, u64 m_slow, int& n_vis) {
    const f2 XY = {r.q0.x, r.q0.y}, R0 = {r.q0.z, r.q0.w}, R1 = {r.q1.x, r.q1.y}, col01 = {r.q1.z, r.q1.w};
    const f2 d0 = XY - pg.p0;
    const f2 col23 = {r.q2.x, CH == 4 ? r.q2.y : 0.f};
    QED_STAT(4, 1);
    int idx_v;                                          // one VGPR copy per Gaussian, not per quadrant
    asm volatile("v_mov_b32 %0, %1" : "=v"(idx_v) : "s"(batch_start + t));
    const unsigned slow = MIXED ? __builtin_amdgcn_readfirstlane((unsigned)(m_slow >> t) & 1u) : 0u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!(mq[q] & bit)) continue;                   // wave-uniform: this quadrant cannot see Gaussian t
        QED_STAT(3, 1);
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(n_vis));              // (a scalar counter: as `++n_vis` it took a VGPR and spilled)
        fwd_quadrant<CH, MIXED>(quadrant_delta<NQ>(pg, d0, q), R0, R1, r.q2.z, col01, col23, idx_v, slow, done[q], px[q]);
        if (done[q] == ~0ull) {                         // quadrant finished: drop it from the masks
            mq[q] = 0;
            u64 m = 0;
#pragma unroll
            for (int k = 0; k < NQ; ++k) m |= mq[k];
            km &= m;
        }
    }
}

// The record is read when its Gaussian's turn comes: the other resident waves hide the LDS latency.  (Requested one
// Gaussian ahead into a second register set, the walk unrolled by two, the kernel is SLOWER -- 136-152 us against 123-130
// in round 5, 154 against 150 in round 3: the scalar bookkeeping of the look-ahead costs more than the latency it hides,
// and 80 registers per lane at six waves per SIMD leave no room for it.  profiles/r05_k6_k7_variants.txt)
template <int CH, int NQ, bool MIXED>
__device__ __forceinline__ void fwd_walk(u64& km, u64 (&mq)[NQ], const PixGrid<NQ>& pg, FwdPixel (&px)[NQ], u64 (&done)[NQ],
                                         int batch_start, u64 m_slow, int& n_vis, const float (*s_rec)[kRecFloats]) {
    while (km) {
        const int t = __builtin_ctzll(km);
        const u64 bit = 1ull << t;
        km &= ~bit;
        FwdRec r;
        fwd_fetch(r, s_rec, t);
        fwd_gaussian<CH, NQ, MIXED>(r, t, bit, km, mq, pg, px, done, batch_start, m_slow, n_vis);
    }
}

template <int NQ>
struct FwdWalkState { u64 km; int n_vis; u64 mq[NQ], done[NQ]; FwdPixel px[NQ]; PixGrid<NQ> pg; };

template <int CH, int NQ>
__device__ __attribute__((noinline)) void fwd_walk_mixed(FwdWalkState<NQ>* st, int batch_start, u64 m_slow,
                                                         const float (*s_rec)[kRecFloats]) {
    FwdWalkState<NQ> w = *st;
    // (arguments and loaded values arrive in vector registers: the wave-uniform ones back into scalar registers)
    w.km = uniform_u64(w.km);
    w.n_vis = __builtin_amdgcn_readfirstlane(w.n_vis);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { w.mq[q] = uniform_u64(w.mq[q]); w.done[q] = uniform_u64(w.done[q]); }
    fwd_walk<CH, NQ, true>(w.km, w.mq, w.pg, w.px, w.done, __builtin_amdgcn_readfirstlane(batch_start), uniform_u64(m_slow),
                           w.n_vis, s_rec);
    *st = w;
}

// ================================================================================================
// forward
// ================================================================================================
// One wave composites either a whole 16x16 tile (NQ = 4: four pixels per lane, one per 8x8 quadrant) or a
// single quadrant q0 of it (NQ = 1: one pixel per lane).
template <int CH, int NQ>
__device__ __forceinline__ void fwd_tile(int tile, int qf, float (*s_rec)[kRecFloats], int C, const float4* __restrict__ splats,
                                         const int* __restrict__ flatten_ids, const int* __restrict__ offsets, int width,
                                         int height, int tile_w, int tile_h, const float* __restrict__ backgrounds,
                                         float* __restrict__ render, float* __restrict__ alpha_out,
                                         float* __restrict__ t_final, int* __restrict__ last_ids,
                                         int* __restrict__ tile_cost, const FwdPost& post) {
    const int q0 = qf & 3;                               // bit 2 of qf: culling off (test hook)
    int n_vis = 0;                                       // quadrant visits of this wave (wave-uniform): K7's work predictor
    const int n_tiles = tile_w * tile_h;
    const int cam = tile / n_tiles;
    const int t_in = tile - cam * n_tiles;
    const int ty = t_in / tile_w, tx = t_in - ty * tile_w;
    const int lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    const float ox = (float)(tx * QED_TILE), oy = (float)(ty * QED_TILE);

    const int start = offsets[tile], end = offsets[tile + 1];
    const int nb = (end - start + kBatch - 1) / kBatch;
#ifdef QED_TILE_TIMING
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif

    PixGrid<NQ> pg;                                     // pixel centres of this lane's NQ pixels
    pg.p0 = (f2){(float)(tx * QED_TILE + lx) + 0.5f, (float)(ty * QED_TILE + ly) + 0.5f};
    pg.off = (f2){(float)((q0 & 1) << 3), (float)((q0 >> 1) << 3)};
    FwdPixel px[NQ];
    u64 done[NQ];                                       // wave-uniform masks
    bool inside[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int qq = NQ == 4 ? q : q0;
        const int ix = tx * QED_TILE + ((qq & 1) << 3) + lx, iy = ty * QED_TILE + ((qq >> 1) << 3) + ly;
        inside[q] = ix < width && iy < height;
        done[q] = __ballot(!inside[q]);
        px[q].T = 1.f;
        px[q].cur = 0;
        px[q].out01 = (f2){0.f, 0.f};
        px[q].out23 = (f2){0.f, 0.f};
    }

    // Two-deep software pipeline over the batches: while batch b is composited the records of b+1 and the ids of b+2
    // are in flight (fetched back to back, the id -> record dependency cost one exposed memory latency per batch).
    // Every load is UNCONDITIONAL from a clamped index (lanes past the end re-read the tile's last entry: one cache
    // line) into registers of its own that are rotated in at the end of the batch: a load under a per-lane
    // condition is merged with the old value right behind it, which makes the wave wait for it on the spot.
    auto id_at = [&](int idx) { return flatten_ids[idx < end ? idx : end - 1]; };
    int rid_n = id_at(start + kBatch + lane);
    float4 r0, r1;
    f3 r2;
    {
        const size_t g = (size_t)id_at(start + lane);
        r0 = splats[3 * g]; r1 = splats[3 * g + 1]; r2 = load_rec_tail(splats, g);
    }
    bool present = start + lane < end;
    auto and_done = [&]() { u64 m = ~0ull;
#pragma unroll
        for (int q = 0; q < NQ; ++q) m &= done[q];
        return m; };
    auto or_masks = [&](const u64* mq) { u64 m = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) m |= mq[q];
        return m; };
    bool all_done = and_done() == ~0ull;
    for (int b = 0; b < nb && !all_done; ++b) {
        // ---- stage this lane's Gaussian: cull per quadrant, pre-scale the conic by -log2(e) ----
        u64 mq[NQ];
        quadrant_masks<NQ>(present, r0, r1, r2.z, ox, oy, qf, mq);
#pragma unroll
        for (int q = 0; q < NQ; ++q) mq[q] = done[q] == ~0ull ? 0ull : uniform_u64(mq[q]);
        u64 km = or_masks(mq);
        QED_STAT(0, 1); QED_STAT(1, min(kBatch, end - (start + b * kBatch))); QED_STAT(2, __builtin_popcountll(km));
        // which per-pixel form this Gaussian takes (see fwd_quadrant); the record carries log2(opacity) for the fast one
        const bool fast = gaussian_is_fast(r0.z, r0.w, r1.x, r1.y);
        const u64 m_slow = uniform_u64(__ballot(!fast));
        if (km) {                                       // park this lane's (pre-scaled) Gaussian for the broadcast reads:
            const float bh = kConicScale * r0.w;        // {x, y, k a, k b | k b, k c, r, g | b, depth, opacity or its log2, -}
            *reinterpret_cast<float4*>(&s_rec[lane][0]) = make_float4(r0.x, r0.y, kConicScale * r0.z, bh);
            *reinterpret_cast<float4*>(&s_rec[lane][4]) = make_float4(bh, kConicScale * r1.x, r1.z, r1.w);
            *reinterpret_cast<float4*>(&s_rec[lane][8]) = make_float4(r2.x, r2.y, fast ? __builtin_amdgcn_logf(r1.y) : r1.y, 0.f);
        }
        __syncthreads();                                // single wave: orders the stores before the reads below
        // issue the gather of the next batch (its ids arrived a batch ago) and the id load of the one after
        const size_t g_n = (size_t)rid_n;
        const float4 n0 = splats[3 * g_n], n1 = splats[3 * g_n + 1];
        const f3 n2 = load_rec_tail(splats, g_n);
        const int rid_nn = id_at(start + (b + 2) * kBatch + lane);
        const int batch_start = start + b * kBatch;
        // the walk over the batch's survivors (fwd_walk): two copies -- a batch whose Gaussians all take the fast form
        // (every batch of an ordinary scene) runs without any per-visit test of the form
#if QED_K6_FORM == 2
        if (m_slow == 0) fwd_walk<CH, NQ, false>(km, mq, pg, px, done, batch_start, m_slow, n_vis, s_rec);
        else fwd_walk<CH, NQ, true>(km, mq, pg, px, done, batch_start, m_slow, n_vis, s_rec);
#elif QED_K6_FORM == 4
        if (__builtin_expect(m_slow == 0, 1)) {
            fwd_walk<CH, NQ, false>(km, mq, pg, px, done, batch_start, m_slow, n_vis, s_rec);
        } else {
            // a batch with a slow-form Gaussian (rare): the walk that tests the form per Gaussian is an OUT-OF-LINE function
            // on a COPY of the state -- inlined beside the fast walk it costs the hot loop its scalar registers, and handing
            // it the state itself would pin that state in memory for the whole kernel
            FwdWalkState<NQ> st;
            st.km = km; st.n_vis = n_vis;
            st.pg = pg;
#pragma unroll
            for (int q = 0; q < NQ; ++q) { st.mq[q] = mq[q]; st.done[q] = done[q]; st.px[q] = px[q]; }
            fwd_walk_mixed<CH, NQ>(&st, batch_start, m_slow, s_rec);
            km = uniform_u64(st.km); n_vis = __builtin_amdgcn_readfirstlane(st.n_vis);
#pragma unroll
            for (int q = 0; q < NQ; ++q) { mq[q] = uniform_u64(st.mq[q]); done[q] = uniform_u64(st.done[q]); px[q] = st.px[q]; }
            // the loads of this copy-back retire HERE: left pending at the join, the fast path inherits "a register of the
            // walk may still be the target of a load" and its first Gaussian waits for vmcnt(1) -- the counter retires in
            // order, so that is the next batch's gather (s_waitcnt vmcnt(0), expcnt / lgkmcnt untouched)
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
#else
        fwd_walk<CH, NQ, true>(km, mq, pg, px, done, batch_start, m_slow, n_vis, s_rec);
#endif
        all_done = and_done() == ~0ull;                 // (a finished quadrant empties its mask, so km ran out by itself)
        r0 = n0; r1 = n1; r2 = n2;                      // rotate the pipeline (the wave waits HERE, not above)
        rid_n = rid_nn;
        present = start + (b + 1) * kBatch + lane < end;
    }
    float post_dmax = -3.0e38f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (inside[q]) {
            const int qq = NQ == 4 ? q : q0;
            const int ix = tx * QED_TILE + ((qq & 1) << 3) + lx, iy = ty * QED_TILE + ((qq >> 1) << 3) + ly;
            const size_t pix = ((size_t)cam * height + iy) * width + ix;
            float out[4] = {px[q].out01.x, px[q].out01.y, px[q].out23.x, px[q].out23.y};
            if (backgrounds != nullptr) {
#pragma unroll
                for (int k = 0; k < CH; ++k) out[k] += px[q].T * backgrounds[cam * CH + k];
            }
            if constexpr (CH == 4) {
                *reinterpret_cast<float4*>(render + 4 * pix) = make_float4(out[0], out[1], out[2], out[3]);
            } else {
                render[3 * pix] = out[0]; render[3 * pix + 1] = out[1]; render[3 * pix + 2] = out[2];
            }
            const float a_out = 1.f - px[q].T;
            alpha_out[pix] = a_out;
            // the transmittance itself for the backward pass: 1 - alpha loses up to 3e-8 / T of it (alpha sits just below 1
            // wherever the pixel saturated), and every gradient term of the pixel scales with T
            if (t_final != nullptr) t_final[pix] = px[q].T;
            last_ids[pix] = px[q].cur;
            if (post.bg != nullptr) {
                const float om = 1.f - a_out;            // (as the separate pass forms it from the stored alpha)
                Float3 c;
                c.a = fminf(fmaxf(out[0] + om * post.bg[0], 0.f), 1.f);
                c.b = fminf(fmaxf(out[1] + om * post.bg[1], 0.f), 1.f);
                c.c = fminf(fmaxf(out[2] + om * post.bg[2], 0.f), 1.f);
                *reinterpret_cast<Float3*>(post.rgb + 3 * pix) = c;
                if constexpr (CH == 4) { post.depth[pix] = out[3]; post_dmax = fmaxf(post_dmax, out[3]); }
            }
        }
    }
    if (post.bg != nullptr && CH == 4) {
        post_dmax = wave_max(post_dmax);
        if constexpr (NQ == 4) { if (lane < 4) post.tile_dmax[4 * tile + lane] = lane == 0 ? post_dmax : -3.0e38f; }
        else { if (lane == 0) post.tile_dmax[4 * tile + q0] = post_dmax; }
    }
    // What the backward pass will cost on this tile: its quadrant visits are (within a per cent) the forward pass's, plus a
    // staging term per batch.  [tile][quadrant] so that whole-tile and quadrant waves both write without initialisation.
    if (tile_cost != nullptr) {
        if constexpr (NQ == 4) { if (lane < 4) tile_cost[4 * tile + lane] = lane == 0 ? n_vis + 2 * nb : 0; }
        else { if (lane == 0) tile_cost[4 * tile + q0] = n_vis + nb; }
    }
#ifdef QED_TILE_TIMING
    // diagnostic build only: duration, start, HW_ID, XCC_ID of this tile's wave in its first four alphas
    if (lane == 0 && NQ == 4) {
        const size_t pix = ((size_t)cam * height + ty * QED_TILE) * width + tx * QED_TILE;
        alpha_out[pix] = (float)(__builtin_amdgcn_s_memtime() - t_start);
        alpha_out[pix + 1] = (float)(t_start & 0xFFFFFFFull);
        alpha_out[pix + 2] = (float)(__builtin_amdgcn_s_getreg(63492) & 0xFFFF);
        alpha_out[pix + 3] = (float)(__builtin_amdgcn_s_getreg(63508) & 0xF);
    }
#endif
}

// Quadrant waves: block r of the tail -> (tile, quadrant).  Workgroups are dealt round-robin over the 8 XCDs, so
// the four waves of a tile get block indices that are congruent mod 8 (r = 32 j + 8 q + x  ->  tile 8 j + x):
// they run on the same XCD at about the same time and share its L2 copy of the tile's records (r -> (r >> 2,
// r & 3) put them on four XCDs and tripled the kernel's HBM fetch).
__device__ __forceinline__ void small_wave(int r, int n_small, int& t, int& q) {
    const int full = n_small & ~7;                       // tiles in whole groups of 8
    if (r < 4 * full) {
        t = ((r >> 5) << 3) | (r & 7);
        q = (r >> 3) & 3;
    } else {
        const int rr = r - 4 * full, rem = n_small - full;
        q = rr / rem;
        t = full + rr - q * rem;
    }
}

// Launch shape: the first n_big positions of the (XCD-remapped) tile order are composited by one wave each;
// every later tile by FOUR waves, one per quadrant.  Workgroups are dispatched in index order, so the
// quarter-size work items arrive last and fill the end of the launch, where whole-tile waves would leave most
// wave slots idle (measured: 2.7 of 5 resident waves per SIMD on average with whole tiles only).
// (six waves per SIMD leave the forward kernel 80 registers per lane: with two copies of the walk it spilled 92 bytes per
// lane per batch -- +38 MB written and +40 MB fetched per launch, profiles/r05_hbm_traffic_pmc.json; five waves = 96
// registers, no scratch, and the same time within the spread: 127.9 vs 130.4 us, profiles/r05_k6_k7_variants.txt)
#ifndef QED_K6_WAVES
#define QED_K6_WAVES 5
#endif
// quadrant waves at the end of a launch, in units of the device's wave slots for the kernel (see big_tiles())
#ifndef QED_K6_SMALL
#define QED_K6_SMALL 2.5
#endif
#ifndef QED_K7_SMALL
#define QED_K7_SMALL 1.2
#endif
template <int CH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QED_K6_WAVES, QED_K6_WAVES)))
composite_fwd_kernel(int C, const float4* __restrict__ splats, const int* __restrict__ flatten_ids,
                     const int* __restrict__ offsets, int width, int height, int tile_w, int tile_h,
                     const float* __restrict__ backgrounds, float* __restrict__ render, float* __restrict__ alpha_out,
                     float* __restrict__ t_final, int* __restrict__ last_ids, int* __restrict__ tile_cost, int n_big_flags,
                     FwdPost post, const int* __restrict__ tile_order) {
    __shared__ __attribute__((aligned(16))) float s_rec[kBatch][kRecFloats];
    const int n_total = C * tile_w * tile_h;
    const int b = blockIdx.x;
    const int keep_all = (n_big_flags >> 30) << 2;      // test hook, see quadrant_masks
    const int n_big = n_big_flags & 0x3fffffff;
    if (tile_order != nullptr) {
        // A costliest-first order handed in by the caller: the launch order the compositing BACKWARD of an earlier frame of
        // the same camera was given (the forward pass's own work counts of that frame: they predict this frame's, and this
        // kernel cannot know its costs in advance -- the list length does not predict them, corr -0.02 at config B).  As in
        // composite_bwd_kernel: the n_split heaviest tiles as four quadrant waves each ahead of everything, then whole tiles
        // in order of decreasing cost.  Any permutation gives the same image; a poor predictor only costs time.
        // Measured at config B (scripts/xcd_order_ab.py, profiles/r05_bench_kernel_stats_v6.txt): 108-111 us static, 119 in the
        // training run, against 122-126 / 129-133 with the raster order + quadrant tail.
        const int n_split = min(max(tile_order[n_total], 0), n_total / 8);
        if (b < 4 * n_split) {
            fwd_tile<CH, 1>(min(max(tile_order[b >> 2], 0), n_total - 1), (b & 3) | keep_all, s_rec, C, splats, flatten_ids, offsets,
                            width, height, tile_w, tile_h, backgrounds, render, alpha_out, t_final, last_ids, tile_cost, post);
            return;
        }
        const int i = b - 3 * n_split;
        if (i >= n_total) return;                        // (the grid is sized for the largest n_split the host allows)
        // (ids clamped: a buffer that is not an order of THIS grid must not fault the device; the caller's contract is a
        // permutation -- the image is wrong otherwise)
        fwd_tile<CH, 4>(min(max(tile_order[i], 0), n_total - 1), keep_all, s_rec, C, splats, flatten_ids, offsets, width, height, tile_w, tile_h,
                        backgrounds, render, alpha_out, t_final, last_ids, tile_cost, post);
        return;
    }
    if (b < n_big) {
        fwd_tile<CH, 4>(xcd_remap(b, n_total), keep_all, s_rec, C, splats, flatten_ids, offsets, width, height, tile_w,
                        tile_h, backgrounds, render, alpha_out, t_final, last_ids, tile_cost, post);
    } else {
        int t, q;
        small_wave(b - n_big, n_total - n_big, t, q);
        fwd_tile<CH, 1>(xcd_remap(n_big + t, n_total), q | keep_all, s_rec, C, splats, flatten_ids, offsets, width, height,
                        tile_w, tile_h, backgrounds, render, alpha_out, t_final, last_ids, tile_cost, post);
    }
}

// ================================================================================================
// backward
// ================================================================================================
// per-Gaussian gradient accumulators of one lane (summed over its four pixels)
struct GradAcc {
    f2 vxy;        // 0, 1   v_x, v_y
    float ax, ay;  // 2, 3   |v_x|, |v_y|
    f2 c01;        // 4, 5   sum v_sigma dx^2 (x 1/2 at flush), sum v_sigma dx dy
    float c2;      // 6      sum v_sigma dy^2 (x 1/2 at flush)
    float s0;      // 7      sum v_sigma      (x -1/opacity at flush)
    f2 rg, bd;     // 8..11  v_r, v_g, v_b, v_depth
};

struct BwdPixel {
    float T, bufv;
    f2 vr01, vr23;
    int bin_final;
};

// MIXED / slow / lo: as fwd_quadrant.
template <int CH, bool MIXED>
__device__ __forceinline__ void bwd_quadrant(f2 d, f2 R0, f2 R1, float lo, f2 col01, f2 col23, int idx,
                                             unsigned slow, BwdPixel& s, u64& any_valid, GradAcc& g) {
    const f2 w = conic_times_d(R0, R1, d);
    float opv, a;
    u64 m_valid, m_vs;
    if constexpr (MIXED) asm volatile("" : "+s"(slow));
    if (!MIXED || __builtin_expect(slow == 0, 1)) {
        opv = a = __builtin_amdgcn_exp2f(__builtin_fmaf(d.x, w.x, __builtin_fmaf(d.y, w.y, lo)));
        m_vs = m_valid = __ballot(s.bin_final >= idx) & __ballot(a >= kAlphaMin);
    } else {
        const float p = __builtin_fmaf(d.x, w.x, d.y * w.y);
        opv = lo * __builtin_amdgcn_exp2f(p);
        a = fminf(kAlphaMax, opv);
        m_valid = __ballot(s.bin_final >= idx) & __ballot(p <= 0.f) & __ballot(a >= kAlphaMin);
        m_vs = m_valid & __ballot(opv <= kAlphaMax);
    }
    any_valid |= m_valid;
    QED_STAT(14, __builtin_popcountll(m_valid)); QED_STAT(15, m_valid == 0 ? 1 : 0);
    // branch-free: an invalid pixel contributes zeros and keeps its state -- through ONE select: with alpha = 0 the
    // reciprocal is v_rcp_f32(1.0) = 1.0 exactly (scripts/ubench/rcp_one.hip checks it on the device), so T * 1 is T bit
    // for bit and the weight is 0 * T = 0
    const float a_eff = sel(m_valid, a, 0.f);
    // (v_rcp_f32 is good to an ulp; a Newton step on it changed no gradient beyond the seventh digit even in scenes that
    // stack hundreds of layers per pixel -- scripts/dense_scene_diag.py.  What did limit the backward pass there was
    // T_final = 1 - alpha, see bwd_tile)
    const float ra = __builtin_amdgcn_rcpf(1.f - a_eff);
    const float Tn = s.T * ra;
    s.T = Tn;
    const float fac = a_eff * Tn;
    const f2 ff = {fac, fac};
    f2 cvv = col01 * s.vr01;
    if constexpr (CH == 4) cvv += col23 * s.vr23;
    else cvv.x += col23.x * s.vr23.x;
    const float cv = cvv.x + cvv.y;
    g.rg += s.vr01 * ff;
    if constexpr (CH == 4) g.bd += s.vr23 * ff;
    else g.bd.x += s.vr23.x * fac;
    const float v_a = __builtin_fmaf(Tn, cv, -(ra * s.bufv));      // mul + fma (the packed form needs a register copy)
    s.bufv = __builtin_fmaf(fac, cv, s.bufv);
    // v_sigma = -o e^(-sigma) v_alpha where the pixel is valid and alpha not clamped.  In the fast form that is every valid
    // pixel, and a_eff already is o e^(-sigma) there and 0 elsewhere: no select
    float vs;
    if (!MIXED || __builtin_expect(slow == 0, 1)) vs = -a_eff * v_a;
    else vs = sel(m_vs, -opv * v_a, 0.f);
    const f2 vv = {vs, vs};
    const f2 sv = d * vv;                               // (v_sigma dx, v_sigma dy)
    g.vxy += w * vv;                                    // k (M d) v_sigma = k x the gradient of the mean (1 / k at the flush)
    // its absolute value without forming it: |w v_sigma| = |w| |v_sigma| bit for bit, as source modifiers of an fma
    asm("v_fma_f32 %0, |%1|, |%2|, %0" : "+v"(g.ax) : "v"(w.x), "v"(vs));
    asm("v_fma_f32 %0, |%1|, |%2|, %0" : "+v"(g.ay) : "v"(w.y), "v"(vs));
    g.c01 += d * (f2){sv.x, sv.x};                      // (sx dx, sx dy)
    g.c2 = __builtin_fmaf(sv.y, d.y, g.c2);
    g.s0 += vs;
}

// Flush of the parked Gaussians: the lanes of a (parked Gaussian, value) add its partials in a fixed order, scale, and
// the wave issues one 64-byte row per Gaussian in a single atomic request.
// the values leave the wave in the units they were accumulated in; one lane per (Gaussian, value) converts:
//   0, 1  k x v_mean -> x 1/k;   2, 3  |k| x |v_mean| -> x -1/k;   4, 6  sum v_sigma d^2 -> x 1/2;
//   7     sum v_sigma -> x -1/opacity (the record holds the opacity, or log2 of it for a fast-form Gaussian)
__device__ __forceinline__ float flush_scale(int k, float v, float lo, bool slow) {
    if (k < 2) v *= 1.f / kConicScale;
    else if (k < 4) v *= -1.f / kConicScale;
    else if (k == 4 || k == 6) v *= 0.5f;
    else if (k == 7) v = -v * (slow ? __builtin_amdgcn_rcpf(lo) : __builtin_amdgcn_exp2f(-lo));  // (an IEEE division is 11 instructions)
    return v;
}

__device__ __forceinline__ void flush_parked(int n_parked, int pt0, int pt1, int /*pt2*/, u64 m_slow,
                                             const float* __restrict__ s_part, const float (*s_rec)[kRecFloats],
                                             float* __restrict__ vsplat, int lane) {
    QED_STAT(13, 1);
    __syncthreads();                                    // single wave: orders the parking stores before these reads
    const int row = lane >> 1, half = lane & 1;         // row = 12 slot + value; lanes 2 row, 2 row + 1 share it
    const int slot = row >= 12 ? 1 : 0, k = row - 12 * slot;
    const bool mine = row < 12 * n_parked;
    float v = 0.f;
    if (mine) {
        const float4* src = reinterpret_cast<const float4*>(s_part + row * kParkStride + 32 * half);
        const float4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3], a4 = src[4], a5 = src[5], a6 = src[6], a7 = src[7];
        const float4 b0 = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        const float4 b1 = make_float4(a2.x + a3.x, a2.y + a3.y, a2.z + a3.z, a2.w + a3.w);
        const float4 b2 = make_float4(a4.x + a5.x, a4.y + a5.y, a4.z + a5.z, a4.w + a5.w);
        const float4 b3 = make_float4(a6.x + a7.x, a6.y + a7.y, a6.z + a7.z, a6.w + a7.w);
        const float4 c0 = make_float4(b0.x + b1.x, b0.y + b1.y, b0.z + b1.z, b0.w + b1.w);
        const float4 c1 = make_float4(b2.x + b3.x, b2.y + b3.y, b2.z + b3.z, b2.w + b3.w);
        v = ((c0.x + c1.x) + (c0.y + c1.y)) + ((c0.z + c1.z) + (c0.w + c1.w));
    }
    // the other half's 32 partials: lanes l and l ^ 1 (quad_perm [1, 0, 3, 2]); every lane takes part
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    if (mine && half == 0) {
        const int tg = slot == 0 ? pt0 : pt1;
        const int id = __float_as_int(s_rec[tg][11]);
        v = flush_scale(k, v, s_rec[tg][10], (m_slow >> tg) & 1);
        if (v != 0.f) atomicAdd(&vsplat[(size_t)id * QED_VSPLAT_FLOATS + k], v);
    }
}

// ---- the walk over one staged batch's surviving Gaussians (plain functions, not lambdas: a closure that refers to another
// closure kept every local they captured in scratch memory) -----------------------------------------------------------
struct ParkState { int n, t0, t1, t2; };                // parked Gaussians (lane indices of this batch)
struct BwdRec { float4 q0, q1, q2; };                   // one Gaussian's record as the loop's register pairs (bwd_tile)

__device__ __forceinline__ void bwd_fetch(BwdRec& r, const float (*s_rec)[kRecFloats], int tt) {
    r.q0 = *reinterpret_cast<const float4*>(&s_rec[tt][0]);
    r.q1 = *reinterpret_cast<const float4*>(&s_rec[tt][4]);
    r.q2 = *reinterpret_cast<const float4*>(&s_rec[tt][8]);
}

// the pixels of Gaussian t (record r), then its twelve sums parked for the flush
template <int CH, int NQ, bool MIXED>
__device__ __forceinline__ void bwd_gaussian(BwdRec& r, int t, u64 bit, u64 km, const u64 (&mq)[NQ], const PixGrid<NQ>& pg,
                                             BwdPixel (&px)[NQ], int batch_hi, u64 m_slow, ParkState& park,
                                             float* park_lane, const float* __restrict__ s_part,
                                             const float (*s_rec)[kRecFloats], float* __restrict__ vsplat, int lane) {
    const f2 XY = {r.q0.x, r.q0.y}, R0 = {r.q0.z, r.q0.w}, R1 = {r.q1.x, r.q1.y};
    const f2 col23 = {r.q1.z, CH == 4 ? r.q1.w : 0.f}, col01 = {r.q2.x, r.q2.y};
    const float lo = r.q2.z;
    const f2 d0 = XY - pg.p0;
    const int idx = batch_hi - t;
    const unsigned slow = MIXED ? __builtin_amdgcn_readfirstlane((unsigned)(m_slow >> t) & 1u) : 0u;
    GradAcc g;
    g.vxy = g.c01 = g.rg = g.bd = (f2){0.f, 0.f};
    g.ax = g.ay = g.c2 = g.s0 = 0.f;
    u64 any_valid = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!(mq[q] & bit)) continue;                   // wave-uniform: this quadrant cannot see Gaussian t
        QED_STAT(11, 1);
        bwd_quadrant<CH, MIXED>(quadrant_delta<NQ>(pg, d0, q), R0, R1, lo, col01, col23, idx, slow, px[q], any_valid, g);
    }
    if (any_valid == 0) return;
    QED_STAT(12, 1);
    {
        float gv[12] = {g.vxy.x, g.vxy.y, g.ax, g.ay, g.c01.x, g.c01.y, g.c2, g.s0, g.rg.x, g.rg.y, g.bd.x, g.bd.y};
        float* dst = park_lane + park.n * kParkSlot;
#pragma unroll
        for (int i = 0; i < 12; ++i) dst[i * kParkStride] = gv[i];
    }
    // (selects, not an if-chain: the compiler turned the chain into an indexed store to the struct -- in scratch memory)
    park.t0 = park.n == 0 ? t : park.t0;
    park.t1 = park.n == 1 ? t : park.t1;
    if constexpr (kParkSlots > 2) park.t2 = park.n == 2 ? t : park.t2;
    if (++park.n == kParkSlots) {
        flush_parked(park.n, park.t0, park.t1, park.t2, m_slow, s_part, s_rec, vsplat, lane);
        park.n = 0;
    }
}

// The record of a surviving Gaussian is requested one Gaussian AHEAD, into a second set of registers (the walk is
// unrolled by two so that the sets alternate without copies): the LDS latency runs beside the pixels of the current
// Gaussian.  (Rounds 3-4 requested it behind the current one's pixels into the very registers it was read from, where
// the latency ran beside the in-wave reduction, which is gone.  On demand / behind / ahead measure alike within the
// spread, 274-277 us: profiles/r05_k6_k7_variants.txt; this form keeps the fetch clear of compiler-inserted copies.)
template <int CH, int NQ, bool MIXED>
__device__ __forceinline__ void bwd_walk(u64 km, const u64 (&mq)[NQ], const PixGrid<NQ>& pg, BwdPixel (&px)[NQ], int batch_hi,
                                         u64 m_slow, ParkState& park, float* park_lane, const float* __restrict__ s_part,
                                         const float (*s_rec)[kRecFloats], float* __restrict__ vsplat, int lane) {
    BwdRec ra, rb;
    ra.q0 = ra.q1 = ra.q2 = rb.q0 = rb.q1 = rb.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (km) bwd_fetch(ra, s_rec, __builtin_ctzll(km));
    while (km) {
        {
            const int t = __builtin_ctzll(km);
            const u64 bit = 1ull << t;
            km &= ~bit;
            if (km) bwd_fetch(rb, s_rec, __builtin_ctzll(km));
            bwd_gaussian<CH, NQ, MIXED>(ra, t, bit, km, mq, pg, px, batch_hi, m_slow, park, park_lane, s_part, s_rec, vsplat, lane);
        }
        if (!km) break;
        {
            const int t = __builtin_ctzll(km);
            const u64 bit = 1ull << t;
            km &= ~bit;
            if (km) bwd_fetch(ra, s_rec, __builtin_ctzll(km));
            bwd_gaussian<CH, NQ, MIXED>(rb, t, bit, km, mq, pg, px, batch_hi, m_slow, park, park_lane, s_part, s_rec, vsplat, lane);
        }
    }
}

// vsplat row layout (QED_VSPLAT_FLOATS = 16):
//  0 v_x  1 v_y  2 |v_x|  3 |v_y|  4 v_conic_a  5 v_conic_b  6 v_conic_c  7 v_opacity  8 v_r  9 v_g  10 v_b  11 v_depth
// Values 4, 6 and 7 are accumulated un-scaled (sum v_sigma dx^2, sum v_sigma dy^2, sum v_sigma) and
// scaled by 0.5, 0.5 and -1/opacity once per (tile, Gaussian) at flush time.
template <int CH, int NQ>
__device__ __forceinline__ void bwd_tile(int tile, int qf, float (*s_rec)[kRecFloats],
                                         float* __restrict__ s_part, int C,
                                         const float4* __restrict__ splats,
                                         const int* __restrict__ flatten_ids, const int* __restrict__ offsets, int width,
                                         int height, int tile_w, int tile_h, const float* __restrict__ backgrounds,
                                         const float* __restrict__ render_alpha, const float* __restrict__ t_final,
                                         const int* __restrict__ last_ids,
                                         const float* __restrict__ v_render, const float* __restrict__ v_alpha,
                                         float* __restrict__ vsplat, const BwdPost& post) {
    const int q0 = qf & 3;                               // bit 2 of qf: culling off (test hook)
    const int n_tiles = tile_w * tile_h;
    const int cam = tile / n_tiles;
    const int t_in = tile - cam * n_tiles;
    const int ty = t_in / tile_w, tx = t_in - ty * tile_w;
    const int lane = threadIdx.x;
    const int lx = lane & 7, ly = lane >> 3;
    const float ox = (float)(tx * QED_TILE), oy = (float)(ty * QED_TILE);

    const int start = offsets[tile], end = offsets[tile + 1];
    if (end <= start) return;

    PixGrid<NQ> pg;
    pg.p0 = (f2){(float)(tx * QED_TILE + lx) + 0.5f, (float)(ty * QED_TILE + ly) + 0.5f};
    pg.off = (f2){(float)((q0 & 1) << 3), (float)((q0 >> 1) << 3)};
    BwdPixel px[NQ];
    int quad_last[NQ];
    int tile_last = -1;
    // Pixel state, in two steps: first every load of every quadrant from a CLAMPED pixel address (no branch, so all of
    // them are in flight together: one memory round trip per wave instead of one per quadrant -- and, with the
    // post-processing gradient, no load that waits for another's result), then the arithmetic, masked by `inside`.
    bool inside[NQ];
    float a_in[NQ], vra_in[NQ], vd_in[NQ];
    int last_in[NQ];
    float4 r4_in[NQ];                                          // v_render, or the render of the post-processing gradient
    Float3 g3_in[NQ];
    size_t pix_in[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int qq = NQ == 4 ? q : q0;
        const int ix = tx * QED_TILE + ((qq & 1) << 3) + lx, iy = ty * QED_TILE + ((qq >> 1) << 3) + ly;
        inside[q] = ix < width && iy < height;
        const size_t pix = ((size_t)cam * height + min(iy, height - 1)) * width + min(ix, width - 1);
        pix_in[q] = pix;
        a_in[q] = (t_final != nullptr ? t_final : render_alpha)[pix];      // T_final itself when the forward pass kept it
        last_in[q] = last_ids[pix];
        vra_in[q] = 0.f; vd_in[q] = 0.f; g3_in[q] = Float3{0.f, 0.f, 0.f};
    }
    {
        const float* rsrc = post.bg == nullptr ? v_render : post.render;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if constexpr (CH == 4) {
                r4_in[q] = *reinterpret_cast<const float4*>(rsrc + 4 * pix_in[q]);
            } else {
                r4_in[q] = make_float4(rsrc[3 * pix_in[q]], rsrc[3 * pix_in[q] + 1], rsrc[3 * pix_in[q] + 2], 0.f);
            }
        }
    }
    if (post.bg == nullptr) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) vra_in[q] = v_alpha[pix_in[q]];
    } else {
        if (post.v_rgb != nullptr) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) g3_in[q] = *reinterpret_cast<const Float3*>(post.v_rgb + 3 * pix_in[q]);
        }
        if constexpr (CH == 4) {
            if (post.v_depth != nullptr) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) vd_in[q] = post.v_depth[pix_in[q]];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float T_final = 1.f, vra = 0.f;
        float vr[4] = {0.f, 0.f, 0.f, 0.f};
        px[q].bin_final = -1;
        if (inside[q]) {
            // (with T_final given, 1 - T_final is the alpha the forward pass stored, bit for bit: the clamp test below takes
            // the side the forward pass took)
            const float a_px = t_final != nullptr ? 1.f - a_in[q] : a_in[q];
            T_final = t_final != nullptr ? a_in[q] : 1.f - a_px;
            px[q].bin_final = last_in[q];
            if (post.bg == nullptr) {
                vra = vra_in[q];
                vr[0] = r4_in[q].x; vr[1] = r4_in[q].y; vr[2] = r4_in[q].z;
                if constexpr (CH == 4) vr[3] = r4_in[q].w;
            } else {
                // the backward of rgb = clamp(render + (1 - alpha) bg, 0, 1) and depth = alpha > 0 ? d : max d (detached), here
                // instead of in a pass of its own (qed_post_process_bwd's arithmetic)
                const float c[3] = {r4_in[q].x, r4_in[q].y, r4_in[q].z};
                if (post.v_rgb != nullptr) {
                    const float g[3] = {g3_in[q].a, g3_in[q].b, g3_in[q].c};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float pre = c[k] + (1.f - a_px) * post.bg[k];
                        vr[k] = (pre >= 0.f && pre <= 1.f) ? g[k] : 0.f;               // torch.clamp backward (inclusive)
                        vra -= vr[k] * post.bg[k];
                    }
                }
                if constexpr (CH == 4) {
                    if (a_px > 0.f) vr[3] = vd_in[q];
                }
            }
            if (backgrounds != nullptr) {
                // render = sum + T_final * bg  ->  d render / d T_final folds into the alpha gradient
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) acc += backgrounds[cam * CH + k] * vr[k];
                vra -= acc;
            }
        }
        px[q].vr01 = (f2){vr[0], vr[1]};
        px[q].vr23 = (f2){vr[2], vr[3]};
        px[q].T = T_final;
        // v_alpha = sum_k (c_k T - buf_k / (1-a)) v_k + T_final / (1-a) v_render_alpha is evaluated as
        // T (c . v) - (buf . v - T_final v_render_alpha) / (1-a): one scalar accumulator per pixel
        px[q].bufv = -T_final * vra;
        // a pixel that composited nothing has last_id 0 and T_final 1: it only "owns" index 0
        quad_last[q] = __builtin_amdgcn_readfirstlane(wave_max_i(px[q].bin_final));
        tile_last = max(tile_last, quad_last[q]);
    }
    const int eff_end = min(end, tile_last + 1);
    if (eff_end <= start) return;
    const int nb = (eff_end - start + kBatch - 1) / kBatch;

    // batches run back to front; lane l of batch b gathers sorted index (eff_end - 1 - 64 b - l)
    // two-deep pipeline as in the forward pass (unconditional loads from clamped indices, rotated at the end of the
    // batch); lane l of batch b holds sorted index eff_end - 1 - 64 b - l, valid while >= start
    auto id_at = [&](int idx) { return flatten_ids[idx > start ? idx : start]; };
    int rid = id_at(eff_end - 1 - lane);
    int rid_n = id_at(eff_end - 1 - kBatch - lane);
    float4 r0 = splats[3 * (size_t)rid], r1 = splats[3 * (size_t)rid + 1];
    f3 r2 = load_rec_tail(splats, (size_t)rid);
    bool present = eff_end - 1 - lane >= start;
    // everything fetched so far (pixel state, first records, second ids) has landed before the loop: inside it the
    // only loads in flight are the prefetches, and the per-Gaussian body never waits on memory (without this the
    // compiler's conservative loop-carried count put s_waitcnt vmcnt(0) in front of every quadrant body)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int b = 0; b < nb; ++b) {
        const int batch_hi = eff_end - 1 - b * kBatch;            // sorted index gathered by lane 0
        u64 mq[NQ];
        quadrant_masks<NQ>(present, r0, r1, r2.z, ox, oy, qf, mq);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            mq[q] = uniform_u64(mq[q]);
            // lane t holds sorted index batch_hi - t: beyond every pixel of the quadrant -> cannot be valid
            const int t0 = batch_hi - quad_last[q];
            if (t0 >= kBatch) mq[q] = 0;
            else if (t0 > 0) mq[q] &= ~0ull << t0;
        }
        u64 km = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) km |= mq[q];
        QED_STAT(8, 1); QED_STAT(9, min(kBatch, batch_hi + 1 - start)); QED_STAT(10, __builtin_popcountll(km));
        QED_STAT(NQ == 4 ? 16 : 17, b == 0 ? 1 : 0);
        // the per-pixel form of this lane's Gaussian: the forward kernel's decision, from the same record (fwd_quadrant)
        const bool fast = gaussian_is_fast(r0.z, r0.w, r1.x, r1.y);
        const u64 m_slow = uniform_u64(__ballot(!fast));
        if (km) {                                       // park this lane's Gaussian, laid out as the loop's register pairs:
            const float bh = kConicScale * r0.w;        // {x, y, k a, k b | k b, k c, b, depth | r, g, opacity or its log2, id}
            *reinterpret_cast<float4*>(&s_rec[lane][0]) = make_float4(r0.x, r0.y, kConicScale * r0.z, bh);
            *reinterpret_cast<float4*>(&s_rec[lane][4]) = make_float4(bh, kConicScale * r1.x, r2.x, r2.y);
            *reinterpret_cast<float4*>(&s_rec[lane][8]) = make_float4(r1.z, r1.w, fast ? __builtin_amdgcn_logf(r1.y) : r1.y,
                                                                       __int_as_float(rid));
        }
        __syncthreads();
        const float4 n0 = splats[3 * (size_t)rid_n], n1 = splats[3 * (size_t)rid_n + 1];
        const f3 n2 = load_rec_tail(splats, (size_t)rid_n);
        const int rid_nn = id_at(batch_hi - 2 * kBatch - lane);
        float* const park_lane = s_part + lane;         // [slot][value][lane]
        // two copies of the walk, as in the forward kernel: a batch whose Gaussians all take the fast form carries no test
        ParkState park{0, 0, 0, 0};
#if QED_K7_FORM == 2
        if (m_slow == 0) bwd_walk<CH, NQ, false>(km, mq, pg, px, batch_hi, m_slow, park, park_lane, s_part, s_rec, vsplat, lane);
        else bwd_walk<CH, NQ, true>(km, mq, pg, px, batch_hi, m_slow, park, park_lane, s_part, s_rec, vsplat, lane);
#else
        bwd_walk<CH, NQ, true>(km, mq, pg, px, batch_hi, m_slow, park, park_lane, s_part, s_rec, vsplat, lane);
#endif
        const int n_parked = park.n, pt0 = park.t0, pt1 = park.t1, pt2 = park.t2;
        if (n_parked)                                   // before the next batch overwrites s_rec (ids, opacities)
            flush_parked(n_parked, pt0, pt1, pt2, m_slow, s_part, s_rec, vsplat, lane);
        r0 = n0; r1 = n1; r2 = n2;                      // rotate the pipeline
        rid = rid_n; rid_n = rid_nn;
        present = batch_hi - kBatch - lane >= start;
    }
}

// same launch shape as composite_fwd_kernel: whole-tile waves first, quadrant waves for the last tiles
template <int CH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QED_K7_WAVES, QED_K7_WAVES)))
composite_bwd_kernel(int C, const float4* __restrict__ splats, const int* __restrict__ flatten_ids,
                     const int* __restrict__ offsets, int width, int height, int tile_w, int tile_h,
                     const float* __restrict__ backgrounds, const float* __restrict__ render_alpha,
                     const float* __restrict__ t_final, const int* __restrict__ last_ids, const float* __restrict__ v_render,
                     const float* __restrict__ v_alpha, float* __restrict__ vsplat, int n_big_flags,
                     const int* __restrict__ tile_order, const int* __restrict__ n_split_dev, BwdPost post) {
    __shared__ __attribute__((aligned(16))) float s_part[kParkSlots * kParkSlot];
    __shared__ __attribute__((aligned(16))) float s_rec[kBatch][kRecFloats];
    const int n_total = C * tile_w * tile_h;
    const int b = blockIdx.x;
    const int keep_all = (n_big_flags >> 30) << 2;
    const int n_big = n_big_flags & 0x3fffffff;
    if (tile_order != nullptr) {
        // Costliest-first order (tile_order_body, from the forward pass's per-tile visit counts): workgroups are dispatched
        // in index order as wave slots free up, i.e. greedy longest-processing-time-first scheduling.  The first n_split
        // tiles of the order -- too heavy for one wave not to set the length of the launch -- are dealt as four quadrant
        // waves each, ahead of everything else.
        const int n_split = n_split_dev[0];
        if (b < 4 * n_split) {
            bwd_tile<CH, 1>(tile_order[b >> 2], (b & 3) | keep_all, s_rec, s_part, C, splats, flatten_ids, offsets,
                            width, height, tile_w, tile_h, backgrounds, render_alpha, t_final, last_ids, v_render, v_alpha, vsplat, post);
        } else {
            // (Dealing the CHEAPEST tiles -- handed out last -- as quadrant waves too, to fill the ragged end of a launch of
            // ~2 whole tiles per wave slot, was measured: 261 us without, 268 / 287 / 306 with 10 / 20 / 30 % of the tiles;
            // a quadrant wave stages and culls the tile's whole list again)
            const int i = b - 3 * n_split;
            if (i >= n_total) return;                    // (the grid is sized for the largest n_split the host allows)
            bwd_tile<CH, 4>(tile_order[i], keep_all, s_rec, s_part, C, splats, flatten_ids, offsets, width, height,
                            tile_w, tile_h, backgrounds, render_alpha, t_final, last_ids, v_render, v_alpha, vsplat, post);
        }
        return;
    }
    if (b < n_big) {
        bwd_tile<CH, 4>(xcd_remap(b, n_total), keep_all, s_rec, s_part, C, splats, flatten_ids, offsets, width, height, tile_w, tile_h,
                        backgrounds, render_alpha, t_final, last_ids, v_render, v_alpha, vsplat, post);
    } else {
        int t, q;
        small_wave(b - n_big, n_total - n_big, t, q);
        bwd_tile<CH, 1>(xcd_remap(n_big + t, n_total), q | keep_all, s_rec, s_part, C, splats, flatten_ids, offsets, width, height,
                        tile_w, tile_h, backgrounds, render_alpha, t_final, last_ids, v_render, v_alpha, vsplat, post);
    }
}

// depth = alpha > 0 ? depth : max depth (model.py:304-306; the max is detached), in place on the depth image the forward
// kernel wrote, with the max folded from its per-wave partials by every workgroup (a few thousand L2-resident floats)
__global__ void __launch_bounds__(256)
depth_fixup_kernel(int n_pix, const float* __restrict__ alpha, float* __restrict__ depth,
                   const float* __restrict__ part, int n_part) {
    __shared__ float s[4];
    float dm = -3.0e38f;
    // n_part is a multiple of 4 (four slots per tile) and the array 16-byte aligned: float4, eight vectors in flight per
    // trip.  Every workgroup folds all partials itself, so the grid is kept to one workgroup per CU (1 024 workgroups
    // re-read 130 KB each at 1080p: 12 us for this kernel instead of 4).
    const float4* p4 = reinterpret_cast<const float4*>(part);
    const int n4 = n_part >> 2;
    for (int b0 = threadIdx.x; b0 < n4; b0 += 16 * 256) {             // sixteen vectors in flight per trip
        float4 e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = p4[b0 + 256 * j < n4 ? b0 + 256 * j : 0];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (b0 + 256 * j < n4) dm = fmaxf(fmaxf(dm, fmaxf(e[j].x, e[j].y)), fmaxf(e[j].z, e[j].w));
    }
    dm = wave_max(dm);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = dm;
    __syncthreads();
    const float dmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    // four pixels per 16-byte load, eight loads in flight per trip (n_pix need not be a multiple of 4: scalar tail)
    const size_t n_vec = (size_t)n_pix >> 2, stride = (size_t)gridDim.x * 256;
    const float4* a4 = reinterpret_cast<const float4*>(alpha);
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n_vec; i0 += 8 * stride) {
        float4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = a4[i0 + j * stride < n_vec ? i0 + j * stride : i0];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t i = i0 + j * stride;
            if (i >= n_vec) break;
            if (!(a[j].x > 0.f)) depth[4 * i] = dmax;
            if (!(a[j].y > 0.f)) depth[4 * i + 1] = dmax;
            if (!(a[j].z > 0.f)) depth[4 * i + 2] = dmax;
            if (!(a[j].w > 0.f)) depth[4 * i + 3] = dmax;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_pix & 3)) {
        const size_t i = (n_vec << 2) + threadIdx.x;
        if (!(alpha[i] > 0.f)) depth[i] = dmax;
    }
}

// ---- costliest-first tile order for the backward launch: tile_order_body (qed_common.h) as a launch of its own --------
__global__ void __launch_bounds__(1024)
tile_order_kernel(TileOrderJob job) {
    __shared__ __attribute__((aligned(8))) int lds[kOrderLdsInts];
    tile_order_body<1024>(job, lds);
}

}  // namespace qed

using namespace qed;

#ifdef QED_COMPOSITE_STATS
// diagnostic build only: copy the 32 counters to the host (synchronises) and optionally clear them
extern "C" int qed_debug_composite_stats(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// Number of leading tiles (in dispatch order) composited by whole-tile waves; the rest get one wave per
// quadrant.  The quadrant waves are there to fill the end of the launch: their number is a multiple of the
// device's wave slots for the kernel (measured optimum at 1080p on MI355X: 1.9 x 5120 slots for the forward,
// 1.2 x 4096 for the backward kernel), independent of the image size; an image with fewer tiles than that is
// composited by quadrant waves only (it could not fill the device with whole-tile waves anyway).
static long long big_tiles(long long n_tiles, double small_waves_per_slot, int waves_per_simd, int launch_flags) {
    // QED_CL_* (include/qed_splat.h): tests force one launch shape so that small images exercise all of them
    switch (launch_flags & 3) {
        case QED_CL_TILE_WAVES: return n_tiles;
        case QED_CL_QUADRANT_WAVES: return 0;
        case QED_CL_HALF_AND_HALF: return n_tiles / 2;
        default: break;
    }
    // (asked of the runtime at every call: no state is kept in the library)
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    const long long slots = (long long)n_cu * 4 * waves_per_simd;
    const long long n_small = (long long)(small_waves_per_slot * (double)slots / 4.0);
    return n_tiles > n_small ? n_tiles - n_small : 0;
}

// QED_CL_NO_CULL turns the quadrant culling off (bit 30 of the kernels' n_big argument)
static int no_cull_flag(int launch_flags) { return (launch_flags & QED_CL_NO_CULL) ? (1 << 30) : 0; }

extern "C" int qed_composite_fwd(int32_t C, int32_t N, const float* splats, const int32_t* flatten_ids,
                                 const int32_t* offsets, int32_t width, int32_t height, int32_t tile_w,
                                 int32_t tile_h, int32_t channels, const float* backgrounds, float* render,
                                 float* alpha, float* t_final, int32_t* last_ids, int32_t* tile_cost,
                                 const int32_t* tile_order, const qed_post_t* post, int32_t launch_flags, void* stream) {
    QED_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "bad extents");
    QED_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (RGB) or 4 (RGB+D)");
    QED_REQUIRE(tile_w == (width + QED_TILE - 1) / QED_TILE && tile_h == (height + QED_TILE - 1) / QED_TILE,
                "tile grid does not match the image (tile size is 16)");
    QED_REQUIRE(offsets && render && alpha && last_ids, "null buffers");
    // flatten_ids may be NULL when the sorted list is empty (offsets are then all zero)
    QED_REQUIRE(N == 0 || splats, "null splat buffer");
    const long long grid = (long long)C * tile_w * tile_h;
    QED_REQUIRE(grid < (1ll << 29), "too many tiles");
    FwdPost fp{nullptr, nullptr, nullptr, nullptr};
    if (post != nullptr) {
        QED_REQUIRE(post->background && post->rgb, "post: background and rgb required");
        QED_REQUIRE(channels == 3 || (post->depth && post->tile_dmax), "post: depth image and tile_dmax required with a depth channel");
        fp = FwdPost{post->background, post->rgb, post->depth, post->tile_dmax};
    }
    hipStream_t st = (hipStream_t)stream;
    const long long n_big = big_tiles(grid, QED_K6_SMALL, QED_K6_WAVES, launch_flags);
    unsigned blocks = (unsigned)(n_big + 4 * (grid - n_big));
    const int* fwd_order = nullptr;
    // a forced launch shape (test hook) keeps the plain tile order
    if (tile_order != nullptr && (launch_flags & 3) == 0) {
        fwd_order = tile_order;
        blocks = (unsigned)(grid + 3ll * max_split_tiles(grid));
    }
    if (channels == 4)
        hipLaunchKernelGGL(composite_fwd_kernel<4>, dim3(blocks), dim3(64), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render, alpha, t_final, last_ids,
                           tile_cost, (int)n_big | no_cull_flag(launch_flags), fp, fwd_order);
    else
        hipLaunchKernelGGL(composite_fwd_kernel<3>, dim3(blocks), dim3(64), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render, alpha, t_final, last_ids,
                           tile_cost, (int)n_big | no_cull_flag(launch_flags), fp, fwd_order);
    if (post != nullptr && channels == 4) {
        const long long n_pix = (long long)C * width * height;
        long long g = (n_pix / 4 + 255) / 256;
        if (g > 256) g = 256;
        if (g < 1) g = 1;
        hipLaunchKernelGGL(depth_fixup_kernel, dim3((unsigned)g), dim3(256), 0, st, (int)n_pix, (const float*)alpha, post->depth,
                           (const float*)post->tile_dmax, (int)(4 * grid));
    }
    return check_launch("qed_composite_fwd");
}

static void launch_tile_order(const int32_t* tile_cost, long long grid, int32_t* order_ws, hipStream_t st) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, tile_order_job(tile_cost, grid, order_ws));
}

extern "C" int qed_composite_bwd(int32_t C, int32_t N, const float* splats, const int32_t* flatten_ids,
                                 const int32_t* offsets, int32_t width, int32_t height, int32_t tile_w,
                                 int32_t tile_h, int32_t channels, const float* backgrounds,
                                 const float* render_alpha, const float* t_final, const int32_t* last_ids,
                                 const float* v_render, const float* v_alpha, float* vsplat, const int32_t* tile_cost,
                                 int32_t* order_ws, const qed_post_grad_t* post, int32_t launch_flags, void* stream) {
    QED_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "bad extents");
    QED_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (RGB) or 4 (RGB+D)");
    QED_REQUIRE(tile_w == (width + QED_TILE - 1) / QED_TILE && tile_h == (height + QED_TILE - 1) / QED_TILE,
                "tile grid does not match the image (tile size is 16)");
    QED_REQUIRE(offsets && render_alpha && last_ids, "null buffers");
    QED_REQUIRE(post != nullptr || (v_render && v_alpha), "v_render and v_alpha, or the post-processing gradients");
    BwdPost bp{nullptr, nullptr, nullptr, nullptr};
    if (post != nullptr) {
        QED_REQUIRE(post->background && post->render, "post: background and the forward pass's render required");
        QED_REQUIRE(v_render == nullptr && v_alpha == nullptr, "post gradients replace v_render / v_alpha");
        bp = BwdPost{post->background, post->render, post->v_rgb, post->v_depth};
    }
    QED_REQUIRE((tile_cost == nullptr) == (order_ws == nullptr), "tile_cost and order_ws go together");
    QED_REQUIRE(((uintptr_t)tile_cost & 15) == 0, "tile_cost must be 16-byte aligned");
    if (N == 0) return QED_OK;
    QED_REQUIRE(splats && vsplat, "null splat buffers");
    const long long grid = (long long)C * tile_w * tile_h;
    QED_REQUIRE(grid < (1ll << 29), "too many tiles");
    hipStream_t st = (hipStream_t)stream;
    const long long n_big = big_tiles(grid, QED_K7_SMALL, QED_K7_WAVES, launch_flags);
    unsigned blocks = (unsigned)(n_big + 4 * (grid - n_big));
    const int* tile_order = nullptr;
    // a forced launch shape (test hook) keeps the plain tile order
    if (tile_cost != nullptr && (launch_flags & 3) == 0 && (launch_flags & QED_CL_ORDER_READY)) {
        // order_ws already holds the order of THIS tile_cost: qed_ssim_fwd_step computed it as a passenger of its launch
        tile_order = order_ws;
        blocks = (unsigned)(grid + 3ll * max_split_tiles(grid));
    } else if (tile_cost != nullptr && (launch_flags & 3) == 0) {
        // (Launched by the caller on a SECOND stream, beside the loss passes that sit between the two compositing kernels, the
        // ordering is off the critical path on paper; replayed from a hipGraph the fork / join cost more than its 10 us:
        // 1.002 against 0.992 ms per step, profiles/r05_negative_results.txt)
        launch_tile_order(tile_cost, grid, order_ws, st);
        tile_order = order_ws;
        blocks = (unsigned)(grid + 3ll * max_split_tiles(grid));
    }
    if (channels == 4)
        hipLaunchKernelGGL(composite_bwd_kernel<4>, dim3(blocks), dim3(64), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render_alpha, t_final, last_ids,
                           v_render, v_alpha, vsplat, (int)n_big | no_cull_flag(launch_flags), tile_order,
                           tile_order ? tile_order + grid : nullptr, bp);
    else
        hipLaunchKernelGGL(composite_bwd_kernel<3>, dim3(blocks), dim3(64), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render_alpha, t_final, last_ids,
                           v_render, v_alpha, vsplat, (int)n_big | no_cull_flag(launch_flags), tile_order,
                           tile_order ? tile_order + grid : nullptr, bp);
    return check_launch("qed_composite_bwd");
}
