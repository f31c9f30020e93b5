// the (1 - SSIM) term of the parent's RGB loss, fused forward + backward.
//
// The reference's get_loss_dict (model.py:83-85) calls SplatfactoModel.get_loss_dict, whose main loss
// is (1 - ssim_lambda) * L1 + ssim_lambda * (1 - SSIM(gt, pred)) with pytorch_msssim's SSIM
// (data_range 1, 11-tap Gaussian window sigma 1.5 applied separably WITHOUT padding, K = (0.01, 0.03),
// mean over the (H-10) x (W-10) map and the 3 channels).  Neither library is vendored in the
// reference, so this follows the published definition; oracle/splat_oracle.py::ssim is
// the restatement the tests compare against.
//
// Forward: one 256-thread workgroup per TW x TH block of the SSIM map.  The (TW+10) x (TH+10) patch
// of both images is staged in LDS; the window is applied along rows with 4 outputs per work item
// (16-byte LDS reads, every loaded value feeds up to 4 outputs), then along columns with CB outputs
// per thread.  Besides the block's partial sum of the SSIM map the kernel stores, per map pixel q and
// channel, the three coefficient maps the backward needs:
//     d ssim(q) / d x(p) = w(p-q) * [ A(q) + 2 x(p) B(q) + y(p) C(q) ]
//     A = dS/dmu_x - 2 mu_x dS/dvar_x - mu_y dS/dcov,  B = dS/dvar_x,  C = dS/dcov
// Backward: grad(p) = conv(A)(p) + 2 x(p) conv(B)(p) + y(p) conv(C)(p) ("full" correlation, zero
// outside the map), the same two passes over the three maps.
// 55 + 55 window FMAs per map pixel and channel forward, 33 + 33 backward; HBM traffic ~70 MB in + 75 MB of maps
// forward, 75 + 70 MB in + 25 MB out backward at 1080p.  Measured (SQ_ACTIVE_INST_VALU): 50 % / 35 % of the vector
// pipe -- both kernels are bound by their barrier-separated phases, not by arithmetic or bytes, which is why the
// backward pass can take the loss-gradient pass's work along (FUSE below) for 11 of the 24 us that pass costs alone.
#include "qed_common.h"

namespace qed {

constexpr int kWin = 11;
constexpr int kHalo = kWin - 1;            // 10

// pytorch_msssim _fspecial_gauss_1d(11, 1.5) evaluated in fp32 exactly as the library does.
// Symmetric: correlation and convolution coincide, which the backward pass relies on.
__device__ __constant__ float c_win[kWin] = {
    1.028380357e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f, 2.130055279e-01f, 2.660117149e-01f,
    2.130055279e-01f, 1.093606874e-01f, 3.600077331e-02f, 7.598758209e-03f, 1.028380357e-03f};

template <int TW_, int TH_, int NT_ = 256>
struct SsimTile {
    static constexpr int TW = TW_, TH = TH_, NT = NT_;          // NT threads per workgroup
    static constexpr int NW = NT / 64;
    static constexpr int PW = TW + kHalo, PH = TH + kHalo;
    static constexpr int SP = TW + 12;                 // patch row stride: 16-float reads stay in the row; odd multiple of 4
    static constexpr int SH = TW + 4;                  // row-pass output stride (odd multiple of 4 floats)
    static constexpr int ROW_ITEMS = PH * (TW / 4);    // row pass: 4 outputs per item
    // Row-pass work items -> threads.  Item (gx, py) reads 16 floats at py * SP + 4 gx with ds_read_b128, whose 64 lanes are
    // served in four groups of 16 over 16 four-bank slots: slot = (SP / 4) py + gx + j (mod 16), SP / 4 odd.  Lanes of
    // one wave with the SAME gx and consecutive py never collide; dealt as consecutive items i -> (i % PH, i / PH), every
    // wave straddles a column change (PH = 42 rows against 64 lanes) and the reads took 6.5 instead of 4 LDS cycles
    // (scripts/ubench/lds_patterns, profiles/r03_lds_patterns.txt).  So: one column group per wave and round, lane = row.
    static constexpr bool WAVE_COLUMNS = PH <= 64;
    static constexpr int ROW_SLOTS = WAVE_COLUMNS ? 64 * (TW / 4) : ROW_ITEMS;
    static __device__ __forceinline__ bool row_item(int i, int& gx, int& py) {
        if constexpr (WAVE_COLUMNS) { gx = i >> 6; py = i & 63; return py < PH; }
        else { gx = i / PH; py = i - gx * PH; return true; }
    }
    static constexpr int COL_GROUPS = NT / TW;
    static constexpr int CB = TH / COL_GROUPS;         // column pass: CB outputs per thread
    static constexpr int RPP = NT / PW;                // staging: patch rows per pass (thread -> one patch column)
    static constexpr int LOADS = (PH + RPP - 1) / RPP; // patch pixels per staging thread
    static_assert(TW % 4 == 0 && NT % TW == 0 && NT % 64 == 0 && TH % COL_GROUPS == 0, "tile shape");
};

// predicted colour channel k of pixel (iy, ix): either a plain [H,W,3] image or composited on the fly
// from render[H,W,CH] + (1 - alpha) * background, clamped to [0,1] (model.py:296-297)
template <bool COMPOSITE>
__device__ __forceinline__ float pred_at(const float* __restrict__ pred, const float* __restrict__ alpha,
                                         const float* __restrict__ bg, int channels, size_t pix, int k) {
    if constexpr (COMPOSITE) {
        const float v = pred[pix * channels + k] + (1.f - alpha[pix]) * bg[k];
        return fminf(fmaxf(v, 0.f), 1.f);
    } else {
        return pred[pix * 3 + k];
    }
}

struct __attribute__((packed, aligned(4))) Float3 { float a, b, c; };

// pass 1 of the image loss as passenger workgroups of the SSIM forward launch (see ssim_fwd_kernel)
struct LossReduceJob {
    const float* gt_depth;
    float* sums;
    int n_pix, channels, n_blocks;      // n_blocks == 0: no job
};

// 4 adjacent window sums of a row: out[o] = sum_d w[d] v[o + d], o = 0..3, from 14 consecutive inputs
__device__ __forceinline__ float4 window4(const float (&v)[16], const float (&w)[kWin]) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int d = 0; d < kWin; ++d) {
        r.x += w[d] * v[d]; r.y += w[d] * v[d + 1]; r.z += w[d] * v[d + 2]; r.w += w[d] * v[d + 3];
    }
    return r;
}

__device__ __forceinline__ void read16(const float* __restrict__ p, float (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * j);
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
}

// MASKED: both images are multiplied by mask[H,W] before the SSIM (the parent's loss does that to the rendered and
// the ground-truth image when the batch holds a mask: SplatfactoModel.get_loss_dict behind model.py:83-85)
template <bool COMPOSITE, bool MASKED, class T>
#ifndef QED_SSIM_FWD_WAVES
#define QED_SSIM_FWD_WAVES 4
#endif
#ifndef QED_SSIM_FWD_PLAIN_WAVES
#define QED_SSIM_FWD_PLAIN_WAVES 3   // the plain-image form spills 16-18 registers at four waves per SIMD: 60 -> 53 us at three
#endif
__global__ void __launch_bounds__(T::NT)
__attribute__((amdgpu_waves_per_eu(COMPOSITE ? QED_SSIM_FWD_WAVES : QED_SSIM_FWD_PLAIN_WAVES,
                                   COMPOSITE ? QED_SSIM_FWD_WAVES : QED_SSIM_FWD_PLAIN_WAVES)))
ssim_fwd_kernel(int H, int W, int channels, const float* __restrict__ pred, const float* __restrict__ alpha,
                const float* __restrict__ bg, const float* __restrict__ gt, const float* __restrict__ mask,
                float* __restrict__ maps, float* __restrict__ ssim_sum, TileOrderJob order_job, LossReduceJob reduce_job) {
    constexpr int TW = T::TW, PW = T::PW, PH = T::PH, SP = T::SP, SH = T::SH, CB = T::CB;
    __shared__ __attribute__((aligned(16))) float s_x[PH * SP], s_y[PH * SP];
    __shared__ __attribute__((aligned(16))) float s_h[4][PH * SH];
    __shared__ float s_red[T::NW];
    // PASSENGERS (the fused training step, qed_ssim_fwd_step): launches of their own on the critical chain otherwise.
    //  * workgroup 0 sorts the compositing backward's tiles by the cost the forward pass counted (a one-workgroup job,
    //    ~10 us in front of that kernel);
    //  * the LAST reduce_job.n_blocks workgroups are pass 1 of the image loss (valid-depth count and largest depth of the
    //    render this launch reads anyway; ~9 us): streaming work that runs while the SSIM workgroups of the second
    //    generation are in their compute phases.
    // The SSIM workgroups are the indices in between.
    const int has_job = order_job.cost4 != nullptr ? 1 : 0;
    if (has_job && blockIdx.x == 0) {
        static_assert(sizeof(s_h) >= kOrderLdsInts * sizeof(int), "the ordering job's LDS fits the row-pass planes");
        tile_order_body<T::NT>(order_job, reinterpret_cast<int*>(&s_h[0][0]));
        return;
    }
    const int block_id = (int)blockIdx.x - has_job, n_blocks = (int)gridDim.x - has_job - reduce_job.n_blocks;
    if (block_id >= n_blocks) {
        float (*s2)[T::NW] = reinterpret_cast<float (*)[T::NW]>(&s_h[0][0]);
        if (reduce_job.channels == 4)
            loss_reduce_body<4, T::NT>(block_id - n_blocks, reduce_job.n_blocks, reduce_job.n_pix, pred, reduce_job.gt_depth, mask,
                                       reduce_job.sums, s2);
        else
            loss_reduce_body<3, T::NT>(block_id - n_blocks, reduce_job.n_blocks, reduce_job.n_pix, pred, reduce_job.gt_depth, mask,
                                       reduce_job.sums, s2);
        return;
    }
    const int Ho = H - kHalo, Wo = W - kHalo;
    // neighbouring blocks share 10-pixel halos: keep them on one XCD's L2 (workgroups are dealt round-robin
    // over the 8 XCDs, so a linear grid is remapped to give each XCD a contiguous run of blocks)
    const int nbx = (Wo + TW - 1) / TW;
    const int blk = xcd_remap(block_id, n_blocks);
    const int by = blk / nbx, bx = blk - by * nbx;
    const int ox = bx * TW, oy = by * T::TH;                            // origin in the SSIM map
    const int tid = threadIdx.x;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const size_t n_out = (size_t)Ho * Wo;
    float w[kWin];
#pragma unroll
    for (int d = 0; d < kWin; ++d) w[d] = c_win[d];
    // Staging: thread -> patch column spx, rows spy + RPP * j.  All pixels of all three channels are
    // fetched up front in one batch of unconditional loads from clamped coordinates (a load inside a
    // bounds branch is waited for inside it): one exposed memory latency per workgroup, every fetched
    // cache line fully used.
    const int spy = tid / PW, spx = tid - spy * PW;
    const bool stager = tid < T::RPP * PW;
    // Two loops: the first contains NOTHING but the loads (no branch on `channels`, no bounds test, no arithmetic
    // on a loaded value), so all of them are issued back to back; the second composites / zeroes with selects.
    // Written as one loop, every iteration ended in a branch and the disassembly showed seven serialised round
    // trips (load, wait, composite, next load) per workgroup.
    float xs[T::LOADS][3], ys[T::LOADS][3];
    {
        Float3 graw[T::LOADS];
        float4 praw[T::LOADS];
        float araw[T::LOADS];
        float mraw[T::LOADS];
        size_t pix[T::LOADS];
#pragma unroll
        for (int j = 0; j < T::LOADS; ++j)
            pix[j] = (size_t)min(oy + spy + T::RPP * j, H - 1) * W + min(ox + spx, W - 1);
#pragma unroll
        for (int j = 0; j < T::LOADS; ++j) mraw[j] = MASKED ? mask[pix[j]] : 1.f;
        if (COMPOSITE && channels == 4) {
#pragma unroll
            for (int j = 0; j < T::LOADS; ++j) {
                graw[j] = *reinterpret_cast<const Float3*>(gt + pix[j] * 3);
                praw[j] = *reinterpret_cast<const float4*>(pred + pix[j] * 4);
                araw[j] = alpha[pix[j]];
            }
        } else {
#pragma unroll
            for (int j = 0; j < T::LOADS; ++j) {
                graw[j] = *reinterpret_cast<const Float3*>(gt + pix[j] * 3);
                const Float3 v = *reinterpret_cast<const Float3*>(pred + pix[j] * 3);
                praw[j] = make_float4(v.a, v.b, v.c, 0.f);
                araw[j] = COMPOSITE ? alpha[pix[j]] : 1.f;
            }
        }
        float bgc[3] = {0.f, 0.f, 0.f};
        if constexpr (COMPOSITE) { bgc[0] = bg[0]; bgc[1] = bg[1]; bgc[2] = bg[2]; }
#pragma unroll
        for (int j = 0; j < T::LOADS; ++j) {
            const bool in = (oy + spy + T::RPP * j < H) & (ox + spx < W);
            const float r[3] = {praw[j].x, praw[j].y, praw[j].z};
            const float g[3] = {graw[j].a, graw[j].b, graw[j].c};
            const float om = 1.f - araw[j];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float c = COMPOSITE ? fminf(fmaxf(r[k] + om * bgc[k], 0.f), 1.f) : r[k];
                xs[j][k] = in ? (MASKED ? c * mraw[j] : c) : 0.f;
                ys[j][k] = in ? (MASKED ? g[k] * mraw[j] : g[k]) : 0.f;
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        __syncthreads();
        if (stager) {
#pragma unroll
            for (int j = 0; j < T::LOADS; ++j)
                if (spy + T::RPP * j < PH) {
                    s_x[(spy + T::RPP * j) * SP + spx] = xs[j][k];
                    s_y[(spy + T::RPP * j) * SP + spx] = ys[j][k];
                }
        }
        __syncthreads();
        // rows: window sums of x, y, x^2 + y^2, x y; consecutive lanes take consecutive rows
        for (int i = tid; i < T::ROW_SLOTS; i += T::NT) {
            int gx, py;
            if (!T::row_item(i, gx, py)) continue;
            float x[16], y[16], p[16];
            read16(&s_x[py * SP + 4 * gx], x);
            read16(&s_y[py * SP + 4 * gx], y);
            const int o = py * SH + 4 * gx;
            *reinterpret_cast<float4*>(&s_h[0][o]) = window4(x, w);
            *reinterpret_cast<float4*>(&s_h[1][o]) = window4(y, w);
            // SSIM only ever needs var_x + var_y, so x^2 and y^2 share one window sum
#pragma unroll
            for (int j = 0; j < 14; ++j) p[j] = x[j] * x[j] + y[j] * y[j];
            *reinterpret_cast<float4*>(&s_h[2][o]) = window4(p, w);
#pragma unroll
            for (int j = 0; j < 14; ++j) p[j] = x[j] * y[j];
            *reinterpret_cast<float4*>(&s_h[3][o]) = window4(p, w);
        }
        __syncthreads();
        // columns: CB vertically adjacent outputs per thread
        const int tx = tid % TW, ty0 = (tid / TW) * CB;
        float st[4][CB];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[CB + kHalo];
#pragma unroll
            for (int j = 0; j < CB + kHalo; ++j) v[j] = s_h[q][(ty0 + j) * SH + tx];
#pragma unroll
            for (int o = 0; o < CB; ++o) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < kWin; ++d) a += w[d] * v[o + d];
                st[q][o] = a;
            }
        }
        const int qx = ox + tx;
        float* m = maps + (size_t)k * 3 * n_out;
#pragma unroll
        for (int o = 0; o < CB; ++o) {
            const int qy = oy + ty0 + o;
            if (qx < Wo && qy < Ho) {
                const float mu1 = st[0][o], mu2 = st[1][o];
                const float mm = mu1 * mu1 + mu2 * mu2;
                const float var12 = st[2][o] - mm, cov = st[3][o] - mu1 * mu2;     // var_x + var_y, cov
                const float a1 = 2.f * mu1 * mu2 + C1, b1 = mm + C1;
                const float a2 = 2.f * cov + C2, b2 = var12 + C2;
                const float ib1 = __builtin_amdgcn_rcpf(b1), ib2 = __builtin_amdgcn_rcpf(b2);   // 1 ulp
                const float lum = a1 * ib1, cs = a2 * ib2;
                acc += lum * cs;
                // partial derivatives of S = lum * cs w.r.t. mu1 (holding var1, cov), var1, cov
                const float dS_dmu1 = cs * (2.f * mu2 * ib1 - a1 * ib1 * ib1 * 2.f * mu1);
                const float dS_dvar1 = -lum * a2 * ib2 * ib2;
                const float dS_dcov = lum * 2.f * ib2;
                if (maps != nullptr) {                                           // (NULL: value only)
                    const size_t q = (size_t)qy * Wo + qx;
                    m[q] = dS_dmu1 - 2.f * mu1 * dS_dvar1 - mu2 * dS_dcov;       // A
                    m[n_out + q] = dS_dvar1;                                     // B
                    m[2 * n_out + q] = dS_dcov;                                  // C
                }
            }
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) s_red[tid >> 6] = acc;
    __syncthreads();
    // one partial per workgroup (folded by the consumer): no zeroing launch in front of the kernel and no 2 040
    // same-address atomics behind it
    if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < T::NW; ++q) tot += s_red[q];
        ssim_sum[block_id] = tot;
    }
}

// v_pred[H,W,3] = scale * (scale_dev ? scale_dev[0] : 1) * d(sum of the SSIM map)/d pred, pred = the colour BEFORE the
// mask multiply (MASKED: x = m pred, y = m gt entered the SSIM, so the chain rule adds one factor m)
// FUSE: the pass also does the work of the loss-gradient pass that would follow it, instead of storing the SSIM gradient
// for that pass to read back together with the images (116 MB of re-reads and one launch at 1080p):
//   FUSE = 3 / 4 (COMPOSITE; the fused training step, qed_loss_grad_ssim): loss.hip's loss_grad_kernel for a render of
//     that many channels.  Inside each channel pass the thread's own colour is at hand before and after the clamp, so
//     the L1 term, the clamp mask and the background-composite gradient are formed there; after the last pass only
//     the depth channel is left: -> v_render / v_alpha, and the workgroup's loss sums are added to the slots
//     qed_loss_reduce zeroed.
//   FUSE = 1 (plain images; get_loss_dict's backward, qed_image_losses_ssim_bwd): image_loss_grad_kernel -- the L1 term
//     scaled by its upstream gradient joins the SSIM term in v_pred, the depth-L1 gradient goes to v_depth.
struct LossFuse {
    const float* depth;     // FUSE 1: the depth image [H,W]
    const float* gt_depth;
    float* sums;            // the loss passes' workspace (FUSE 3/4: pass 1 = qed_loss_reduce has run; FUSE 1: sums[2] = n_valid)
    int n_loss_blocks;      // FUSE 3/4: pass 1's grid
    float w_rgb;            // rgb_weight / (3 n_pix)
    float depth_lambda;
    const float* g_main;    // FUSE 1: upstream gradients of the two loss terms (device scalars; NULL = that term has none)
    const float* g_depth;
    float* v_render;        // FUSE 3/4 outputs
    float* v_alpha;
    float* v_depth;         // FUSE 1 output (may be NULL)
    float4* zero_buf;       // FUSE != 0: a buffer this launch also zeroes (the compositing backward's per-Gaussian
    long long zero_vec;     //   accumulator: saves the fill launch in front of it), as 16-byte vectors; may be NULL
};

template <bool COMPOSITE, bool MASKED, class T, int FUSE = 0>
#ifndef QED_SSIM_BWD_WAVES
#define QED_SSIM_BWD_WAVES 4
#endif
#ifndef QED_SSIM_FUSED_WAVES
#define QED_SSIM_FUSED_WAVES 3      // FUSE 3/4 holds the depth inputs and the alpha gradients of its four pixels as well
#endif
__global__ void __launch_bounds__(T::NT)
__attribute__((amdgpu_waves_per_eu(FUSE >= 3 ? QED_SSIM_FUSED_WAVES : QED_SSIM_BWD_WAVES,
                                   FUSE >= 3 ? QED_SSIM_FUSED_WAVES : QED_SSIM_BWD_WAVES)))
ssim_bwd_kernel(int H, int W, int channels, const float* __restrict__ pred, const float* __restrict__ alpha,
                const float* __restrict__ bg, const float* __restrict__ gt, const float* __restrict__ mask,
                const float* __restrict__ maps, float scale, const float* __restrict__ scale_dev,
                float* __restrict__ v_pred, LossFuse lf) {
    static_assert(FUSE == 0 || (FUSE == 1 && !COMPOSITE) || ((FUSE == 3 || FUSE == 4) && COMPOSITE), "see LossFuse");
    constexpr int TW = T::TW, PW = T::PW, PH = T::PH, SP = T::SP, SH = T::SH, CB = T::CB;
    __shared__ __attribute__((aligned(16))) float s_m[3][PH * SP];
    __shared__ __attribute__((aligned(16))) float s_h[3][PH * SH];
    __shared__ float s_red[2][T::NW];
    const int Ho = H - kHalo, Wo = W - kHalo;
    const int nbx = (W + TW - 1) / TW;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);                   // halo sharing: see ssim_fwd_kernel
    const int by = blk / nbx, bx = blk - by * nbx;
    const int ox = bx * TW, oy = by * T::TH;                            // origin in the image
    const int tid = threadIdx.x;
    const size_t n_out = (size_t)Ho * Wo;
    float w[kWin];
#pragma unroll
    for (int d = 0; d < kWin; ++d) w[d] = c_win[d];
    const int spy = tid / PW, spx = tid - spy * PW;
    const bool stager = tid < T::RPP * PW;
    const int tx = tid % TW, ty0 = (tid / TW) * CB;
    const int ix = ox + tx;
    if (scale_dev != nullptr) scale *= scale_dev[0];
    // FUSE: what the depth part needs is REQUESTED here, ahead of the three channel passes, and used after them (asked for
    // at the end, its latency is exposed once per workgroup generation: the first fused form took 100 us against 50 + 29
    // for the two passes it replaces).  FUSE 4: pass 1's per-workgroup partials (n_valid, the largest rendered depth),
    // folded as loss_grad_kernel does.
    float l1 = 0.f, dsum = 0.f, w_rgb = 0.f, w_d = 0.f, dmax = 0.f;
    float ep_d[CB], ep_a[CB], ep_gd[CB];
    bool inside[CB];
#pragma unroll
    for (int o = 0; o < CB; ++o) { ep_d[o] = 0.f; ep_a[o] = 1.f; ep_gd[o] = 0.f; inside[o] = ix < W && oy + ty0 + o < H; }
    if constexpr (FUSE != 0) {
        w_rgb = lf.w_rgb;
        if constexpr (FUSE == 1) w_rgb *= lf.g_main != nullptr ? lf.g_main[0] : 0.f;
        const bool want_depth = FUSE == 4 || (FUSE == 1 && lf.v_depth != nullptr);
        float ep_nv[4] = {0.f, 0.f, 0.f, 0.f}, ep_dm[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        if constexpr (FUSE == 4) {
            static_assert(kLossMaxGrid <= 4 * T::NT, "four partials per thread cover pass 1's grid");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = tid + T::NT * j;
                const int bc = b < lf.n_loss_blocks ? b : 0;
                const float nv = loss_part(lf.sums, 0)[bc], dm = loss_part(lf.sums, 1)[bc];
                ep_nv[j] = b < lf.n_loss_blocks ? nv : 0.f;
                ep_dm[j] = b < lf.n_loss_blocks ? dm : -3.0e38f;
            }
        }
        if (want_depth) {
#pragma unroll
            for (int o = 0; o < CB; ++o) {
                const size_t pix = (size_t)min(oy + ty0 + o, H - 1) * W + min(ix, W - 1);
                if constexpr (FUSE == 4) { ep_d[o] = pred[4 * pix + 3]; ep_a[o] = alpha[pix]; }
                else ep_d[o] = lf.depth[pix];
                ep_gd[o] = lf.gt_depth[pix];
            }
        }
        if constexpr (FUSE == 4) {
            float nv = (ep_nv[0] + ep_nv[1]) + (ep_nv[2] + ep_nv[3]);
            float dm = fmaxf(fmaxf(ep_dm[0], ep_dm[1]), fmaxf(ep_dm[2], ep_dm[3]));
            nv = wave_sum(nv);
            dm = wave_max(dm);
            if ((tid & 63) == 0) { s_red[0][tid >> 6] = nv; s_red[1][tid >> 6] = dm; }
            __syncthreads();
            float nvalid = 0.f;
            dmax = -3.0e38f;
#pragma unroll
            for (int q = 0; q < T::NW; ++q) { nvalid += s_red[0][q]; dmax = fmaxf(dmax, s_red[1][q]); }
            w_d = nvalid > 0.f ? lf.depth_lambda / nvalid : 0.f;
            // (s_red is written again only after the channel passes, each of which starts with a barrier)
        } else if constexpr (FUSE == 1) {
            const float nvalid = lf.sums[2];
            w_d = nvalid > 0.f ? (lf.g_depth != nullptr ? lf.g_depth[0] : 0.f) * lf.depth_lambda / nvalid : 0.f;
        }
    }
    if constexpr (FUSE != 0) {
        // this workgroup's slice of the buffer to zero: stores only, issued beside the loads of the first pass
        if (lf.zero_buf != nullptr) {
            const long long per = (lf.zero_vec + gridDim.x - 1) / gridDim.x;
            const long long lo = (long long)blockIdx.x * per, hi = lo + per < lf.zero_vec ? lo + per : lf.zero_vec;
            for (long long i = lo + tid; i < hi; i += T::NT) lf.zero_buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float r0[CB], r1[CB];            // channels 0 and 1 wait for channel 2: one 12-byte store per pixel
    float mo[CB];
#pragma unroll
    for (int o = 0; o < CB; ++o) mo[o] = MASKED ? mask[(size_t)min(oy + ty0 + o, H - 1) * W + min(ix, W - 1)] : 1.f;
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        const float* m = maps + (size_t)k * 3 * n_out;
        // map patch rows oy-10 .. oy+TH-1, cols ox-10 .. ox+TW-1 (zero outside the map):
        // grad(p) = sum_d w[d] M(p - d) = sum_d' w[d'] M(p - 10 + d')  (symmetric window)
        float mv[T::LOADS][3];
#pragma unroll
        for (int j = 0; j < T::LOADS; ++j) {
            const int qy = oy + spy + T::RPP * j - kHalo, qx = ox + spx - kHalo;
            const size_t q = (size_t)min(max(qy, 0), Ho - 1) * Wo + min(max(qx, 0), Wo - 1);
            const float al = m[q], bl = m[n_out + q], cl = m[2 * n_out + q];
            const bool in = qy >= 0 && qy < Ho && qx >= 0 && qx < Wo;
            mv[j][0] = in ? al : 0.f; mv[j][1] = in ? bl : 0.f; mv[j][2] = in ? cl : 0.f;
        }
        // the output pixels' own colours, requested with the maps so their latency hides behind both passes
        float xo[CB], yo[CB], pre[CB];
        const float bgk = COMPOSITE ? bg[k] : 0.f;
#pragma unroll
        for (int o = 0; o < CB; ++o) {
            const size_t pix = (size_t)min(oy + ty0 + o, H - 1) * W + min(ix, W - 1);
            if constexpr (COMPOSITE) {
                pre[o] = pred[pix * channels + k] + (1.f - alpha[pix]) * bgk;     // model.py:296 before the clamp
                xo[o] = fminf(fmaxf(pre[o], 0.f), 1.f);
            } else {
                pre[o] = 0.f;
                xo[o] = pred[pix * 3 + k];
            }
            yo[o] = gt[pix * 3 + k];
            if constexpr (MASKED) { xo[o] *= mo[o]; yo[o] *= mo[o]; }
        }
        __syncthreads();
        if (stager) {
#pragma unroll
            for (int j = 0; j < T::LOADS; ++j)
                if (spy + T::RPP * j < PH) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) s_m[q][(spy + T::RPP * j) * SP + spx] = mv[j][q];
                }
        }
        __syncthreads();
        for (int i = tid; i < T::ROW_SLOTS; i += T::NT) {
            int gx, py;
            if (!T::row_item(i, gx, py)) continue;
            const int o = py * SH + 4 * gx;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float v[16];
                read16(&s_m[q][py * SP + 4 * gx], v);
                *reinterpret_cast<float4*>(&s_h[q][o]) = window4(v, w);
            }
        }
        __syncthreads();
        float g[3][CB];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v[CB + kHalo];
#pragma unroll
            for (int j = 0; j < CB + kHalo; ++j) v[j] = s_h[q][(ty0 + j) * SH + tx];
#pragma unroll
            for (int o = 0; o < CB; ++o) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < kWin; ++d) a += w[d] * v[o + d];
                g[q][o] = a;
            }
        }
#pragma unroll
        for (int o = 0; o < CB; ++o) {
            float r = scale * (g[0][o] + 2.f * xo[o] * g[1][o] + yo[o] * g[2][o]);
            if constexpr (MASKED) r *= mo[o];
            if constexpr (FUSE != 0) {
                // the L1 term of the same (masked) colours; FUSE 3/4: then through the clamp and the background composite
                const float diff = xo[o] - yo[o];
                if (inside[o]) l1 += fabsf(diff);
                const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                r = w_rgb * sg * mo[o] + r;
                if constexpr (COMPOSITE)
                    r = (pre[o] >= 0.f && pre[o] <= 1.f) ? r : 0.f;               // torch.clamp backward (inclusive)
            }
            if (k == 0) r0[o] = r;
            else if (k == 1) r1[o] = r;
            else if (inside[o]) {
                const size_t pix = (size_t)(oy + ty0 + o) * W + ix;
                if constexpr (FUSE == 0 || FUSE == 1) {
                    Float3 v; v.a = r0[o]; v.b = r1[o]; v.c = r;
                    *reinterpret_cast<Float3*>(v_pred + pix * 3) = v;
                    if constexpr (FUSE == 1) {
                        if (lf.v_depth != nullptr) {
                            const float dp = ep_d[o] * mo[o], dg = ep_gd[o] * mo[o];
                            float gd = 0.f;
                            if (isfinite(dp) && isfinite(dg) && dg > 0.f) {
                                const float dd = dp - dg;
                                gd = w_d * (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f)) * mo[o];
                            }
                            lf.v_depth[pix] = gd;
                        }
                    }
                } else if constexpr (FUSE == 4) {
                    const float a = ep_a[o];
                    const float dg = ep_gd[o] * mo[o];
                    const float dp = (a > 0.f ? ep_d[o] : dmax) * mo[o];               // model.py:306
                    float v3 = 0.f;
                    if (isfinite(dp) && isfinite(dg) && dg > 0.f) {
                        const float dd = dp - dg;
                        dsum += fabsf(dd);
                        if (a > 0.f) v3 = w_d * (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f)) * mo[o];
                    }
                    *reinterpret_cast<float4*>(lf.v_render + 4 * pix) = make_float4(r0[o], r1[o], r, v3);
                    lf.v_alpha[pix] = -((r0[o] * bg[0] + r1[o] * bg[1]) + r * bgk);   // d colour / d alpha = -background
                } else {
                    Float3 v; v.a = r0[o]; v.b = r1[o]; v.c = r;
                    *reinterpret_cast<Float3*>(lf.v_render + 3 * pix) = v;
                    lf.v_alpha[pix] = -((r0[o] * bg[0] + r1[o] * bg[1]) + r * bgk);
                }
            }
        }
    }
    if constexpr (FUSE >= 3) {
        l1 = wave_sum(l1);
        dsum = wave_sum(dsum);
        __syncthreads();
        if ((tid & 63) == 0) { s_red[0][tid >> 6] = l1; s_red[1][tid >> 6] = dsum; }
        __syncthreads();
        if (tid == 0) {
            const int slot = (int)(blockIdx.x % (unsigned)lf.n_loss_blocks);
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int q = 0; q < T::NW; ++q) { t0 += s_red[0][q]; t1 += s_red[1][q]; }
            atomicAdd(&loss_part(lf.sums, 2)[slot], t0);
            atomicAdd(&loss_part(lf.sums, 3)[slot], t1);
        }
    }
}

#ifndef QED_SSIM_FWD_TW
#define QED_SSIM_FWD_TW 32
#define QED_SSIM_FWD_TH 32
#endif
#ifndef QED_SSIM_FWD_NT
#define QED_SSIM_FWD_NT 256
#endif
#ifndef QED_SSIM_BWD_NT
#define QED_SSIM_BWD_NT 256
#endif
#ifndef QED_SSIM_BWD_TW
#define QED_SSIM_BWD_TW 32
#define QED_SSIM_BWD_TH 32
#endif
using Tile = SsimTile<QED_SSIM_FWD_TW, QED_SSIM_FWD_TH, QED_SSIM_FWD_NT>;
using TileB = SsimTile<QED_SSIM_BWD_TW, QED_SSIM_BWD_TH, QED_SSIM_BWD_NT>;

}  // namespace qed

using namespace qed;


extern "C" int64_t qed_ssim_sum_floats(int32_t height, int32_t width) {
    if (height <= kHalo || width <= kHalo) return QED_E_INVALID_ARG;
    return (int64_t)((width - kHalo + Tile::TW - 1) / Tile::TW) * ((height - kHalo + Tile::TH - 1) / Tile::TH);
}

extern "C" int64_t qed_ssim_maps_floats(int32_t height, int32_t width) {
    if (height <= kHalo || width <= kHalo) return QED_E_INVALID_ARG;
    return 9ll * (height - kHalo) * (width - kHalo);
}

static int ssim_fwd_launch(int32_t height, int32_t width, int32_t channels, const float* pred, const float* alpha,
                           const float* background, const float* gt_rgb, const float* mask, float* maps,
                           float* ssim_sum, const TileOrderJob& job, const LossReduceJob& rjob, void* stream,
                           const char* who) {
    QED_REQUIRE(height > kHalo && width > kHalo, "image smaller than the 11 x 11 SSIM window");
    QED_REQUIRE(pred && gt_rgb && ssim_sum, "null buffers");
    QED_REQUIRE(alpha == nullptr || (background && (channels == 3 || channels == 4)), "composite mode needs a background");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(((width - kHalo + Tile::TW - 1) / Tile::TW) * ((height - kHalo + Tile::TH - 1) / Tile::TH) +
                    (job.cost4 != nullptr ? 1 : 0) + rjob.n_blocks);
#define QED_SSIM_FWD(COMP, MASK, CHN)                                                                               \
    hipLaunchKernelGGL((ssim_fwd_kernel<COMP, MASK, Tile>), grid, dim3(Tile::NT), 0, st, height, width, CHN, pred, alpha,   \
                       background, gt_rgb, mask, maps, ssim_sum, job, rjob)
    if (alpha != nullptr) { if (mask) QED_SSIM_FWD(true, true, channels); else QED_SSIM_FWD(true, false, channels); }
    else { if (mask) QED_SSIM_FWD(false, true, 3); else QED_SSIM_FWD(false, false, 3); }
#undef QED_SSIM_FWD
    return check_launch(who);
}

extern "C" int qed_ssim_fwd(int32_t height, int32_t width, int32_t channels, const float* pred, const float* alpha,
                            const float* background, const float* gt_rgb, const float* mask, float* maps,
                            float* ssim_sum, void* stream) {
    return ssim_fwd_launch(height, width, channels, pred, alpha, background, gt_rgb, mask, maps, ssim_sum,
                           TileOrderJob{nullptr, 0, nullptr, 0.f, 0, 0}, LossReduceJob{nullptr, nullptr, 0, 0, 0}, stream,
                           "qed_ssim_fwd");
}

// qed_ssim_fwd of the fused training step, with its passengers (see ssim_fwd_kernel): the costliest-first tile order of the
// compositing backward that follows (tile_cost != NULL) and pass 1 of the image loss = qed_loss_reduce (sums != NULL)
extern "C" int qed_ssim_fwd_step(int32_t height, int32_t width, int32_t channels, const float* render, const float* alpha,
                                 const float* background, const float* gt_rgb, const float* mask, float* maps,
                                 float* ssim_sum, const int32_t* tile_cost, int64_t n_tiles, int32_t* order_ws,
                                 const float* gt_depth, float* sums, void* stream) {
    QED_REQUIRE(alpha != nullptr, "the training step's render (composite mode)");
    QED_REQUIRE((tile_cost == nullptr) == (order_ws == nullptr), "tile_cost and order_ws go together");
    QED_REQUIRE(tile_cost == nullptr || (n_tiles >= 1 && n_tiles < (1ll << 29)), "the tile count");
    QED_REQUIRE(((uintptr_t)tile_cost & 15) == 0, "tile_cost must be 16-byte aligned");
    QED_REQUIRE(sums == nullptr || channels == 3 || gt_depth, "gt_depth required with a depth channel");
    const int n_pix = height * width;
    const TileOrderJob job = tile_cost != nullptr ? tile_order_job(tile_cost, n_tiles, order_ws)
                                                  : TileOrderJob{nullptr, 0, nullptr, 0.f, 0, 0};
    const LossReduceJob rjob = sums != nullptr ? LossReduceJob{gt_depth, sums, n_pix, channels, (int)loss_reduce_grid(n_pix)}
                                               : LossReduceJob{nullptr, nullptr, 0, 0, 0};
    return ssim_fwd_launch(height, width, channels, render, alpha, background, gt_rgb, mask, maps, ssim_sum, job, rjob, stream,
                           "qed_ssim_fwd_step");
}

extern "C" int qed_ssim_bwd(int32_t height, int32_t width, int32_t channels, const float* pred, const float* alpha,
                            const float* background, const float* gt_rgb, const float* mask, const float* maps,
                            float scale, const float* scale_dev, float* v_pred, void* stream) {
    QED_REQUIRE(height > kHalo && width > kHalo, "image smaller than the 11 x 11 SSIM window");
    QED_REQUIRE(pred && gt_rgb && maps && v_pred, "null buffers");
    QED_REQUIRE(alpha == nullptr || (background && (channels == 3 || channels == 4)), "composite mode needs a background");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(((width + TileB::TW - 1) / TileB::TW) * ((height + TileB::TH - 1) / TileB::TH));
#define QED_SSIM_BWD(COMP, MASK, CHN)                                                                               \
    hipLaunchKernelGGL((ssim_bwd_kernel<COMP, MASK, TileB>), grid, dim3(TileB::NT), 0, st, height, width, CHN, pred, alpha,   \
                       background, gt_rgb, mask, maps, scale, scale_dev, v_pred, LossFuse{})
    if (alpha != nullptr) { if (mask) QED_SSIM_BWD(true, true, channels); else QED_SSIM_BWD(true, false, channels); }
    else { if (mask) QED_SSIM_BWD(false, true, 3); else QED_SSIM_BWD(false, false, 3); }
#undef QED_SSIM_BWD
    return check_launch("qed_ssim_bwd");
}

// SSIM backward + the loss-gradient pass in ONE launch (fused training step): after qed_ssim_fwd and qed_loss_reduce.
extern "C" int qed_loss_grad_ssim(int32_t height, int32_t width, int32_t channels, const float* render, const float* alpha,
                                  const float* background, const float* gt_rgb, const float* gt_depth, const float* mask,
                                  const float* maps, float* sums, float rgb_weight, float depth_lambda, float ssim_scale,
                                  float* v_render, float* v_alpha, float* losses, const float* ssim_sum,
                                  int32_t ssim_sum_n, float ssim_offset, float* zero_buf, int64_t zero_floats,
                                  const qed_adam_tick_t* tick, void* stream) {
    QED_REQUIRE(height > kHalo && width > kHalo, "image smaller than the 11 x 11 SSIM window");
    QED_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (RGB) or 4 (RGB+D)");
    QED_REQUIRE(render && alpha && background && gt_rgb && maps && sums && v_render && v_alpha && losses && ssim_sum,
                "null buffers");
    QED_REQUIRE(channels == 3 || gt_depth, "gt_depth required with a depth channel");
    QED_REQUIRE(zero_floats >= 0 && (zero_floats == 0 || zero_buf) && zero_floats % 4 == 0 && ((uintptr_t)zero_buf & 15) == 0,
                "zero_buf: 16-byte aligned, a multiple of 4 floats");
    hipStream_t st = (hipStream_t)stream;
    const int n_pix = height * width;
    const unsigned n_loss = loss_reduce_grid(n_pix);
    const LossFuse lf{nullptr, gt_depth, sums, (int)n_loss, rgb_weight / (3.f * (float)n_pix), depth_lambda, nullptr, nullptr,
                      v_render, v_alpha, nullptr, zero_floats > 0 ? (float4*)zero_buf : nullptr, zero_floats / 4};
    const dim3 grid(((width + TileB::TW - 1) / TileB::TW) * ((height + TileB::TH - 1) / TileB::TH));
#define QED_SSIM_BWD_FUSED(MASK, CHN)                                                                                \
    hipLaunchKernelGGL((ssim_bwd_kernel<true, MASK, TileB, CHN>), grid, dim3(TileB::NT), 0, st, height, width, CHN, render, \
                       alpha, background, gt_rgb, mask, maps, ssim_scale, (const float*)nullptr, (float*)nullptr, lf)
    if (channels == 4) { if (mask) QED_SSIM_BWD_FUSED(true, 4); else QED_SSIM_BWD_FUSED(false, 4); }
    else { if (mask) QED_SSIM_BWD_FUSED(true, 3); else QED_SSIM_BWD_FUSED(false, 3); }
#undef QED_SSIM_BWD_FUSED
    AdamTick tk{};
    if (tick != nullptr) {
        QED_REQUIRE(tick->dev_state && tick->beta1 > 0.f && tick->beta2 > 0.f, "tick: device state and betas required");
        QED_REQUIRE(tick->dev_lr_slot == nullptr || (tick->lr_init > 0.f && tick->lr_final > 0.f && tick->max_steps > 0),
                    "tick: a scheduled rate needs positive rates and max_steps");
        tk = AdamTick{tick->dev_state, tick->beta1, tick->beta2, tick->dev_lr_slot,
                      tick->dev_lr_slot ? logf(tick->lr_init) : 0.f, tick->dev_lr_slot ? logf(tick->lr_final) : 0.f,
                      tick->dev_lr_slot ? 1.f / (float)tick->max_steps : 0.f, tick->skip_flag};
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, n_pix, (int)n_loss, channels == 4 ? 1 : 0, sums,
                       rgb_weight, depth_lambda, losses, ssim_sum, (int)ssim_sum_n, ssim_scale, ssim_offset, tk);
    return check_launch("qed_loss_grad_ssim");
}

// qed_ssim_bwd + qed_image_losses_bwd in ONE launch (get_loss_dict's backward when the main loss has an upstream
// gradient and an SSIM term): v_rgb = g_main * d main_loss / d rgb (L1 + SSIM), v_depth = g_depth * d depth_loss / d depth.
extern "C" int qed_image_losses_ssim_bwd(int32_t height, int32_t width, const float* rgb, const float* depth,
                                         const float* gt_rgb, const float* gt_depth, const float* mask, const float* maps,
                                         const float* sums, float rgb_weight, float depth_lambda, float ssim_scale,
                                         const float* g_main, const float* g_depth, float* v_rgb, float* v_depth,
                                         float* zero_buf, int64_t zero_floats, void* stream) {
    QED_REQUIRE(height > kHalo && width > kHalo, "image smaller than the 11 x 11 SSIM window");
    QED_REQUIRE(rgb && gt_rgb && maps && sums && g_main && v_rgb, "null buffers");
    QED_REQUIRE(v_depth == nullptr || (depth && gt_depth), "depth images required for a depth gradient");
    QED_REQUIRE(zero_buf == nullptr || (zero_floats >= 0 && zero_floats % 4 == 0 && ((uintptr_t)zero_buf & 15) == 0),
                "zero_buf must be 16-byte aligned and a multiple of 4 floats long");
    hipStream_t st = (hipStream_t)stream;
    const int n_pix = height * width;
    const LossFuse lf{depth, gt_depth, const_cast<float*>(sums), 0, rgb_weight / (3.f * (float)n_pix), depth_lambda, g_main,
                      g_depth, nullptr, nullptr, v_depth, reinterpret_cast<float4*>(zero_buf),
                      zero_buf ? (long long)(zero_floats / 4) : 0};
    const dim3 grid(((width + TileB::TW - 1) / TileB::TW) * ((height + TileB::TH - 1) / TileB::TH));
    if (mask)
        hipLaunchKernelGGL((ssim_bwd_kernel<false, true, TileB, 1>), grid, dim3(TileB::NT), 0, st, height, width, 3, rgb,
                           (const float*)nullptr, (const float*)nullptr, gt_rgb, mask, maps, ssim_scale, g_main, v_rgb, lf);
    else
        hipLaunchKernelGGL((ssim_bwd_kernel<false, false, TileB, 1>), grid, dim3(TileB::NT), 0, st, height, width, 3, rgb,
                           (const float*)nullptr, (const float*)nullptr, gt_rgb, mask, maps, ssim_scale, g_main, v_rgb, lf);
    return check_launch("qed_image_losses_ssim_bwd");
}
