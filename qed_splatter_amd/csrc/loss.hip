// K8: fused image-space loss + gradient, and the fused flat Adam step.
//
// K8 collapses what the reference does in eager torch after rasterization():
//   rgb   = clamp(render[..., :3] + (1 - alpha) * background, 0, 1)        model.py:296-297
//   depth = where(alpha > 0, render[..., 3:4], render[..., 3:4].max())      model.py:304-306
//   depth_loss = depth_lambda * mean |depth - gt| over finite & gt > 0     model.py:87-116
//   L1 part of the parent's RGB loss (SplatfactoModel.get_loss_dict, upstream of model.py:83-85)
// into two streaming passes (a global max / count must be known before gradients can be written).
// Pure HBM streaming: 20 B/pixel render+alpha, 16-20 B/pixel ground truth in, 20 B/pixel out.
#include <stdarg.h>
#include <stdlib.h>

#include "qed_common.h"

namespace qed {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// order-preserving float <-> int map so that atomicMax works for any sign
__device__ __forceinline__ int float_to_ordered(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

struct PixelEval {
    float pre[3];      // rgb before the clamp
    float rgb[3];
    float a;
    float d_render;    // render depth channel (0 if RGB only)
};

template <int CH>
__device__ __forceinline__ PixelEval eval_pixel(const float* __restrict__ render, const float* __restrict__ alpha,
                                                const float* __restrict__ bg, size_t i) {
    PixelEval p;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CH == 4) {
        const float4 t = *reinterpret_cast<const float4*>(render + 4 * i);
        c[0] = t.x; c[1] = t.y; c[2] = t.z; c[3] = t.w;
    } else {
        c[0] = render[3 * i]; c[1] = render[3 * i + 1]; c[2] = render[3 * i + 2];
    }
    p.a = alpha[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.pre[k] = c[k] + (1.f - p.a) * bg[k];
        p.rgb[k] = fminf(fmaxf(p.pre[k], 0.f), 1.f);
    }
    p.d_render = c[3];
    return p;
}

// sums (QED_LOSS_SUMS_FLOATS floats): from [8] on four rows of kLossMaxGrid per-workgroup partials: n_valid, max depth
// (pass 1), sum |rgb - gt|, sum |depth - gt| (pass 2); loss_finalize_kernel folds them into sums[0..3]
// and the scalar losses.  Per-workgroup slots instead of same-address
// atomics: those serialise at ~12 ns each (measured: pass time grew linearly with the grid, 23-37 ns
// per workgroup), which capped the grid at 2 workgroups per CU and the passes at ~3 TB/s.
//
// Pass 1 only needs what must be known BEFORE a gradient can be written: the number of valid depth
// pixels (its reciprocal scales every depth gradient) and the largest rendered depth (the value
// alpha == 0 pixels take, model.py:306).  It reads the depth channel, the ground-truth depth and the
// mask -- not the colours.
// (kLossMaxGrid / loss_part: qed_common.h -- the fused SSIM-backward + loss-gradient pass of ssim.hip uses them too)

template <int CH>
__global__ void __launch_bounds__(256)
loss_reduce_kernel(int n_pix, const float* __restrict__ render, const float* __restrict__ alpha,
                   const float* __restrict__ bg, const float* __restrict__ gt_rgb, const float* __restrict__ gt_depth,
                   const float* __restrict__ mask, float* __restrict__ sums) {
    __shared__ float s[2][4];
    loss_reduce_body<CH, 256>((int)blockIdx.x, (int)gridDim.x, n_pix, render, gt_depth, mask, sums, s);
}

// Pass 2: gradients and the per-workgroup partials of the two loss sums.
template <int CH>
__global__ void __launch_bounds__(256)
loss_grad_kernel(int n_pix, const float* __restrict__ render, const float* __restrict__ alpha,
                 const float* __restrict__ bg, const float* __restrict__ gt_rgb, const float* __restrict__ gt_depth,
                 const float* __restrict__ mask, float* __restrict__ sums, float rgb_weight, float depth_lambda,
                 float* __restrict__ v_render, float* __restrict__ v_alpha, const float* __restrict__ v_rgb_extra) {
    const float w_rgb = rgb_weight / (3.f * (float)n_pix);
    __shared__ float s[2][4];
    // every workgroup folds pass 1's per-workgroup partials (same grid) into n_valid and the max depth
    float nvalid = 0.f, dmax = 0.f;
    if constexpr (CH == 4) {
        float nv = 0.f, dm = -3.0e38f;
        for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) {
            nv += loss_part(sums, 0)[b];
            dm = fmaxf(dm, loss_part(sums, 1)[b]);
        }
        nv = wave_sum(nv);
        dm = wave_max(dm);
        if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = nv; s[1][threadIdx.x >> 6] = dm; }
        __syncthreads();
        nvalid = s[0][0] + s[0][1] + s[0][2] + s[0][3];
        dmax = fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3]));
        __syncthreads();
    }
    const float w_d = nvalid > 0.f ? depth_lambda / nvalid : 0.f;
    float dsum = 0.f, l1 = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_pix; i += (size_t)gridDim.x * 256) {
        const PixelEval p = eval_pixel<CH>(render, alpha, bg, i);
        float vr[4] = {0.f, 0.f, 0.f, 0.f};
        float va = 0.f;
        // the parent's loss multiplies BOTH images by the mask before L1 and SSIM (SplatfactoModel.get_loss_dict,
        // behind model.py:83-85); the depth term does the same at model.py:93-97
        const float m = mask ? mask[i] : 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float diff = p.rgb[k] * m - gt_rgb[3 * i + k] * m;
            l1 += fabsf(diff);
            const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            const bool pass = p.pre[k] >= 0.f && p.pre[k] <= 1.f;     // torch.clamp backward (inclusive)
            // v_rgb_extra: gradient of an additional term w.r.t. the same (unmasked) clamped colour (SSIM, ssim.hip)
            const float g_in = w_rgb * sg * m + (v_rgb_extra ? v_rgb_extra[3 * i + k] : 0.f);
            const float g = pass ? g_in : 0.f;
            vr[k] = g;
            va -= g * bg[k];
        }
        if constexpr (CH == 4) {
            const float dg = gt_depth[i] * m;
            const float dsel = p.a > 0.f ? p.d_render : dmax;          // model.py:306
            const float dp = dsel * m;
            if (isfinite(dp) && isfinite(dg) && dg > 0.f) {
                const float diff = dp - dg;
                dsum += fabsf(diff);
                const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                if (p.a > 0.f) vr[3] = w_d * sg * m;
            }
            *reinterpret_cast<float4*>(v_render + 4 * i) = make_float4(vr[0], vr[1], vr[2], vr[3]);
        } else {
            v_render[3 * i] = vr[0]; v_render[3 * i + 1] = vr[1]; v_render[3 * i + 2] = vr[2];
        }
        v_alpha[i] = va;
    }
    dsum = wave_sum(dsum);
    l1 = wave_sum(l1);
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = dsum; s[1][threadIdx.x >> 6] = l1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_part(sums, 2)[blockIdx.x] = s[1][0] + s[1][1] + s[1][2] + s[1][3];
        loss_part(sums, 3)[blockIdx.x] = s[0][0] + s[0][1] + s[0][2] + s[0][3];
    }
}

// One workgroup folds the per-workgroup partials of both passes into sums[0..3] and the three scalar
// losses.  A separate (tiny) launch rather than a "last workgroup" ticket inside pass 2: the ticket
// needs a device-scope fence behind each workgroup's stores plus a returning same-address atomic, which
// measured ~20 ns per workgroup, serialised.
__global__ void __launch_bounds__(256)
loss_finalize_kernel(int n_pix, int n_blocks, int has_depth, float* __restrict__ sums, float rgb_weight,
                     float depth_lambda, float* __restrict__ losses, const float* __restrict__ extra_sum, int extra_n,
                     float extra_scale, float extra_offset, AdamTick tick) {
    if (threadIdx.x == 64 && tick.state != nullptr) adam_tick(tick);     // (a passenger: see AdamTick)
    float nv = 0.f, dm = -3.0e38f, tl = 0.f, td = 0.f, ex = 0.f;
    // every partial this thread folds is REQUESTED before the first one is used (clamped index, switched off by a
    // select): as a loop of load-then-add rounds this one-workgroup launch took 9 us, twice its launch floor
    static_assert(kLossMaxGrid <= 4 * 256, "four partials per row and thread");
    float r0[4], r1[4], r2[4], r3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = threadIdx.x + 256 * j;
        const int bc = b < n_blocks ? b : 0;
        r0[j] = has_depth ? loss_part(sums, 0)[bc] : 0.f;               // has_depth < 0: a valid count but no max row
        r1[j] = has_depth > 0 ? loss_part(sums, 1)[bc] : -3.0e38f;
        r2[j] = loss_part(sums, 2)[bc];
        r3[j] = loss_part(sums, 3)[bc];
    }
    if (extra_sum != nullptr) {                             // per-workgroup partials of the SSIM map sum (qed_ssim_fwd)
        for (int b0 = threadIdx.x; b0 < extra_n; b0 += 8 * 256) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = extra_sum[b0 + 256 * j < extra_n ? b0 + 256 * j : 0];
#pragma unroll
            for (int j = 0; j < 8; ++j) ex += b0 + 256 * j < extra_n ? e[j] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = (int)threadIdx.x + 256 * j < n_blocks;
        nv += ok ? r0[j] : 0.f;
        dm = fmaxf(dm, ok ? r1[j] : -3.0e38f);
        tl += ok ? r2[j] : 0.f;
        td += ok ? r3[j] : 0.f;
    }
    nv = wave_sum(nv); dm = wave_max(dm); tl = wave_sum(tl); td = wave_sum(td); ex = wave_sum(ex);
    __shared__ float s[5][4];
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s[0][w] = nv; s[1][w] = dm; s[2][w] = tl; s[3][w] = td; s[4][w] = ex;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nvalid = s[0][0] + s[0][1] + s[0][2] + s[0][3];
        const float dmax = fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3]));
        const float tot_l1 = s[2][0] + s[2][1] + s[2][2] + s[2][3], tot_d = s[3][0] + s[3][1] + s[3][2] + s[3][3];
        sums[0] = tot_l1; sums[1] = tot_d; sums[2] = nvalid; sums[3] = has_depth > 0 ? dmax : 0.f;
        losses[0] = rgb_weight * tot_l1 / (3.f * (float)n_pix);
        if (extra_sum != nullptr) losses[0] += extra_offset + extra_scale * (s[4][0] + s[4][1] + s[4][2] + s[4][3]);
        losses[1] = nvalid > 0.f ? depth_lambda * tot_d / nvalid : 0.f;       // empty -> 0.0 (model.py:111-114)
        losses[2] = losses[0] + losses[1];
    }
}


// =====================================================================================================
// The same arithmetic split the way the reference's call sequence splits it: get_outputs() returns rgb /
// depth images (model.py:295-297, 304-306), get_loss_dict() turns them into two scalar losses (the parent's
// main loss behind model.py:83-85 and the depth term of :87-116) that the trainer sums and differentiates.
// Each half is one autograd node on the host side (qed_splatter_amd/model.py: _PostProcess, _ImageLosses).
// =====================================================================================================

// per-workgroup maxima of the rendered depth channel -> part[blockIdx.x]
__global__ void __launch_bounds__(256)
post_max_kernel(int n_pix, const float* __restrict__ render, float* __restrict__ part) {
    float dmax = -3.0e38f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_pix; i += (size_t)gridDim.x * 256)
        dmax = fmaxf(dmax, render[4 * i + 3]);
    dmax = wave_max(dmax);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = dmax;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}

// rgb = clamp(render[..., :3] + (1 - alpha) bg, 0, 1); depth = alpha > 0 ? render[..., 3] : max(render[..., 3])
template <int CH>
__global__ void __launch_bounds__(256)
post_fwd_kernel(int n_pix, const float* __restrict__ render, const float* __restrict__ alpha,
                const float* __restrict__ bg, const float* __restrict__ part, int n_part, float* __restrict__ rgb,
                float* __restrict__ depth) {
    float dmax = 0.f;
    if constexpr (CH == 4) {
        __shared__ float s[4];
        float dm = -3.0e38f;
        for (int b = threadIdx.x; b < n_part; b += 256) dm = fmaxf(dm, part[b]);
        dm = wave_max(dm);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = dm;
        __syncthreads();
        dmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_pix; i += (size_t)gridDim.x * 256) {
        const PixelEval p = eval_pixel<CH>(render, alpha, bg, i);
        rgb[3 * i] = p.rgb[0]; rgb[3 * i + 1] = p.rgb[1]; rgb[3 * i + 2] = p.rgb[2];
        if constexpr (CH == 4) depth[i] = p.a > 0.f ? p.d_render : dmax;
    }
}

// backward of the above: v_rgb[H,W,3], v_depth[H,W] (either may be NULL = no gradient) -> v_render, v_alpha
template <int CH>
__global__ void __launch_bounds__(256)
post_bwd_kernel(int n_pix, const float* __restrict__ render, const float* __restrict__ alpha,
                const float* __restrict__ bg, const float* __restrict__ v_rgb, const float* __restrict__ v_depth,
                float* __restrict__ v_render, float* __restrict__ v_alpha) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_pix; i += (size_t)gridDim.x * 256) {
        const PixelEval p = eval_pixel<CH>(render, alpha, bg, i);
        float vr[4] = {0.f, 0.f, 0.f, 0.f}, va = 0.f;
        if (v_rgb != nullptr) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool pass = p.pre[k] >= 0.f && p.pre[k] <= 1.f;     // torch.clamp backward (inclusive)
                const float g = pass ? v_rgb[3 * i + k] : 0.f;
                vr[k] = g;
                va -= g * bg[k];
            }
        }
        if constexpr (CH == 4) {
            // the max of model.py:306 is detached: pixels with alpha == 0 pass no depth gradient
            if (v_depth != nullptr && p.a > 0.f) vr[3] = v_depth[i];
            *reinterpret_cast<float4*>(v_render + 4 * i) = make_float4(vr[0], vr[1], vr[2], vr[3]);
        } else {
            v_render[3 * i] = vr[0]; v_render[3 * i + 1] = vr[1]; v_render[3 * i + 2] = vr[2];
        }
        v_alpha[i] = va;
    }
}

// get_loss_dict on images: per-workgroup partials of sum |m rgb - m gt| (row 2), sum |m d - m dgt| over valid pixels
// (row 3) and their number (row 0)
__global__ void __launch_bounds__(256)
image_loss_reduce_kernel(int n_pix, const float* __restrict__ rgb, const float* __restrict__ depth,
                         const float* __restrict__ gt_rgb, const float* __restrict__ gt_depth,
                         const float* __restrict__ mask, float* __restrict__ sums) {
    float l1 = 0.f, dsum = 0.f, nv = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_pix; i += (size_t)gridDim.x * 256) {
        const float m = mask ? mask[i] : 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) l1 += fabsf(rgb[3 * i + k] * m - gt_rgb[3 * i + k] * m);
        if (depth != nullptr) {
            const float dp = depth[i] * m, dg = gt_depth[i] * m;
            if (isfinite(dp) && isfinite(dg) && dg > 0.f) { dsum += fabsf(dp - dg); nv += 1.f; }
        }
    }
    l1 = wave_sum(l1); dsum = wave_sum(dsum); nv = wave_sum(nv);
    __shared__ float s[3][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[0][w] = l1; s[1][w] = dsum; s[2][w] = nv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_part(sums, 2)[blockIdx.x] = s[0][0] + s[0][1] + s[0][2] + s[0][3];
        loss_part(sums, 3)[blockIdx.x] = s[1][0] + s[1][1] + s[1][2] + s[1][3];
        loss_part(sums, 0)[blockIdx.x] = s[2][0] + s[2][1] + s[2][2] + s[2][3];
    }
}

// gradients of the two losses w.r.t. the images, each scaled by its upstream gradient (device scalars: the trainer
// sums the loss dict and calls backward, and may weight or scale the terms).  v_rgb holds the SSIM part on entry
// when `accumulate` (qed_ssim_bwd, already scaled) and receives the L1 part on top.
__global__ void __launch_bounds__(256)
image_loss_grad_kernel(int n_pix, const float* __restrict__ rgb, const float* __restrict__ depth,
                       const float* __restrict__ gt_rgb, const float* __restrict__ gt_depth,
                       const float* __restrict__ mask, const float* __restrict__ sums, float rgb_weight,
                       float depth_lambda, const float* __restrict__ g_main, const float* __restrict__ g_depth,
                       int accumulate, float* __restrict__ v_rgb, float* __restrict__ v_depth) {
    const float gm = g_main ? g_main[0] : 0.f, gd = g_depth ? g_depth[0] : 0.f;
    const float w_rgb = gm * rgb_weight / (3.f * (float)n_pix);
    const float nvalid = sums[2];
    const float w_d = nvalid > 0.f ? gd * depth_lambda / nvalid : 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_pix; i += (size_t)gridDim.x * 256) {
        const float m = mask ? mask[i] : 1.f;
        if (v_rgb != nullptr) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float diff = rgb[3 * i + k] * m - gt_rgb[3 * i + k] * m;
                const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                const float g = w_rgb * sg * m;
                v_rgb[3 * i + k] = accumulate ? v_rgb[3 * i + k] + g : g;
            }
        }
        if (v_depth != nullptr) {
            const float dp = depth[i] * m, dg = gt_depth[i] * m;
            float g = 0.f;
            if (isfinite(dp) && isfinite(dg) && dg > 0.f) {
                const float diff = dp - dg;
                g = w_d * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * m;
            }
            v_depth[i] = g;
        }
    }
}

// ---- fused flat Adam ---------------------------------------------------------------------------------
struct AdamGroups {
    long long begin[9];
    float lr[8];
    int n;
};

// omb1 / omb2 = 1 - beta formed in DOUBLE on the host and rounded once, as torch.optim.Adam's Python scalars are
// (1.f - (float)0.999 is off by 1.3e-5 relative, which exp_avg_sq then carries)
struct AdamCoef {
    float beta1, beta2, omb1, omb2, eps, inv_bc1, inv_bc2_sqrt;
};
static AdamCoef adam_coef(double beta1, double beta2, float eps, float inv_bc1, float inv_bc2_sqrt) {
    return AdamCoef{(float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, inv_bc1, inv_bc2_sqrt};
}

// `skip` (may be NULL): a device word that is non-zero when this step's frame must not be trained on -- the binning
// status word of qed_bin_tiles: an intersection overflow rendered the frame empty.  Read by every Adam launch (and the
// step-state tick): the update is then a no-op, with no host round trip.
__device__ __forceinline__ bool adam_skipped(const int* __restrict__ skip) { return skip != nullptr && skip[0] != 0; }

__device__ __forceinline__ float adam_update(const AdamCoef& a, float pp, float gg, float& mm, float& vv, float lr) {
    mm = a.beta1 * mm + a.omb1 * gg;
    vv = a.beta2 * vv + a.omb2 * gg * gg;
    const float denom = sqrtf(vv) * a.inv_bc2_sqrt + a.eps;
    return pp - lr * a.inv_bc1 * mm / denom;
}

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            AdamGroups grp, AdamCoef co, const float* __restrict__ dev_state, const float* __restrict__ dev_lr,
            const int* __restrict__ skip) {
    if (adam_skipped(skip)) return;
    // device-resident step state / learning rates (hipGraph replays cannot change kernel arguments)
    if (dev_state != nullptr) { co.inv_bc1 = dev_state[1]; co.inv_bc2_sqrt = dev_state[2]; }
    float lrs[8];                                        // (not written back into grp: a modified by-value
#pragma unroll                                           //  kernel argument is copied to scratch)
    for (int k = 0; k < 8; ++k) lrs[k] = dev_lr != nullptr ? dev_lr[k] : grp.lr[k];
    const long long total = grp.begin[grp.n];
    const long long nvec = total >> 2;
    // a gradient that is a view at an odd offset of a larger allocation (one group of separately held parameters, N not
    // a multiple of 4) is read with dword loads: a wave-uniform choice
    const bool g_aligned = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
    auto lr_of = [&](long long i) {
        float lr = lrs[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < grp.n && i >= grp.begin[k]) lr = lrs[k];
        return lr;
    };
    auto upd = [&](float pp, float gg, float& mm, float& vv, float lr) { return adam_update(co, pp, gg, mm, vv, lr); };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        // gradient and moments: read / written once per step -> non-temporal, so that they do not displace the
        // parameters (re-read by the next projection pass) from the last-level cache
        typedef float v4f __attribute__((ext_vector_type(4)));
        float4 pp = reinterpret_cast<float4*>(p)[i];
        v4f g4;
        if (g_aligned) g4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g) + i);
        else g4 = (v4f){g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3]};
        const v4f m4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(m) + i);
        const v4f v4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v) + i);
        const float4 gg = make_float4(g4.x, g4.y, g4.z, g4.w);
        float4 mm = make_float4(m4.x, m4.y, m4.z, m4.w), vv = make_float4(v4.x, v4.y, v4.z, v4.w);
        const long long e = i << 2;
        pp.x = upd(pp.x, gg.x, mm.x, vv.x, lr_of(e));
        pp.y = upd(pp.y, gg.y, mm.y, vv.y, lr_of(e + 1));
        pp.z = upd(pp.z, gg.z, mm.z, vv.z, lr_of(e + 2));
        pp.w = upd(pp.w, gg.w, mm.w, vv.w, lr_of(e + 3));
        reinterpret_cast<float4*>(p)[i] = pp;
        __builtin_nontemporal_store((v4f){mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<v4f*>(m) + i);
        __builtin_nontemporal_store((v4f){vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<v4f*>(v) + i);
    }
    // tail (total not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const long long e = (nvec << 2) + threadIdx.x;
        float mm = m[e], vv = v[e];
        p[e] = upd(p[e], g[e], mm, vv, lr_of(e));
        m[e] = mm; v[e] = vv;
    }
}

// ---- Adam over the two SH groups with the coefficient gradients rebuilt on the fly ---------------------
// The gradient of SH coefficient k, channel c of Gaussian n is a rank-1 product per view:
//   g[n][k][c] = scale * sum_views b_k(dir_view(n)) * v_view[n][c]        (v = clamp-masked colour gradient)
// so the 48 N coefficient gradients never need to exist in memory: project_bwd (QED_F_SH_GRAD_COMPACT)
// leaves 3 N floats, and this kernel evaluates the products while it streams p / m / v of the features_dc
// [N,3] and features_rest [N,KR,3] segments.  One workgroup pass = 256 Gaussians: one thread per Gaussian
// writes its row of KR*3 gradients to LDS, then all 256 threads stream the (contiguous) rows as float4.
// Element ranges need no alignment: a pass vectorises the 16-byte aligned interior of its rows and updates
// the <= 3 + 3 elements at the ends one by one; neighbouring passes own disjoint elements.
constexpr int kShChunk = 256;

__device__ __forceinline__ void adam_one(const AdamCoef& a, float* __restrict__ p, float* __restrict__ m,
                                         float* __restrict__ v, long long e, float g, float lr) {
    float mm = m[e], vv = v[e];
    p[e] = adam_update(a, p[e], g, mm, vv, lr);
    m[e] = mm; v[e] = vv;
}

__device__ __forceinline__ void adam_vec(const AdamCoef& a, float* __restrict__ p, float* __restrict__ m,
                                         float* __restrict__ v, long long e, const float4& gg, float lr) {
    // The moments are touched once per step and by nobody else: non-temporal loads / stores keep them from
    // displacing the parameters, which the next projection pass reads again, in the last-level cache.
    typedef float v4f __attribute__((ext_vector_type(4)));
    float4 pp = *reinterpret_cast<float4*>(p + e);
#ifndef QED_ADAM_TEMPORAL
    const v4f m4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(m + e));
    const v4f v4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v + e));
    float4 mm = make_float4(m4.x, m4.y, m4.z, m4.w), vv = make_float4(v4.x, v4.y, v4.z, v4.w);
#else
    float4 mm = *reinterpret_cast<float4*>(m + e);
    float4 vv = *reinterpret_cast<float4*>(v + e);
#endif
    pp.x = adam_update(a, pp.x, gg.x, mm.x, vv.x, lr);
    pp.y = adam_update(a, pp.y, gg.y, mm.y, vv.y, lr);
    pp.z = adam_update(a, pp.z, gg.z, mm.z, vv.z, lr);
    pp.w = adam_update(a, pp.w, gg.w, mm.w, vv.w, lr);
    *reinterpret_cast<float4*>(p + e) = pp;
#ifndef QED_ADAM_TEMPORAL
    __builtin_nontemporal_store((v4f){mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<v4f*>(m + e));
    __builtin_nontemporal_store((v4f){vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<v4f*>(v + e));
#else
    *reinterpret_cast<float4*>(m + e) = mm;
    *reinterpret_cast<float4*>(v + e) = vv;
#endif
}

// Adam over elements [lo, hi) by one workgroup: float4 over the 16-byte aligned interior, the <= 3 + 3 elements
// at the ends one by one.  g1(e) / g4(e) give the gradient of element e / of the aligned vector at e.
template <class G1, class G4>
__device__ __forceinline__ void adam_span(const AdamCoef& a, float* __restrict__ p, float* __restrict__ m,
                                          float* __restrict__ v, long long lo, long long hi, float lr, int tid, G1 g1,
                                          G4 g4) {
    const long long e_lo = (lo + 3) & ~3LL, e_hi = hi & ~3LL;
    for (long long e = e_lo + 4 * tid; e < e_hi; e += 4 * 256) adam_vec(a, p, m, v, e, g4(e), lr);
    const long long head_end = e_lo < hi ? e_lo : hi;
    if (lo + tid < head_end) adam_one(a, p, m, v, lo + tid, g1(lo + tid), lr);
    const long long tail = e_hi > e_lo ? e_hi : e_lo;
    if (tail + tid < hi) adam_one(a, p, m, v, tail + tid, g1(tail + tid), lr);
}

// Every group is [N, width[k]] rows; the last two are features_dc (width 3) and features_rest (width 3 KR).
struct AdamRows {
    long long begin[9];
    float lr[8];
    int width[8];
    int n;
};

template <int DEG>
__global__ void __launch_bounds__(256)
adam_sh_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, long long b_dc, long long b_rest,
               int RW, int N, const float* __restrict__ means, int n_views, const float* __restrict__ viewmats,
               long long viewmat_stride, const float* __restrict__ v_views, long long view_stride, float scale,
               AdamCoef a, const float* __restrict__ dev_state, const float* __restrict__ dev_lr, int dc_group,
               float lr_dc, float lr_rest, const int* __restrict__ skip) {
    extern __shared__ float s_g[];                       // kShChunk * RW + 4 floats
    constexpr int K = (DEG + 1) * (DEG + 1);             // coefficient rows with a non-zero gradient
    if (adam_skipped(skip)) return;
    if (dev_state != nullptr) { a.inv_bc1 = dev_state[1]; a.inv_bc2_sqrt = dev_state[2]; }
    if (dev_lr != nullptr) { lr_dc = dev_lr[dc_group]; lr_rest = dev_lr[dc_group + 1]; }
    const int tid = threadIdx.x;
    const int n_chunks = (N + kShChunk - 1) / kShChunk;
    auto g_dc = [&](long long e) {       // features_dc: b_0 is a constant, elementwise in the views' colour gradients
        const long long i = e - b_dc;
        float acc = 0.f;
        for (int c = 0; c < n_views; ++c) acc += SH_C0 * v_views[view_stride * c + i];
        return acc * scale;
    };
    // ---- features_dc: one grid-wide streaming pass ----
    {
        const long long gt = (long long)blockIdx.x * 256 + tid, gs = (long long)gridDim.x * 256;
        const float lr = lr_dc;
        const long long e_lo = (b_dc + 3) & ~3LL, e_hi = b_rest & ~3LL;
        for (long long e = e_lo + 4 * gt; e < e_hi; e += 4 * gs)
            adam_vec(a, p, m, v, e, make_float4(g_dc(e), g_dc(e + 1), g_dc(e + 2), g_dc(e + 3)), lr);
        if (blockIdx.x == 0) {
            const long long head_end = e_lo < b_rest ? e_lo : b_rest;
            if (b_dc + tid < head_end) adam_one(a, p, m, v, b_dc + tid, g_dc(b_dc + tid), lr);
            const long long tail = e_hi > e_lo ? e_hi : e_lo;
            if (tail + tid < b_rest) adam_one(a, p, m, v, tail + tid, g_dc(tail + tid), lr);
        }
    }
    // ---- features_rest: passes of 256 Gaussians, gradients through LDS ----
    for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int n0 = ch * kShChunk;
        const int cnt = N - n0 < kShChunk ? N - n0 : kShChunk;
        const long long lo = b_rest + (long long)n0 * RW, hi = lo + (long long)cnt * RW;
        const int sh = (int)(lo & 3);                    // LDS index of element e is e - (lo - sh): 16-byte
        if (tid < cnt && RW > 0) {                       // aligned exactly where the global address is
            float* row = s_g + sh + tid * RW;
            int filled = 0;
            if constexpr (K > 1) {
                const int n = n0 + tid;
                const float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
                float acc[3 * (K - 1)];
#pragma unroll
                for (int i = 0; i < 3 * (K - 1); ++i) acc[i] = 0.f;
                for (int c = 0; c < n_views; ++c) {
                    const float* vm = viewmats + viewmat_stride * c;
                    const float* vv = v_views + view_stride * c + (size_t)n * 3;
                    const float cv[3] = {vv[0], vv[1], vv[2]};
                    if (cv[0] == 0.f && cv[1] == 0.f && cv[2] == 0.f) continue;      // not visible in this view
                    float dir[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j)                                      // campos = -R^T t
                        dir[j] = mean[j] + (vm[0 + j] * vm[3] + vm[4 + j] * vm[7] + vm[8 + j] * vm[11]);
                    const float inorm = rsqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                    float b[K];
                    sh_basis<DEG>(dir[0] * inorm, dir[1] * inorm, dir[2] * inorm, b);
#pragma unroll
                    for (int k = 1; k < K; ++k) {
                        acc[3 * (k - 1)] += b[k] * cv[0]; acc[3 * (k - 1) + 1] += b[k] * cv[1];
                        acc[3 * (k - 1) + 2] += b[k] * cv[2];
                    }
                }
                filled = 3 * (K - 1) < RW ? 3 * (K - 1) : RW;
#pragma unroll
                for (int i = 0; i < 3 * (K - 1); ++i)
                    if (i < RW) row[i] = acc[i] * scale;
            }
            for (int i = filled; i < RW; ++i) row[i] = 0.f;                          // degrees not active yet
        }
        __syncthreads();
        // features_rest rows of this pass, gradients from LDS
        {
            const long long base = lo - sh;
            adam_span(a, p, m, v, lo, hi, lr_rest, tid, [&](long long e) { return s_g[e - base]; },
                      [&](long long e) { return *reinterpret_cast<const float4*>(s_g + (e - base)); });
        }
        __syncthreads();
    }
}

// advances the step by one (AdamTick, qed_common.h)
__global__ void adam_tick_kernel(AdamTick tick) {
    if (threadIdx.x == 0 && blockIdx.x == 0) adam_tick(tick);
}

// ExponentialDecayScheduler of the "means" group (config.py:46-51) evaluated from the device step
// counter: lr = exp((1-t) log lr_init + t log lr_final), t = clip(step / max_steps, 0, 1), step = the
// number of optimiser steps taken so far (the scheduler advances after optimizer.step()).
__global__ void lr_exp_decay_kernel(float* __restrict__ lr_slot, const float* __restrict__ state, float log_init,
                                    float log_final, float inv_max_steps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float t = fminf(fmaxf(state[0] * inv_max_steps, 0.f), 1.f);
        lr_slot[0] = expf(log_init * (1.f - t) + log_final * t);
    }
}

}  // namespace qed

using namespace qed;

extern "C" int qed_version(void) { return QED_ABI_VERSION; }
extern "C" const char* qed_last_error(void) { return g_err; }

// The device-side address of a pinned (page-locked, mapped) host allocation: what a kernel must be given to store into
// it.  Equal to the host address for hipHostMalloc'ed memory under unified addressing, but not for memory registered
// after the fact (hipHostRegister; PYTORCH_CUDA_ALLOC_CONF=pinned_use_cuda_host_register) -- ask the runtime.
extern "C" int qed_host_device_pointer(void* host, void** device) {
    QED_REQUIRE(host && device, "null pointers");
    void* d = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&d, host, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        qed::set_error("qed_host_device_pointer: %s", hipGetErrorString(e));
        return QED_E_INVALID_ARG;
    }
    *device = d;
    return QED_OK;
}

static unsigned stream_grid(long long n_items, long long cap = 2048) {
    long long g = (n_items + 255) / 256;
    if (g > cap) g = cap;        // 256 CUs x 8 workgroups, grid-stride the rest
    if (g < 1) g = 1;
    return (unsigned)g;
}
// both loss passes use the same grid (at most kLossMaxGrid workgroups: one slot of per-workgroup partials each):
// pass 2 reads pass 1's partials by index
static unsigned reduce_grid(long long n_items) { return stream_grid(n_items, kLossMaxGrid); }

extern "C" int qed_loss_reduce(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                               const float* background, const float* gt_rgb, const float* gt_depth,
                               const float* mask, float* sums, void* stream) {
    QED_REQUIRE(n_pix > 0 && (channels == 3 || channels == 4), "bad arguments");
    QED_REQUIRE(render && alpha && background && gt_rgb && sums, "null buffers");
    QED_REQUIRE(channels == 3 || gt_depth, "gt_depth required with a depth channel");
    hipStream_t st = (hipStream_t)stream;
    if (channels == 4)
        hipLaunchKernelGGL(loss_reduce_kernel<4>, dim3(reduce_grid(n_pix)), dim3(256), 0, st, n_pix, render, alpha,
                           background, gt_rgb, gt_depth, mask, sums);
    else
        hipLaunchKernelGGL(loss_reduce_kernel<3>, dim3(reduce_grid(n_pix)), dim3(256), 0, st, n_pix, render, alpha,
                           background, gt_rgb, gt_depth, mask, sums);
    return check_launch("qed_loss_reduce");
}

extern "C" int qed_loss_grad(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                             const float* background, const float* gt_rgb, const float* gt_depth, const float* mask,
                             const float* sums, float rgb_weight, float depth_lambda, float* v_render,
                             float* v_alpha, float* losses, const float* v_rgb_extra, const float* extra_sum,
                             int32_t extra_n, float extra_scale, float extra_offset, void* stream) {
    QED_REQUIRE(n_pix > 0 && (channels == 3 || channels == 4), "bad arguments");
    QED_REQUIRE(render && alpha && background && gt_rgb && sums && v_render && v_alpha && losses, "null buffers");
    QED_REQUIRE(channels == 3 || gt_depth, "gt_depth required with a depth channel");
    hipStream_t st = (hipStream_t)stream;
    float* sums_rw = const_cast<float*>(sums);     // partial rows 2, 3 and the totals are written by this pass
    const unsigned grid = reduce_grid(n_pix);
    if (channels == 4)
        hipLaunchKernelGGL(loss_grad_kernel<4>, dim3(grid), dim3(256), 0, st, n_pix, render, alpha, background, gt_rgb,
                           gt_depth, mask, sums_rw, rgb_weight, depth_lambda, v_render, v_alpha, v_rgb_extra);
    else
        hipLaunchKernelGGL(loss_grad_kernel<3>, dim3(grid), dim3(256), 0, st, n_pix, render, alpha, background, gt_rgb,
                           gt_depth, mask, sums_rw, rgb_weight, depth_lambda, v_render, v_alpha, v_rgb_extra);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, n_pix, (int)grid, channels == 4 ? 1 : 0, sums_rw,
                       rgb_weight, depth_lambda, losses, extra_sum, (int)extra_n, extra_scale, extra_offset, AdamTick{nullptr, 0.f, 0.f, nullptr, 0.f, 0.f, 0.f, nullptr});
    return check_launch("qed_loss_grad");
}


extern "C" int qed_post_process_fwd(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                                    const float* background, float* rgb, float* depth, float* workspace,
                                    void* stream) {
    QED_REQUIRE(n_pix > 0 && (channels == 3 || channels == 4), "bad arguments");
    QED_REQUIRE(render && alpha && background && rgb, "null buffers");
    QED_REQUIRE(channels == 3 || (depth && workspace), "depth output and workspace required with a depth channel");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = reduce_grid(n_pix);
    if (channels == 4) {
        hipLaunchKernelGGL(post_max_kernel, dim3(grid), dim3(256), 0, st, n_pix, render, workspace);
        hipLaunchKernelGGL(post_fwd_kernel<4>, dim3(stream_grid(n_pix)), dim3(256), 0, st, n_pix, render, alpha,
                           background, workspace, (int)grid, rgb, depth);
    } else {
        hipLaunchKernelGGL(post_fwd_kernel<3>, dim3(stream_grid(n_pix)), dim3(256), 0, st, n_pix, render, alpha,
                           background, workspace, 0, rgb, depth);
    }
    return check_launch("qed_post_process_fwd");
}

extern "C" int qed_post_process_bwd(int32_t n_pix, int32_t channels, const float* render, const float* alpha,
                                    const float* background, const float* v_rgb, const float* v_depth,
                                    float* v_render, float* v_alpha, void* stream) {
    QED_REQUIRE(n_pix > 0 && (channels == 3 || channels == 4), "bad arguments");
    QED_REQUIRE(render && alpha && background && v_render && v_alpha, "null buffers");
    hipStream_t st = (hipStream_t)stream;
    if (channels == 4)
        hipLaunchKernelGGL(post_bwd_kernel<4>, dim3(stream_grid(n_pix)), dim3(256), 0, st, n_pix, render, alpha,
                           background, v_rgb, v_depth, v_render, v_alpha);
    else
        hipLaunchKernelGGL(post_bwd_kernel<3>, dim3(stream_grid(n_pix)), dim3(256), 0, st, n_pix, render, alpha,
                           background, v_rgb, v_depth, v_render, v_alpha);
    return check_launch("qed_post_process_bwd");
}

extern "C" int qed_image_losses_fwd(int32_t n_pix, const float* rgb, const float* depth, const float* gt_rgb,
                                    const float* gt_depth, const float* mask, float rgb_weight, float depth_lambda,
                                    const float* extra_sum, int32_t extra_n, float extra_scale, float extra_offset,
                                    float* sums, float* losses, void* stream) {
    QED_REQUIRE(n_pix > 0, "bad arguments");
    QED_REQUIRE(rgb && gt_rgb && sums && losses, "null buffers");
    QED_REQUIRE(depth == nullptr || gt_depth, "gt_depth required with a depth image");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = reduce_grid(n_pix);
    hipLaunchKernelGGL(image_loss_reduce_kernel, dim3(grid), dim3(256), 0, st, n_pix, rgb, depth, gt_rgb, gt_depth, mask,
                       sums);
    // has_depth = -1: fold row 0 (the valid count) but no row of maxima
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, n_pix, (int)grid, -1, sums, rgb_weight,
                       depth_lambda, losses, extra_sum, (int)extra_n, extra_scale, extra_offset, AdamTick{nullptr, 0.f, 0.f, nullptr, 0.f, 0.f, 0.f, nullptr});
    return check_launch("qed_image_losses_fwd");
}

extern "C" int qed_image_losses_bwd(int32_t n_pix, const float* rgb, const float* depth, const float* gt_rgb,
                                    const float* gt_depth, const float* mask, const float* sums, float rgb_weight,
                                    float depth_lambda, const float* g_main, const float* g_depth,
                                    int32_t accumulate, float* v_rgb, float* v_depth, void* stream) {
    QED_REQUIRE(n_pix > 0, "bad arguments");
    QED_REQUIRE(rgb && gt_rgb && sums, "null buffers");
    QED_REQUIRE(v_depth == nullptr || (depth && gt_depth), "depth images required for a depth gradient");
    QED_REQUIRE(!accumulate || v_rgb, "accumulate needs v_rgb");
    if (v_rgb == nullptr && v_depth == nullptr) return QED_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(image_loss_grad_kernel, dim3(stream_grid(n_pix)), dim3(256), 0, st, n_pix, rgb, depth, gt_rgb,
                       gt_depth, mask, sums, rgb_weight, depth_lambda, g_main, g_depth, (int)accumulate, v_rgb, v_depth);
    return check_launch("qed_image_losses_bwd");
}

static int adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t n_groups,
                       const int64_t* h_group_begin, const float* h_lr, double beta1, double beta2, float eps,
                       int32_t step, float* dev_state, const float* dev_lr, const int32_t* skip, void* stream);

extern "C" int qed_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t n_groups,
                             const int64_t* h_group_begin, const float* h_lr, double beta1, double beta2, float eps,
                             int32_t step, const int32_t* skip_flag, void* stream) {
    QED_REQUIRE(h_lr && step >= 1, "bad arguments");
    return adam_launch(params, grads, exp_avg, exp_avg_sq, n_groups, h_group_begin, h_lr, beta1, beta2, eps, step,
                       nullptr, nullptr, skip_flag, stream);
}

extern "C" int qed_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                 int32_t n_groups, const int64_t* h_group_begin, const float* dev_lr, double beta1,
                                 double beta2, float eps, float* dev_state, const int32_t* skip_flag, void* stream) {
    QED_REQUIRE(dev_lr && dev_state, "device lr / state required");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       AdamTick{dev_state, (float)beta1, (float)beta2, nullptr, 0.f, 0.f, 0.f, skip_flag});
    return adam_launch(params, grads, exp_avg, exp_avg_sq, n_groups, h_group_begin, nullptr, beta1, beta2, eps, 1,
                       dev_state, dev_lr, skip_flag, stream);
}

extern "C" int qed_adam_step_sh(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t n_groups,
                                const int64_t* h_group_begin, const float* h_lr, float* dev_lr, double beta1,
                                double beta2, float eps, int32_t step, float* dev_state, int32_t sched_group,
                                float sched_lr_init, float sched_lr_final, int32_t sched_max_steps, int32_t N,
                                int32_t sh_degree, const float* means, int32_t n_views, const float* viewmats,
                                int64_t viewmat_stride, const float* v_views, int64_t view_stride, float scale,
                                int32_t parts, const int32_t* skip_flag, void* stream) {
    QED_REQUIRE(n_groups >= 2 && n_groups <= 8 && h_group_begin, "2..8 groups, the last two features_dc, features_rest");
    QED_REQUIRE((dev_state != nullptr) == (dev_lr != nullptr), "device state and device rates go together");
    QED_REQUIRE(dev_state || (h_lr && step >= 1), "host rates and a 1-based step, or device state");
    QED_REQUIRE(params && exp_avg && exp_avg_sq && N > 0 && means && n_views >= 1 && viewmats && v_views,
                "bad arguments");
    QED_REQUIRE((parts & 3) != 0 && parts >= 1 && parts <= 7, "parts: QED_ADAM_PART_SH | QED_ADAM_PART_LEADING [| QED_ADAM_PART_TICKED]");
    QED_REQUIRE(!(parts & QED_ADAM_PART_TICKED) || dev_state, "QED_ADAM_PART_TICKED is about the device-resident state");
    QED_REQUIRE(grads || n_groups == 2 || !(parts & QED_ADAM_PART_LEADING), "gradients of the leading groups required");
    QED_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "buffers must be 16-byte aligned");
    QED_REQUIRE(sched_group < n_groups && (sched_group < 0 || (dev_state && sched_lr_init > 0.f && sched_lr_final > 0.f &&
                                                                sched_max_steps > 0)),
                "a scheduled group needs device state, positive rates and max_steps (host state: schedule h_lr)");
    AdamRows grp;
    QED_REQUIRE(h_group_begin[0] == 0, "group 0 must start at element 0");
    for (int i = 0; i < 9; ++i) grp.begin[i] = h_group_begin[i < n_groups ? i : n_groups];
    for (int i = 0; i < 8; ++i) {
        grp.lr[i] = (h_lr != nullptr && i < n_groups) ? h_lr[i] : 0.f;
        const long long len = grp.begin[i + 1] - grp.begin[i];
        QED_REQUIRE(len >= 0 && len % N == 0, "every group must be [N, width]");
        grp.width[i] = (int)(len / N);
    }
    grp.n = n_groups;
    QED_REQUIRE(grp.width[n_groups - 2] == 3, "features_dc must be [N,3]");
    const int RW = grp.width[n_groups - 1];
    QED_REQUIRE(RW % 3 == 0 && RW <= 45, "features_rest must be [N,KR,3] with KR <= 15");
    QED_REQUIRE(sh_degree >= 0 && sh_degree <= 3 && 3 * ((sh_degree + 1) * (sh_degree + 1) - 1) <= RW,
                "sh_degree 0..3 within the stored coefficient rows");
    hipStream_t st = (hipStream_t)stream;
    AdamCoef co = adam_coef(beta1, beta2, eps, 1.f, 1.f);
    if (!(parts & QED_ADAM_PART_SH)) {                 // the leading groups only: the SH part of this step has ticked
        if (n_groups == 2) return QED_OK;
        return adam_launch(params, grads, exp_avg, exp_avg_sq, n_groups - 2, h_group_begin, h_lr, beta1, beta2, eps,
                           dev_state ? 1 : step, dev_state, dev_lr, skip_flag, stream);
    }
    if (dev_state != nullptr) {
        if (!(parts & QED_ADAM_PART_TICKED)) {           // (else qed_loss_grad_ssim's fold launch has advanced the state)
            const bool sched = sched_group >= 0;
            hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st,
                               AdamTick{dev_state, (float)beta1, (float)beta2,
                                        sched ? dev_lr + sched_group : (float*)nullptr,
                                        sched ? logf(sched_lr_init) : 0.f, sched ? logf(sched_lr_final) : 0.f,
                                        sched ? 1.f / (float)sched_max_steps : 0.f, skip_flag});
        }
    } else {
        co.inv_bc1 = (float)(1.0 / (1.0 - pow(beta1, (double)step)));
        co.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
    }
    // 3 workgroups per CU (46 KB of LDS each; measured at 500 k, both launches: 256 -> 150 us, 512 -> 125 us,
    // 768 -> 121 us, 1024 -> 132 us).  Folding the leading groups into the same launch was slower in two
    // forms (their rows per pass: 139 us; grid-wide per-group passes + means rows per pass: 154 us): short spans
    // expose one memory latency each.
    constexpr int grid_cap = 768;
    const int n_chunks = (N + kShChunk - 1) / kShChunk;
    const unsigned grid = (unsigned)(n_chunks < grid_cap ? n_chunks : grid_cap);
    const size_t lds = ((size_t)kShChunk * RW + 4) * sizeof(float);
    // the SH pass reads `means`, which the pass over the leading groups updates: SH first
#define QED_LAUNCH_ASH(D)                                                                                            \
    hipLaunchKernelGGL(adam_sh_kernel<D>, dim3(grid), dim3(256), lds, st, params, exp_avg, exp_avg_sq,                \
                       grp.begin[n_groups - 2], grp.begin[n_groups - 1], RW, N, means, n_views, viewmats,            \
                       (long long)viewmat_stride, v_views, (long long)view_stride, scale, co,                        \
                       (const float*)dev_state, (const float*)dev_lr, n_groups - 2, grp.lr[n_groups - 2],            \
                       grp.lr[n_groups - 1], skip_flag)
    switch (sh_degree) {
        case 0: QED_LAUNCH_ASH(0); break;
        case 1: QED_LAUNCH_ASH(1); break;
        case 2: QED_LAUNCH_ASH(2); break;
        default: QED_LAUNCH_ASH(3); break;
    }
#undef QED_LAUNCH_ASH
    if (n_groups == 2 || !(parts & QED_ADAM_PART_LEADING)) return check_launch("qed_adam_step_sh");
    return adam_launch(params, grads, exp_avg, exp_avg_sq, n_groups - 2, h_group_begin, h_lr, beta1, beta2, eps,
                       dev_state ? 1 : step, dev_state, dev_lr, skip_flag, stream);
}

extern "C" int qed_lr_exp_decay_dev(float* dev_lr_slot, const float* dev_state, float lr_init, float lr_final,
                                    int32_t max_steps, void* stream) {
    QED_REQUIRE(dev_lr_slot && dev_state && lr_init > 0.f && lr_final > 0.f && max_steps > 0, "bad arguments");
    hipLaunchKernelGGL(lr_exp_decay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_lr_slot, dev_state,
                       logf(lr_init), logf(lr_final), 1.f / (float)max_steps);
    return check_launch("qed_lr_exp_decay_dev");
}

static int adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t n_groups,
                       const int64_t* h_group_begin, const float* h_lr, double beta1, double beta2, float eps,
                       int32_t step, float* dev_state, const float* dev_lr, const int32_t* skip, void* stream) {
    QED_REQUIRE(n_groups >= 1 && n_groups <= 8, "1..8 parameter groups");
    QED_REQUIRE(params && grads && exp_avg && exp_avg_sq && h_group_begin && step >= 1, "bad arguments");
    QED_REQUIRE((((uintptr_t)params | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 && ((uintptr_t)grads & 3) == 0,
                "params / moments must be 16-byte aligned (grads: 4)");
    AdamGroups grp;
    for (int i = 0; i <= n_groups; ++i) grp.begin[i] = h_group_begin[i];
    for (int i = n_groups + 1; i < 9; ++i) grp.begin[i] = h_group_begin[n_groups];
    for (int i = 0; i < 8; ++i) grp.lr[i] = (h_lr != nullptr && i < n_groups) ? h_lr[i] : 0.f;
    grp.n = n_groups;
    QED_REQUIRE(grp.begin[0] == 0, "group 0 must start at element 0");
    const long long total = grp.begin[n_groups];
    if (total == 0) return QED_OK;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    // 2 workgroups per CU: measured 136 us (6.1 TB/s) against 157 us with 8 per CU and 196 us with 1 --
    // seven concurrent streams per wave favour fewer, longer-running waves (DRAM page locality)
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(total / 4 + 1, 512)), dim3(256), 0, (hipStream_t)stream, params, grads,
                       exp_avg, exp_avg_sq, grp, adam_coef(beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2))),
                       (const float*)dev_state, dev_lr, skip);
    return check_launch("qed_adam_step");
}
