// the per-step evaluation metrics of get_metrics_dict (model.py:120-197) as one
// streaming pass with the results left in device memory (the reference pulls ~12 scalars to the host
// with .item()/float() every iteration, model.py:160-182).
//
//   RGB   (metrics.py:84-112, torchmetrics PeakSignalNoiseRatio(data_range=1), nn.MSELoss):
//         mse = mean (pred - gt)^2 over H*W*3,  psnr = 10 log10(1 / mse)
//   depth (metrics.py:115-156): valid = finite(pred) & finite(gt) & gt > tolerance;
//         abs_rel = mean |gt-pred|/gt, sq_rel = mean (gt-pred)^2/gt, rmse = sqrt(mean (gt-pred)^2),
//         rmse_log = sqrt(nanmean (log gt - log pred)^2), a_k = mean [max(gt/pred, pred/gt) < 1.25^k];
//         all NaN when no pixel is valid (metrics.py:134-143).
//   SSIM  is qed_ssim_fwd's value (torchmetrics' reflect-pad + crop equals the unpadded "valid"
//         window sums); LPIPS needs pretrained network weights and is out of scope.
#include "qed_common.h"

namespace qed {

constexpr int kMetricCols = 10;
constexpr int kMetricMaxGrid = QED_METRICS_WS_DOUBLES / kMetricCols;
// columns: 0 sum (dr^2+dg^2+db^2) | 1 n_valid | 2 sum |g-p|/g | 3 sum (g-p)^2/g | 4 sum (g-p)^2
//          5 sum (log g - log p)^2 over non-NaN | 6 count non-NaN | 7,8,9 counts a1,a2,a3
// Each workgroup writes its ten partial sums to its own slots (workspace[col][block]) and a second,
// one-workgroup launch folds them: same-address atomics serialise (~12 ns each; ten per workgroup made
// this pass 103 us at 1080p instead of ~15).

// per-wave partial of slot `slot` -> s_tmp[slot][wave]
__device__ __forceinline__ void park_wave_sum(float v, double* s_tmp, int slot) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_tmp[slot * 4 + (threadIdx.x >> 6)] = (double)v;
}

__global__ void __launch_bounds__(256)
metrics_kernel(int n_pix, const float* __restrict__ pred_rgb, const float* __restrict__ gt_rgb,
               const float* __restrict__ pred_depth, const float* __restrict__ gt_depth, float tolerance,
               double* __restrict__ partials) {
    float acc[kMetricCols];
#pragma unroll
    for (int i = 0; i < kMetricCols; ++i) acc[i] = 0.f;
    // kU pixels per trip, every load of the trip issued before the first use (clamped index, contribution switched off
    // by `in`): one memory round trip per kU pixels instead of one per pixel -- with at most 1024 workgroups a thread
    // walks ~8 pixels at 1080p and the pass was bound by those serial round trips (25 us for 66 MB)
    constexpr int kU = 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < (size_t)n_pix; i0 += kU * stride) {
        float pr[kU][3], gr[kU][3], pd[kU], gd[kU];
        bool in[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t iu = i0 + u * stride;
            in[u] = iu < (size_t)n_pix;
            const size_t i = in[u] ? iu : i0;
            if (pred_rgb != nullptr) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { pr[u][k] = pred_rgb[3 * i + k]; gr[u][k] = gt_rgb[3 * i + k]; }
            }
            if (pred_depth != nullptr) { pd[u] = pred_depth[i]; gd[u] = gt_depth[i]; }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (!in[u]) continue;
            if (pred_rgb != nullptr) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float d = pr[u][k] - gr[u][k];
                    acc[0] += d * d;
                }
            }
            if (pred_depth != nullptr) {
                const float p = pd[u], g = gd[u];
                if (isfinite(p) && isfinite(g) && g > tolerance) {
                    const float d = g - p;
                    acc[1] += 1.f;
                    acc[2] += fabsf(d) / g;
                    acc[3] += d * d / g;
                    acc[4] += d * d;
                    const float l = logf(g) - logf(p);          // NaN for p < 0, +-inf for p == 0 (kept, like nanmean)
                    const float l2 = l * l;
                    if (!isnan(l2)) { acc[5] += l2; acc[6] += 1.f; }
                    const float t = fmaxf(g / p, p / g);
                    acc[7] += t < 1.25f ? 1.f : 0.f;
                    acc[8] += t < 1.25f * 1.25f ? 1.f : 0.f;
                    acc[9] += t < 1.25f * 1.25f * 1.25f ? 1.f : 0.f;
                }
            }
        }
    }
    __shared__ double s_tmp[kMetricCols * 4];
#pragma unroll
    for (int i = 0; i < kMetricCols; ++i) park_wave_sum(acc[i], s_tmp, i);
    __syncthreads();
    if (threadIdx.x < kMetricCols) {
        const int i = threadIdx.x;
        partials[(size_t)i * kMetricMaxGrid + blockIdx.x] = s_tmp[4 * i] + s_tmp[4 * i + 1] + s_tmp[4 * i + 2] + s_tmp[4 * i + 3];
    }
}

__global__ void __launch_bounds__(256)
metrics_finalize_kernel(int n_pix, int n_blocks, int has_rgb, int has_depth, const double* __restrict__ partials,
                        float* __restrict__ out) {
    __shared__ double s_w[kMetricCols][4];
    __shared__ double s[kMetricCols];
    double v[kMetricCols];
#pragma unroll
    for (int c = 0; c < kMetricCols; ++c) v[c] = 0.0;
    // (all ten columns of a row of partials are requested together: ten dependent load / reduce rounds made this
    // one-workgroup launch 15 us)
    for (int b = threadIdx.x; b < n_blocks; b += 256) {
#pragma unroll
        for (int c = 0; c < kMetricCols; ++c) v[c] += partials[(size_t)c * kMetricMaxGrid + b];
    }
#pragma unroll
    for (int c = 0; c < kMetricCols; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_xor(v[c], o, 64);
        if ((threadIdx.x & 63) == 0) s_w[c][threadIdx.x >> 6] = v[c];
    }
    __syncthreads();
    if (threadIdx.x < kMetricCols) s[threadIdx.x] = s_w[threadIdx.x][0] + s_w[threadIdx.x][1] + s_w[threadIdx.x][2] + s_w[threadIdx.x][3];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float nanv = __builtin_nanf("");
    if (has_rgb) {
        const double mse = s[0] / (3.0 * (double)n_pix);
        out[0] = (float)mse;
        out[1] = (float)(10.0 * log10(1.0 / mse));
    } else {
        out[0] = nanv; out[1] = nanv;
    }
    const double n = s[1];
    if (has_depth && n > 0.0) {
        out[2] = (float)(s[2] / n);
        out[3] = (float)(s[3] / n);
        out[4] = (float)sqrt(s[4] / n);
        out[5] = s[6] > 0.0 ? (float)sqrt(s[5] / s[6]) : nanv;
        out[6] = (float)(s[7] / n);
        out[7] = (float)(s[8] / n);
        out[8] = (float)(s[9] / n);
    } else {
        for (int i = 2; i < 9; ++i) out[i] = nanv;
    }
    out[9] = (float)n;
}

// model.py:192-194 `torch.nanmean(torch.exp(self.scales[..., -1]))`: ~8 eager launches over N values per step in the
// reference; here one pass + a one-workgroup fold.  partials: [2][kMetricMaxGrid] doubles (sum, count of non-NaN).
__global__ void __launch_bounds__(256)
nanmean_exp_kernel(int n, const float* __restrict__ x, int stride, double* __restrict__ partials) {
    float sum = 0.f, cnt = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) {
        const float e = expf(x[i * (size_t)stride]);
        if (!isnan(e)) { sum += e; cnt += 1.f; }
    }
    __shared__ double s_tmp[2 * 4];
    park_wave_sum(sum, s_tmp, 0);
    park_wave_sum(cnt, s_tmp, 1);
    __syncthreads();
    if (threadIdx.x < 2) {
        const int i = threadIdx.x;
        partials[(size_t)i * kMetricMaxGrid + blockIdx.x] = s_tmp[4 * i] + s_tmp[4 * i + 1] + s_tmp[4 * i + 2] + s_tmp[4 * i + 3];
    }
}

__global__ void __launch_bounds__(256)
nanmean_exp_finalize_kernel(int n_blocks, const double* __restrict__ partials, float* __restrict__ out) {
    __shared__ double s_w[2][4];
    for (int c = 0; c < 2; ++c) {
        double v = 0.0;
        for (int b = threadIdx.x; b < n_blocks; b += 256) v += partials[(size_t)c * kMetricMaxGrid + b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) s_w[c][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sum = s_w[0][0] + s_w[0][1] + s_w[0][2] + s_w[0][3], cnt = s_w[1][0] + s_w[1][1] + s_w[1][2] + s_w[1][3];
        out[0] = cnt > 0.0 ? (float)(sum / cnt) : __builtin_nanf("");      // torch.nanmean of nothing is NaN
    }
}

// ---- everything get_metrics_dict needs from the images in ONE pass + ONE fold (qed_step_metrics) -------------------
// The reference-shaped route is made of short launches: PSNR / depth metrics (2 launches), the sum of the SSIM map's
// per-workgroup partials and its division (2 eager launches), nanmean(exp(scales[:, -1])) (2 launches) and, in
// get_loss_dict right behind, the L1 / depth-L1 sums of the SAME images (2 launches) -- eight launches of 4-19 us where two
// do.  Columns 0-9 as metrics_kernel; 10 sum |m rgb - m gt|, 11 sum |m d - m dgt| over valid, 12 their count (the loss
// terms of qed_image_losses_fwd, mask m); 13 sum exp(scale), 14 count of non-NaN.
constexpr int kStepCols = 15;
constexpr int kStepMaxGrid = QED_STEP_METRICS_WS_DOUBLES / 16;

__global__ void __launch_bounds__(256)
step_metrics_kernel(int n_pix, const float* __restrict__ pred_rgb, const float* __restrict__ gt_rgb,
                    const float* __restrict__ pred_depth, const float* __restrict__ gt_depth, float tolerance,
                    const float* __restrict__ mask, int want_loss, const float* __restrict__ scales, int n_scales,
                    int scale_stride, double* __restrict__ partials) {
    float acc[kStepCols];
#pragma unroll
    for (int i = 0; i < kStepCols; ++i) acc[i] = 0.f;
    constexpr int kU = 4;                                    // pixels per trip, all loads of a trip issued before the first use
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < (size_t)n_pix; i0 += kU * stride) {
        float pr[kU][3], gr[kU][3], pd[kU], gd[kU], mk[kU];
        bool in[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t iu = i0 + u * stride;
            in[u] = iu < (size_t)n_pix;
            const size_t i = in[u] ? iu : i0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { pr[u][k] = pred_rgb[3 * i + k]; gr[u][k] = gt_rgb[3 * i + k]; }
            pd[u] = pred_depth != nullptr ? pred_depth[i] : 0.f;
            gd[u] = pred_depth != nullptr ? gt_depth[i] : 0.f;
            mk[u] = mask != nullptr ? mask[i] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (!in[u]) continue;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = pr[u][k] - gr[u][k];
                acc[0] += d * d;
                if (want_loss) acc[10] += fabsf(pr[u][k] * mk[u] - gr[u][k] * mk[u]);
            }
            if (pred_depth != nullptr) {
                const float p = pd[u], g = gd[u];
                if (isfinite(p) && isfinite(g) && g > tolerance) {
                    const float d = g - p;
                    acc[1] += 1.f;
                    acc[2] += fabsf(d) / g;
                    acc[3] += d * d / g;
                    acc[4] += d * d;
                    const float l = logf(g) - logf(p);
                    const float l2 = l * l;
                    if (!isnan(l2)) { acc[5] += l2; acc[6] += 1.f; }
                    const float t = fmaxf(g / p, p / g);
                    acc[7] += t < 1.25f ? 1.f : 0.f;
                    acc[8] += t < 1.25f * 1.25f ? 1.f : 0.f;
                    acc[9] += t < 1.25f * 1.25f * 1.25f ? 1.f : 0.f;
                }
                if (want_loss) {                               // model.py:93-105: masked depths, finite, target > 0
                    const float dp = p * mk[u], dg = g * mk[u];
                    if (isfinite(dp) && isfinite(dg) && dg > 0.f) { acc[11] += fabsf(dp - dg); acc[12] += 1.f; }
                }
            }
        }
    }
    if (scales != nullptr) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n_scales; i += stride) {
            const float e = expf(scales[i * (size_t)scale_stride]);
            if (!isnan(e)) { acc[13] += e; acc[14] += 1.f; }
        }
    }
    __shared__ double s_tmp[kStepCols * 4];
#pragma unroll
    for (int i = 0; i < kStepCols; ++i) park_wave_sum(acc[i], s_tmp, i);
    __syncthreads();
    if (threadIdx.x < kStepCols) {
        const int i = threadIdx.x;
        // (a workgroup's sum of <= 2 048 pixels' terms: exact enough as a float; the fold adds the 1 024 of them in double)
        reinterpret_cast<float*>(partials)[(size_t)i * kStepMaxGrid + blockIdx.x] =
            (float)(s_tmp[4 * i] + s_tmp[4 * i + 1] + s_tmp[4 * i + 2] + s_tmp[4 * i + 3]);
    }
}

__global__ void __launch_bounds__(256)
step_metrics_finalize_kernel(int n_pix, int n_blocks, int has_depth, const double* __restrict__ partials,
                             const float* __restrict__ ssim_sum, int ssim_n, float ssim_norm, int has_scales,
                             float rgb_weight, float depth_lambda, float ssim_lambda, float* __restrict__ loss_sums,
                             float* __restrict__ losses, float* __restrict__ out) {
    // 256 threads x four rows of partials: all sixty values of a thread (and its share of the SSIM partials) are REQUESTED
    // before anything is added -- one memory round trip for the whole fold.  (A loop of load / add rounds took 10.7 us;
    // 1 024 threads with one row each 14 us: sixteen waves of double-precision shuffles.)
    static_assert(kStepMaxGrid == 1024, "four rows per thread");
    __shared__ double s_w[kStepCols + 1][4];
    __shared__ double s[kStepCols + 1];
    const float* pf = reinterpret_cast<const float*>(partials);
    float r[4][kStepCols];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = threadIdx.x + 256 * j;
#pragma unroll
        for (int c = 0; c < kStepCols; ++c) r[j][c] = pf[(size_t)c * kStepMaxGrid + (b < n_blocks ? b : 0)];
    }
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        e[j] = ssim_sum != nullptr ? ssim_sum[(int)threadIdx.x + 256 * j < ssim_n ? threadIdx.x + 256 * j : 0] : 0.f;
    double v[kStepCols + 1];
#pragma unroll
    for (int c = 0; c < kStepCols; ++c) {
        v[c] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[c] += (int)threadIdx.x + 256 * j < n_blocks ? (double)r[j][c] : 0.0;
    }
    v[kStepCols] = 0.0;
    if (ssim_sum != nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[kStepCols] += (int)threadIdx.x + 256 * j < ssim_n ? (double)e[j] : 0.0;
        for (int b0 = threadIdx.x + 2048; b0 < ssim_n; b0 += 256) v[kStepCols] += (double)ssim_sum[b0];
    }
#pragma unroll
    for (int c = 0; c <= kStepCols; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_xor(v[c], o, 64);
        if ((threadIdx.x & 63) == 0) s_w[c][threadIdx.x >> 6] = v[c];
    }
    __syncthreads();
    if (threadIdx.x <= kStepCols) s[threadIdx.x] = s_w[threadIdx.x][0] + s_w[threadIdx.x][1] + s_w[threadIdx.x][2] + s_w[threadIdx.x][3];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float nanv = __builtin_nanf("");
    const double mse = s[0] / (3.0 * (double)n_pix);
    out[0] = (float)mse;
    out[1] = (float)(10.0 * log10(1.0 / mse));
    const double n = s[1];
    if (has_depth && n > 0.0) {
        out[2] = (float)(s[2] / n);
        out[3] = (float)(s[3] / n);
        out[4] = (float)sqrt(s[4] / n);
        out[5] = s[6] > 0.0 ? (float)sqrt(s[5] / s[6]) : nanv;
        out[6] = (float)(s[7] / n);
        out[7] = (float)(s[8] / n);
        out[8] = (float)(s[9] / n);
    } else {
        for (int i = 2; i < 9; ++i) out[i] = nanv;
    }
    out[9] = (float)n;
    out[10] = ssim_sum != nullptr ? (float)(s[kStepCols] * (double)ssim_norm) : nanv;
    out[11] = has_scales && s[14] > 0.0 ? (float)(s[13] / s[14]) : nanv;          // torch.nanmean of nothing is NaN
    if (losses != nullptr) {
        // what qed_image_losses_fwd leaves: the two scalar losses, their sum, and sums[2] = n_valid for the backward pass
        const float tot_l1 = (float)s[10], tot_d = (float)s[11], nvalid = (float)s[12];
        loss_sums[0] = tot_l1; loss_sums[1] = tot_d; loss_sums[2] = nvalid; loss_sums[3] = 0.f;
        losses[0] = rgb_weight * tot_l1 / (3.f * (float)n_pix);
        if (ssim_sum != nullptr && ssim_lambda > 0.f)
            losses[0] += ssim_lambda - ssim_lambda * (float)(s[kStepCols] * (double)ssim_norm);
        losses[1] = nvalid > 0.f ? depth_lambda * tot_d / nvalid : 0.f;             // empty -> 0.0 (model.py:111-114)
        losses[2] = losses[0] + losses[1];
    }
}

}  // namespace qed

using namespace qed;

extern "C" int qed_nanmean_exp(int32_t n, const float* x, int32_t stride, double* workspace, float* out, void* stream) {
    QED_REQUIRE(n >= 0 && stride >= 1 && workspace && out, "bad arguments");
    QED_REQUIRE(n == 0 || x, "null input");
    hipStream_t st = (hipStream_t)stream;
    long long g = ((long long)n + 255) / 256;
    if (g > kMetricMaxGrid) g = kMetricMaxGrid;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(nanmean_exp_kernel, dim3((unsigned)g), dim3(256), 0, st, n, x, stride, workspace);
    hipLaunchKernelGGL(nanmean_exp_finalize_kernel, dim3(1), dim3(256), 0, st, (int)g, (const double*)workspace, out);
    return check_launch("qed_nanmean_exp");
}

extern "C" int qed_image_metrics(int32_t n_pix, const float* pred_rgb, const float* gt_rgb, const float* pred_depth,
                                 const float* gt_depth, float tolerance, double* workspace, float* out, void* stream) {
    QED_REQUIRE(n_pix > 0 && workspace && out, "bad arguments");
    QED_REQUIRE((pred_rgb == nullptr) == (gt_rgb == nullptr), "pred_rgb and gt_rgb go together");
    QED_REQUIRE((pred_depth == nullptr) == (gt_depth == nullptr), "pred_depth and gt_depth go together");
    QED_REQUIRE(pred_rgb || pred_depth, "nothing to measure");
    hipStream_t st = (hipStream_t)stream;
    long long g = ((long long)n_pix + 255) / 256;
    if (g > kMetricMaxGrid) g = kMetricMaxGrid;
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)g), dim3(256), 0, st, n_pix, pred_rgb, gt_rgb, pred_depth, gt_depth,
                       tolerance, workspace);
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(256), 0, st, n_pix, (int)g, pred_rgb != nullptr ? 1 : 0,
                       pred_depth != nullptr ? 1 : 0, (const double*)workspace, out);
    return check_launch("qed_image_metrics");
}

extern "C" int qed_step_metrics(int32_t n_pix, const float* pred_rgb, const float* gt_rgb, const float* pred_depth,
                                const float* gt_depth, float tolerance, const float* ssim_sum, int32_t ssim_n,
                                float ssim_norm, const float* scales, int32_t n_scales, int32_t scale_stride,
                                const float* loss_mask, float rgb_weight, float depth_lambda, float ssim_lambda,
                                float* loss_sums, float* losses, double* workspace, float* out, void* stream) {
    QED_REQUIRE(n_pix > 0 && pred_rgb && gt_rgb && workspace && out, "bad arguments");
    QED_REQUIRE((pred_depth == nullptr) == (gt_depth == nullptr), "pred_depth and gt_depth go together");
    QED_REQUIRE(ssim_sum == nullptr || ssim_n > 0, "ssim_sum needs its length");
    QED_REQUIRE(scales == nullptr || (n_scales > 0 && scale_stride >= 1), "scales need a length and a stride");
    QED_REQUIRE((loss_sums == nullptr) == (losses == nullptr), "loss_sums and losses go together");
    QED_REQUIRE(losses != nullptr || loss_mask == nullptr, "a loss mask without loss outputs");
    hipStream_t st = (hipStream_t)stream;
    long long g = ((long long)n_pix + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(step_metrics_kernel, dim3((unsigned)g), dim3(256), 0, st, n_pix, pred_rgb, gt_rgb, pred_depth,
                       gt_depth, tolerance, loss_mask, losses != nullptr ? 1 : 0, scales, n_scales, scale_stride, workspace);
    hipLaunchKernelGGL(step_metrics_finalize_kernel, dim3(1), dim3(256), 0, st, n_pix, (int)g, pred_depth != nullptr ? 1 : 0,
                       (const double*)workspace, ssim_sum, ssim_n, ssim_norm, scales != nullptr ? 1 : 0, rgb_weight,
                       depth_lambda, ssim_lambda, loss_sums, losses, out);
    return check_launch("qed_step_metrics");
}
