// the (1 - SSIM) term of the parent's RGB loss, fused forward + backward.
//
// The reference's get_loss_dict (model.py:83-85) calls SplatfactoModel.get_loss_dict, whose main loss
// is (1 - ssim_lambda) * L1 + ssim_lambda * (1 - SSIM(gt, pred)) with pytorch_msssim's SSIM
// (data_range 1, 11-tap Gaussian window sigma 1.5 applied separably WITHOUT padding, K = (0.01, 0.03),
// mean over the (H-10) x (W-10) map and the 3 channels).  Neither library is vendored in the
// reference, so this follows the published definition; oracle/splat_oracle.py::ssim is
// the restatement the tests compare against.
//
// Forward: one 256-thread workgroup per 16x16 block of the SSIM map; the 26x26 input patch of both
// images goes through LDS, the window is applied separably (rows, then columns), and besides the
// block's partial sum of the SSIM map the kernel stores, per map pixel q and channel, the three
// coefficient maps the backward needs:
//     d ssim(q) / d x(p) = w(p-q) * [ A(q) + 2 x(p) B(q) + y(p) C(q) ]
//     A = dS/dmu_x - 2 mu_x dS/dvar_x - mu_y dS/dcov,  B = dS/dvar_x,  C = dS/dcov
// Backward: grad(p) = conv(A)(p) + 2 x(p) conv(B)(p) + y(p) conv(C)(p) ("full" correlation, zero
// outside the map), again separable through LDS.  HBM-bound: ~50 MB in, ~75 MB of maps, 25 MB out.
#include "qed_common.h"

namespace qed {

constexpr int kWin = 11;
constexpr int kHalo = kWin - 1;            // 10
constexpr int kSTile = 16;
constexpr int kSPatch = kSTile + kHalo;    // 26

// pytorch_msssim _fspecial_gauss_1d(11, 1.5) evaluated in fp32 exactly as the library does
__device__ __constant__ float c_win[kWin] = {
    1.028380357e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f, 2.130055279e-01f, 2.660117149e-01f,
    2.130055279e-01f, 1.093606874e-01f, 3.600077331e-02f, 7.598758209e-03f, 1.028380357e-03f};

// predicted colour channel k of pixel (iy, ix): either a plain [H,W,3] image or composited on the fly
// from render[H,W,CH] + (1 - alpha) * background, clamped to [0,1] (model.py:296-297)
template <bool COMPOSITE>
__device__ __forceinline__ float pred_at(const float* __restrict__ pred, const float* __restrict__ alpha,
                                         const float* __restrict__ bg, int channels, int W, int iy, int ix, int k) {
    const size_t pix = (size_t)iy * W + ix;
    if constexpr (COMPOSITE) {
        const float v = pred[pix * channels + k] + (1.f - alpha[pix]) * bg[k];
        return fminf(fmaxf(v, 0.f), 1.f);
    } else {
        return pred[pix * 3 + k];
    }
}

template <bool COMPOSITE>
__global__ void __launch_bounds__(256)
ssim_fwd_kernel(int H, int W, int channels, const float* __restrict__ pred, const float* __restrict__ alpha,
                const float* __restrict__ bg, const float* __restrict__ gt, float* __restrict__ maps,
                float* __restrict__ ssim_sum) {
    __shared__ float s_x[kSPatch][kSPatch + 1], s_y[kSPatch][kSPatch + 1];
    __shared__ float s_h[5][kSPatch][kSTile + 1];
    __shared__ float s_red[4];
    const int Ho = H - kHalo, Wo = W - kHalo;
    const int ox = blockIdx.x * kSTile, oy = blockIdx.y * kSTile;        // origin in the SSIM map
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const size_t n_out = (size_t)Ho * Wo;
    float acc = 0.f;
    for (int k = 0; k < 3; ++k) {
        __syncthreads();
        for (int i = tid; i < kSPatch * kSPatch; i += 256) {
            const int py = i / kSPatch, px = i - py * kSPatch;
            const int iy = oy + py, ix = ox + px;
            float x = 0.f, y = 0.f;
            if (iy < H && ix < W) {
                x = pred_at<COMPOSITE>(pred, alpha, bg, channels, W, iy, ix, k);
                y = gt[((size_t)iy * W + ix) * 3 + k];
            }
            s_x[py][px] = x; s_y[py][px] = y;
        }
        __syncthreads();
        // rows: 26 x 16 window sums of x, y, x^2, y^2, x y
        for (int i = tid; i < kSPatch * kSTile; i += 256) {
            const int py = i / kSTile, cx = i - py * kSTile;
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int d = 0; d < kWin; ++d) {
                const float w = c_win[d], x = s_x[py][cx + d], y = s_y[py][cx + d];
                sx += w * x; sy += w * y; sxx += w * x * x; syy += w * y * y; sxy += w * x * y;
            }
            s_h[0][py][cx] = sx; s_h[1][py][cx] = sy; s_h[2][py][cx] = sxx; s_h[3][py][cx] = syy; s_h[4][py][cx] = sxy;
        }
        __syncthreads();
        // columns
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int d = 0; d < kWin; ++d) {
            const float w = c_win[d];
            mu1 += w * s_h[0][ty + d][tx]; mu2 += w * s_h[1][ty + d][tx];
            e11 += w * s_h[2][ty + d][tx]; e22 += w * s_h[3][ty + d][tx]; e12 += w * s_h[4][ty + d][tx];
        }
        const int qx = ox + tx, qy = oy + ty;
        if (qx < Wo && qy < Ho) {
            const float var1 = e11 - mu1 * mu1, var2 = e22 - mu2 * mu2, cov = e12 - mu1 * mu2;
            const float a1 = 2.f * mu1 * mu2 + C1, b1 = mu1 * mu1 + mu2 * mu2 + C1;
            const float a2 = 2.f * cov + C2, b2 = var1 + var2 + C2;
            const float ib1 = 1.f / b1, ib2 = 1.f / b2;
            const float lum = a1 * ib1, cs = a2 * ib2;
            acc += lum * cs;
            // partial derivatives of S = lum * cs w.r.t. mu1 (holding var1, cov), var1, cov
            const float dS_dmu1 = cs * (2.f * mu2 * ib1 - a1 * ib1 * ib1 * 2.f * mu1);
            const float dS_dvar1 = -lum * a2 * ib2 * ib2;
            const float dS_dcov = lum * 2.f * ib2;
            const size_t q = (size_t)qy * Wo + qx;
            float* m = maps + (size_t)k * 3 * n_out;
            m[q] = dS_dmu1 - 2.f * mu1 * dS_dvar1 - mu2 * dS_dcov;       // A
            m[n_out + q] = dS_dvar1;                                     // B
            m[2 * n_out + q] = dS_dcov;                                  // C
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) s_red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) atomicAdd(ssim_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// v_pred[H,W,3] = scale * d(sum of the SSIM map)/d pred
template <bool COMPOSITE>
__global__ void __launch_bounds__(256)
ssim_bwd_kernel(int H, int W, int channels, const float* __restrict__ pred, const float* __restrict__ alpha,
                const float* __restrict__ bg, const float* __restrict__ gt, const float* __restrict__ maps,
                float scale, float* __restrict__ v_pred) {
    __shared__ float s_m[3][kSPatch][kSPatch + 1];
    __shared__ float s_h[3][kSPatch][kSTile + 1];
    const int Ho = H - kHalo, Wo = W - kHalo;
    const int ox = blockIdx.x * kSTile, oy = blockIdx.y * kSTile;        // origin in the image
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const size_t n_out = (size_t)Ho * Wo;
    for (int k = 0; k < 3; ++k) {
        const float* m = maps + (size_t)k * 3 * n_out;
        __syncthreads();
        // map patch rows oy-10 .. oy+15, cols ox-10 .. ox+15 (zero outside the map)
        for (int i = tid; i < kSPatch * kSPatch; i += 256) {
            const int py = i / kSPatch, px = i - py * kSPatch;
            const int qy = oy + py - kHalo, qx = ox + px - kHalo;
            float a = 0.f, b = 0.f, c = 0.f;
            if (qy >= 0 && qy < Ho && qx >= 0 && qx < Wo) {
                const size_t q = (size_t)qy * Wo + qx;
                a = m[q]; b = m[n_out + q]; c = m[2 * n_out + q];
            }
            s_m[0][py][px] = a; s_m[1][py][px] = b; s_m[2][py][px] = c;
        }
        __syncthreads();
        // rows: out(px) = sum_d w[d] M(px - d)  -> patch column (cx + 10 - d)
        for (int i = tid; i < kSPatch * kSTile; i += 256) {
            const int py = i / kSTile, cx = i - py * kSTile;
            float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
            for (int d = 0; d < kWin; ++d) {
                const float w = c_win[d];
                sa += w * s_m[0][py][cx + kHalo - d]; sb += w * s_m[1][py][cx + kHalo - d];
                sc += w * s_m[2][py][cx + kHalo - d];
            }
            s_h[0][py][cx] = sa; s_h[1][py][cx] = sb; s_h[2][py][cx] = sc;
        }
        __syncthreads();
        float ga = 0.f, gb = 0.f, gc = 0.f;
#pragma unroll
        for (int d = 0; d < kWin; ++d) {
            const float w = c_win[d];
            ga += w * s_h[0][ty + kHalo - d][tx]; gb += w * s_h[1][ty + kHalo - d][tx];
            gc += w * s_h[2][ty + kHalo - d][tx];
        }
        const int ix = ox + tx, iy = oy + ty;
        if (ix < W && iy < H) {
            const float x = pred_at<COMPOSITE>(pred, alpha, bg, channels, W, iy, ix, k);
            const float y = gt[((size_t)iy * W + ix) * 3 + k];
            v_pred[((size_t)iy * W + ix) * 3 + k] = scale * (ga + 2.f * x * gb + y * gc);
        }
    }
}

}  // namespace qed

using namespace qed;

extern "C" int64_t qed_ssim_maps_floats(int32_t height, int32_t width) {
    if (height <= kHalo || width <= kHalo) return QED_E_INVALID_ARG;
    return 9ll * (height - kHalo) * (width - kHalo);
}

extern "C" int qed_ssim_fwd(int32_t height, int32_t width, int32_t channels, const float* pred, const float* alpha,
                            const float* background, const float* gt_rgb, float* maps, float* ssim_sum,
                            void* stream) {
    QED_REQUIRE(height > kHalo && width > kHalo, "image smaller than the 11 x 11 SSIM window");
    QED_REQUIRE(pred && gt_rgb && maps && ssim_sum, "null buffers");
    QED_REQUIRE(alpha == nullptr || (background && (channels == 3 || channels == 4)), "composite mode needs a background");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ssim_sum, 0, sizeof(float), st) != hipSuccess) { set_error("qed_ssim_fwd: memset failed"); return QED_E_LAUNCH; }
    const dim3 grid((width - kHalo + kSTile - 1) / kSTile, (height - kHalo + kSTile - 1) / kSTile);
    if (alpha != nullptr)
        hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid, dim3(256), 0, st, height, width, channels, pred, alpha, background,
                           gt_rgb, maps, ssim_sum);
    else
        hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(256), 0, st, height, width, 3, pred, alpha, background,
                           gt_rgb, maps, ssim_sum);
    return check_launch("qed_ssim_fwd");
}

extern "C" int qed_ssim_bwd(int32_t height, int32_t width, int32_t channels, const float* pred, const float* alpha,
                            const float* background, const float* gt_rgb, const float* maps, float scale,
                            float* v_pred, void* stream) {
    QED_REQUIRE(height > kHalo && width > kHalo, "image smaller than the 11 x 11 SSIM window");
    QED_REQUIRE(pred && gt_rgb && maps && v_pred, "null buffers");
    QED_REQUIRE(alpha == nullptr || (background && (channels == 3 || channels == 4)), "composite mode needs a background");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((width + kSTile - 1) / kSTile, (height + kSTile - 1) / kSTile);
    if (alpha != nullptr)
        hipLaunchKernelGGL(ssim_bwd_kernel<true>, grid, dim3(256), 0, st, height, width, channels, pred, alpha, background,
                           gt_rgb, maps, scale, v_pred);
    else
        hipLaunchKernelGGL(ssim_bwd_kernel<false>, grid, dim3(256), 0, st, height, width, 3, pred, alpha, background,
                           gt_rgb, maps, scale, v_pred);
    return check_launch("qed_ssim_bwd");
}
