// K1 + K2 fused: projection + spherical-harmonics colour, forward and backward.
//
// Replaces the projection / SH part of gsplat.rendering.rasterization as called at
// /root/reference/qed_splatter/model.py:267-288 (math: SURVEY.md Appendix A.1-A.3, A.8) and,
// with the QED_F_* activation flags, the eager torch ops around it (model.py:241, 264, 269-271).
//
// HBM-bound: one thread per Gaussian streams 44 B of geometry + 192 B of SH coefficients and
// writes the 48-byte packed record the compositing kernels gather from, plus the separate
// info[...] arrays the reference reads (model.py:289-292).
#include "qed_common.h"

namespace qed {

struct Cam {
    float R[9];
    float t[3];
    float fx, fy, cx, cy;
    float campos[3];
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ viewmats, const float* __restrict__ Ks, int c) {
    Cam cam;
    const float* V = viewmats + 16 * c;
    const float* K = Ks + 9 * c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) cam.R[3 * i + j] = V[4 * i + j];
        cam.t[i] = V[4 * i + 3];
    }
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
    // camera position = -R^T t  (translation of the inverse view matrix)
#pragma unroll
    for (int j = 0; j < 3; ++j)
        cam.campos[j] = -(cam.R[0 + j] * cam.t[0] + cam.R[3 + j] * cam.t[1] + cam.R[6 + j] * cam.t[2]);
    return cam;
}

// QED_F_CAMERA_C2W: the same camera from c2w[C,3,4] (OpenGL) and intrinsics[C,4] = (fx, fy, cx, cy) -- get_viewmat
// (model.py:22-38: flip the y/z columns of R, analytic rigid inverse) evaluated by every thread instead of a launch of
// its own; `V_out` / `K_out` (non-NULL in ONE thread per camera) receive the view matrix and K for the later kernels.
__device__ __forceinline__ Cam load_cam_c2w(const float* __restrict__ c2w, const float* __restrict__ intr, int c,
                                            float* __restrict__ V_out, float* __restrict__ K_out) {
    Cam cam;
    const float* m = c2w + 12 * c;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        R[3 * i] = m[4 * i];
        R[3 * i + 1] = -m[4 * i + 1];
        R[3 * i + 2] = -m[4 * i + 2];
        t[i] = m[4 * i + 3];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) cam.R[3 * i + j] = R[3 * j + i];                         // R^T
        cam.t[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);                       // -R^T t
    }
    cam.fx = intr[4 * c]; cam.fy = intr[4 * c + 1]; cam.cx = intr[4 * c + 2]; cam.cy = intr[4 * c + 3];
#pragma unroll
    for (int j = 0; j < 3; ++j)                          // as load_cam derives it from the view matrix (bit for bit)
        cam.campos[j] = -(cam.R[0 + j] * cam.t[0] + cam.R[3 + j] * cam.t[1] + cam.R[6 + j] * cam.t[2]);
    if (V_out != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) V_out[4 * i + j] = cam.R[3 * i + j];
            V_out[4 * i + 3] = cam.t[i];
        }
        V_out[12] = 0.f; V_out[13] = 0.f; V_out[14] = 0.f; V_out[15] = 1.f;
        K_out[0] = cam.fx; K_out[1] = 0.f; K_out[2] = cam.cx;
        K_out[3] = 0.f; K_out[4] = cam.fy; K_out[5] = cam.cy;
        K_out[6] = 0.f; K_out[7] = 0.f; K_out[8] = 1.f;
    }
    return cam;
}

// normalised wxyz quaternion -> rotation matrix (row major)
__device__ __forceinline__ void quat_to_rotmat(float w, float x, float y, float z, float* R) {
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// C = A(3x3) * B(3x3), row major
__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Everything the projection of one Gaussian into one camera produces.
struct Proj {
    bool valid;
    float mx, my, z;        // mean2d, depth
    float A, B, Cc, det;    // blurred 2D covariance and its determinant
    float ca, cb, cc;       // conic
    float comp;             // anti-aliasing compensation
    float radius;
    float x, y, rz;         // camera-space mean
    float tx, ty;           // clamped x, y used by the Jacobian
    bool clamp_x, clamp_y;
    float Tm[6];            // T = J * W (2x3), cov2d = T T^T + eps I
    float W[9];             // W = R_cam * R_q * diag(s)
    float M[9];             // M = R_q * diag(s)
    float Rq[9];
};

__device__ __forceinline__ void project_one(const Cam& cam, const float* mean, const float* qn, const float* s,
                                            int width, int height, float eps2d, float near_plane, float far_plane,
                                            float radius_clip, Proj& p) {
    p.valid = false;
    p.x = cam.R[0] * mean[0] + cam.R[1] * mean[1] + cam.R[2] * mean[2] + cam.t[0];
    p.y = cam.R[3] * mean[0] + cam.R[4] * mean[1] + cam.R[5] * mean[2] + cam.t[1];
    p.z = cam.R[6] * mean[0] + cam.R[7] * mean[1] + cam.R[8] * mean[2] + cam.t[2];
    if (p.z < near_plane || p.z > far_plane) return;

    quat_to_rotmat(qn[0], qn[1], qn[2], qn[3], p.Rq);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) p.M[3 * i + j] = p.Rq[3 * i + j] * s[j];
    mat3_mul(cam.R, p.M, p.W);

    const float rz = 1.f / p.z;
    const float rz2 = rz * rz;
    p.rz = rz;
    const float tan_fovx = 0.5f * width / cam.fx;
    const float tan_fovy = 0.5f * height / cam.fy;
    const float lim_x_pos = (width - cam.cx) / cam.fx + kJacMargin * tan_fovx;
    const float lim_x_neg = cam.cx / cam.fx + kJacMargin * tan_fovx;
    const float lim_y_pos = (height - cam.cy) / cam.fy + kJacMargin * tan_fovy;
    const float lim_y_neg = cam.cy / cam.fy + kJacMargin * tan_fovy;
    const float xr = p.x * rz, yr = p.y * rz;
    p.clamp_x = !(xr <= lim_x_pos && xr >= -lim_x_neg);
    p.clamp_y = !(yr <= lim_y_pos && yr >= -lim_y_neg);
    p.tx = p.z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
    p.ty = p.z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
    const float j00 = cam.fx * rz, j02 = -cam.fx * p.tx * rz2;
    const float j11 = cam.fy * rz, j12 = -cam.fy * p.ty * rz2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        p.Tm[j] = j00 * p.W[j] + j02 * p.W[6 + j];
        p.Tm[3 + j] = j11 * p.W[3 + j] + j12 * p.W[6 + j];
    }
    const float a0 = p.Tm[0] * p.Tm[0] + p.Tm[1] * p.Tm[1] + p.Tm[2] * p.Tm[2];
    const float b0 = p.Tm[0] * p.Tm[3] + p.Tm[1] * p.Tm[4] + p.Tm[2] * p.Tm[5];
    const float c0 = p.Tm[3] * p.Tm[3] + p.Tm[4] * p.Tm[4] + p.Tm[5] * p.Tm[5];
    p.mx = cam.fx * p.x * rz + cam.cx;
    p.my = cam.fy * p.y * rz + cam.cy;

    // det(T T^T) as the sum of the squared 2x2 minors of T (Cauchy-Binet), and det(T T^T + eps I) = det(T T^T) +
    // eps tr(T T^T) + eps^2: sums of non-negative terms.  a0 c0 - b0^2 is the difference of two numbers 1e4 .. 1e8 times its
    // size for an elongated Gaussian (the conic of a 2 500 : 1 needle came out 4e-3 off, and with it every gradient that
    // passes through it: DESIGN.md section 2)
    const float m01 = p.Tm[0] * p.Tm[4] - p.Tm[1] * p.Tm[3];
    const float m02 = p.Tm[0] * p.Tm[5] - p.Tm[2] * p.Tm[3];
    const float m12 = p.Tm[1] * p.Tm[5] - p.Tm[2] * p.Tm[4];
    const float det_orig = m01 * m01 + m02 * m02 + m12 * m12;
    p.A = a0 + eps2d;
    p.B = b0;
    p.Cc = c0 + eps2d;
    p.det = det_orig + eps2d * (a0 + c0) + eps2d * eps2d;
    if (!(p.det > 0.f)) return;
    p.comp = sqrtf(fmaxf(0.f, det_orig / p.det));
    const float rdet = 1.f / p.det;
    p.ca = p.Cc * rdet;
    p.cb = -p.B * rdet;
    p.cc = p.A * rdet;

    const float bh = 0.5f * (p.A + p.Cc);
    const float hd = 0.5f * (p.A - p.Cc);
    const float v1 = bh + sqrtf(fmaxf(0.01f, hd * hd + p.B * p.B));     // bh^2 - det = ((A - C) / 2)^2 + B^2
    p.radius = ceilf(3.f * sqrtf(v1));
    if (p.radius <= radius_clip) return;
    if (p.mx + p.radius <= 0.f || p.mx - p.radius >= (float)width || p.my + p.radius <= 0.f ||
        p.my - p.radius >= (float)height)
        return;
    p.valid = true;
}

// ---- spherical harmonics (standard real SH basis, 3DGS constants) ------------------------------
__device__ __constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

// d(basis)/dx, /dy, /dz
template <int DEG>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float* bx, float* by, float* bz) {
    bx[0] = by[0] = bz[0] = 0.f;
    if constexpr (DEG > 0) {
        bx[1] = 0.f;     by[1] = -SH_C1; bz[1] = 0.f;
        bx[2] = 0.f;     by[2] = 0.f;    bz[2] = SH_C1;
        bx[3] = -SH_C1;  by[3] = 0.f;    bz[3] = 0.f;
    }
    if constexpr (DEG > 1) {
        const float c0 = 1.0925484305920792f, c2 = 0.31539156525252005f, c4 = 0.5462742152960396f;
        bx[4] = c0 * y;         by[4] = c0 * x;         bz[4] = 0.f;
        bx[5] = 0.f;            by[5] = -c0 * z;        bz[5] = -c0 * y;
        bx[6] = -2.f * c2 * x;  by[6] = -2.f * c2 * y;  bz[6] = 4.f * c2 * z;
        bx[7] = -c0 * z;        by[7] = 0.f;            bz[7] = -c0 * x;
        bx[8] = 2.f * c4 * x;   by[8] = -2.f * c4 * y;  bz[8] = 0.f;
        if constexpr (DEG > 2) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float k0 = -0.5900435899266435f, k1 = 2.890611442640554f, k2 = -0.4570457994644658f,
                        k3 = 0.3731763325901154f, k5 = 1.445305721320277f;
            bx[9] = k0 * 6.f * xy;                  by[9] = k0 * 3.f * (xx - yy);            bz[9] = 0.f;
            bx[10] = k1 * yz;                       by[10] = k1 * xz;                        bz[10] = k1 * xy;
            bx[11] = k2 * (-2.f * xy);              by[11] = k2 * (4.f * zz - xx - 3.f * yy); bz[11] = k2 * 8.f * yz;
            bx[12] = k3 * (-6.f * xz);              by[12] = k3 * (-6.f * yz);               bz[12] = k3 * 3.f * (2.f * zz - xx - yy);
            bx[13] = k2 * (4.f * zz - 3.f * xx - yy); by[13] = k2 * (-2.f * xy);             bz[13] = k2 * 8.f * xz;
            bx[14] = k5 * 2.f * xz;                 by[14] = k5 * (-2.f * yz);               bz[14] = k5 * (xx - yy);
            bx[15] = k0 * 3.f * (xx - yy);          by[15] = k0 * (-6.f * xy);               bz[15] = 0.f;
        }
    }
}

// gather one Gaussian's SH coefficients (K x 3) into registers
template <int K>
__device__ __forceinline__ void load_sh(const float* __restrict__ sh0, const float* __restrict__ shN, float* c) {
    c[0] = sh0[0]; c[1] = sh0[1]; c[2] = sh0[2];
    if constexpr (K > 1) {
#pragma unroll
        for (int i = 0; i < 3 * (K - 1); ++i) c[3 + i] = shN[i];
    }
}

// SH colour (+0.5, clamp) that also hands the backward pass what it would otherwise re-derive from all 3 K coefficients
// (96 MB of reads at config B, and the basis-derivative tables beside the projection state: 256 registers, 22 of them
// spilled): J[axis][ch] = sum_k d b_k / d axis * c_k,ch -- the colour's derivative with respect to the unit direction --
// and the clamp mask (bit ch: colour_ch + 0.5 >= 0).  Ten floats per (camera, Gaussian) instead of 48 coefficients; the
// sums run in the order sh_bwd_stream uses.
template <int DEG>
__device__ __forceinline__ void sh_color_jac(const float* __restrict__ sh0, const float* __restrict__ shN,
                                             const float* dir, float* rgb, float* J /*9: [axis][ch]*/, unsigned& mask) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    const float inorm = rsqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const float x = dir[0] * inorm, y = dir[1] * inorm, z = dir[2] * inorm;
    float b[K], bx[K], by[K], bz[K];
    sh_basis<DEG>(x, y, z, b);
    sh_basis_grad<DEG>(x, y, z, bx, by, bz);
    float col[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float* ck = k == 0 ? sh0 : shN + 3 * (k - 1);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float c = ck[ch];
            col[ch] += b[k] * c;
            if (k > 0) { J[ch] += bx[k] * c; J[3 + ch] += by[k] * c; J[6 + ch] += bz[k] * c; }
        }
    }
    mask = 0u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (col[ch] + 0.5f >= 0.f) mask |= 1u << ch;
        rgb[ch] = fmaxf(col[ch] + 0.5f, 0.f);
    }
}

__device__ __forceinline__ void sh_color_jac_dyn(int deg, const float* sh0, const float* shN, const float* dir, float* rgb,
                                                 float* J, unsigned& mask) {
    switch (deg) {
        case 0: sh_color_jac<0>(sh0, shN, dir, rgb, J, mask); break;
        case 1: sh_color_jac<1>(sh0, shN, dir, rgb, J, mask); break;
        case 2: sh_color_jac<2>(sh0, shN, dir, rgb, J, mask); break;
        default: sh_color_jac<3>(sh0, shN, dir, rgb, J, mask); break;
    }
}

// ================================================================================================
// forward
// ================================================================================================
// ONE_CAM (C == 1, the training case): the camera index is the constant 0 instead of a per-thread 64-bit
// division, so the view matrix / intrinsics are fetched once per wave through the scalar cache
template <bool ONE_CAM>
__global__ void __launch_bounds__(256)
project_fwd_kernel(int N, int C, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ opacities,
                   const float* __restrict__ sh0, int sh0_stride, const float* __restrict__ shN, int shN_stride,
                   int sh_degree, const float* __restrict__ viewmats, const float* __restrict__ Ks, int width,
                   int height, int tile_w, int tile_h, float eps2d, float near_plane, float far_plane,
                   float radius_clip, unsigned flags, int* __restrict__ radii, float* __restrict__ means2d,
                   float* __restrict__ depths, float* __restrict__ conics, float* __restrict__ opac_out,
                   float* __restrict__ colors_out, float4* __restrict__ splats, int* __restrict__ tiles_per_gauss,
                   int* __restrict__ block_sums, float* __restrict__ viewmats_out, float* __restrict__ Ks_out,
                   float* __restrict__ sh_jac, unsigned long long* __restrict__ tile_masks) {
    // slot = c * N + n ; 256 consecutive slots per block (block_sums granularity)
    const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)C * N;
    int ntiles = 0;
    // what the exact per-tile test below needs of this thread's Gaussian (tile_masks)
    float4 m_c0 = make_float4(0.f, 0.f, 1.f, 0.f);
    float m_cc = 1.f, m_tau = 0.f;
    unsigned m_rect = 0;
    if (slot < total) {
        const int c = ONE_CAM ? 0 : (int)(slot / N);
        const int n = (int)(slot - (long long)c * N);
        const Cam cam = (flags & QED_F_CAMERA_C2W)
                            ? load_cam_c2w(viewmats, Ks, c, n == 0 ? viewmats_out + 16 * c : nullptr, Ks_out + 9 * c)
                            : load_cam(viewmats, Ks, c);
        float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
        float q[4] = {quats[4 * n], quats[4 * n + 1], quats[4 * n + 2], quats[4 * n + 3]};
        float s[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
        float op_raw = opacities[n];
        // all eleven parameter floats are requested together and have arrived here: left to itself the compiler
        // sinks each load into the block that first needs it (mean | near-plane test | quats, scales | visibility
        // | opacity), four dependent round trips per Gaussian where one will do
        asm volatile("" : "+v"(mean[0]), "+v"(mean[1]), "+v"(mean[2]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]),
                          "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(op_raw));
        const float qin = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        q[0] *= qin; q[1] *= qin; q[2] *= qin; q[3] *= qin;
        if (flags & QED_F_LOG_SCALES) { s[0] = __expf(s[0]); s[1] = __expf(s[1]); s[2] = __expf(s[2]); }
        Proj p;
        project_one(cam, mean, q, s, width, height, eps2d, near_plane, far_plane, radius_clip, p);

        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
        int rad = 0;
        float mx = 0.f, my = 0.f, z = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f;
        float rgb[3] = {0.f, 0.f, 0.f};
        if (p.valid) {
            rad = (int)p.radius;
            mx = p.mx; my = p.my; z = p.z; ca = p.ca; cb = p.cb; cc = p.cc;
            op = op_raw;
            if (flags & QED_F_LOGIT_OPAC) op = sigmoidf_dev(op);
            if (flags & QED_F_ANTIALIASED) op *= p.comp;
            if (sh_degree >= 0) {
                const float dir[3] = {mean[0] - cam.campos[0], mean[1] - cam.campos[1], mean[2] - cam.campos[2]};
                // ONE evaluation whether or not the hand-over is wanted: a render with gradients and one without are
                // the same image bit for bit (planes of C N floats: every store of a wave is 256 contiguous bytes,
                // and so is every load of the backward pass)
                float J[9];
                unsigned mask;
                sh_color_jac_dyn(sh_degree, sh0 + (size_t)n * sh0_stride, shN + (size_t)n * shN_stride, dir, rgb, J, mask);
                if (sh_jac != nullptr) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) sh_jac[(size_t)i * total + slot] = J[i];
                    sh_jac[(size_t)9 * total + slot] = __uint_as_float(mask);
                }
            } else {
                const float* cptr = sh0 + (size_t)n * sh0_stride;
                rgb[0] = cptr[0]; rgb[1] = cptr[1]; rgb[2] = cptr[2];
                if (flags & QED_F_SIGMOID_COLORS) {
                    rgb[0] = sigmoidf_dev(rgb[0]); rgb[1] = sigmoidf_dev(rgb[1]); rgb[2] = sigmoidf_dev(rgb[2]);
                }
            }
            // tau = ln(255 o): alpha >= 1/255  <=>  sigma <= tau (used by the compositing kernels' quadrant culling)
            const float tau = logf(255.f * op);
            int x0, y0, x1, y1;
            if (flags & QED_F_TIGHT_TILES) tight_tile_rect(mx, my, p.radius, ca, cb, cc, tau, tile_w, tile_h, x0, y0, x1, y1);
            else tile_rect(mx, my, p.radius, tile_w, tile_h, x0, y0, x1, y1);
            ntiles = (x1 - x0) * (y1 - y0);
            m_c0 = make_float4(mx, my, ca, cb); m_cc = cc; m_tau = tau; m_rect = pack_tile_rect(x0, y0, x1);
            r0 = make_float4(mx, my, ca, cb);
            r1 = make_float4(cc, op, rgb[0], rgb[1]);
            r2 = make_float4(rgb[2], (flags & QED_F_DEPTH_CHANNEL) ? z : 0.f, tau,
                             __uint_as_float(pack_tile_rect(x0, y0, x1)));
        }
        radii[slot] = rad;
        means2d[2 * slot] = mx; means2d[2 * slot + 1] = my;
        depths[slot] = z;
        conics[3 * slot] = ca; conics[3 * slot + 1] = cb; conics[3 * slot + 2] = cc;
        opac_out[slot] = op;
        colors_out[3 * slot] = rgb[0]; colors_out[3 * slot + 1] = rgb[1]; colors_out[3 * slot + 2] = rgb[2];
        splats[3 * slot] = r0; splats[3 * slot + 1] = r1; splats[3 * slot + 2] = r2;
    }
    // Exact tile lists (tile_masks != NULL, with QED_F_TIGHT_TILES): of the rectangle counted above only the tiles in which
    // SOME pixel can reach alpha >= 1/255 are listed -- the compositing kernels' own rectangle test (qed_common.h), applied
    // to whole tiles: a tile it drops is one all four of whose quadrants those kernels would have culled after staging
    // the entry, so images and gradients do not change and the list loses another ~18 % (config B: 3.35 M -> 2.75 M).
    // A thread per Gaussian would loop over its own rectangle (the largest of a wave's 64 is ~40 tiles, the mean 7), so
    // the wave tests all its candidates together, one per lane and trip (candidate j of the wave's concatenated
    // rectangles), and a surviving tile sets its bit in the Gaussian's 64-bit mask, which the emit pass expands -- count
    // and emission cannot disagree.
    // Rectangles of more than 64 tiles keep every tile (mask ~0: 1 % of the entries at config B).
    if (tile_masks != nullptr) {
        __shared__ __attribute__((aligned(16))) float4 s_c0[4][64];      // (x, y, conic b, tau)
        __shared__ __attribute__((aligned(16))) float4 s_c1[4][64];      // (a / 2, c / 2, 1 / a, 1 / c)
        __shared__ unsigned s_rect[4][64];
        __shared__ int s_pref[4][65];
        __shared__ int s_nz[4][64];                                      // the lanes that have candidates, in order
        __shared__ unsigned long long s_start[4][64];                    // bit j: a Gaussian's candidates start at j
        __shared__ unsigned long long s_mask[4][64];
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const int cand = ntiles <= 64 ? ntiles : 0;
        int x = cand;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        const int pref = x - cand;
        const int wtot = __shfl(x, 63, 64);
        const unsigned long long nz = __ballot(cand > 0);
        s_pref[wid][lane] = pref;
        s_c0[wid][lane] = make_float4(m_c0.x, m_c0.y, m_c0.w, m_tau);
        s_c1[wid][lane] = make_float4(0.5f * m_c0.z, 0.5f * m_cc, __builtin_amdgcn_rcpf(m_c0.z), __builtin_amdgcn_rcpf(m_cc));
        s_rect[wid][lane] = m_rect;
        s_mask[wid][lane] = 0ull;
        s_start[wid][lane] = 0ull;
        if (cand > 0) s_nz[wid][__popcll(nz & ((1ull << lane) - 1ull))] = lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (cand > 0) atomicOr(&s_start[wid][pref >> 6], 1ull << (pref & 63));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Candidate j = 64 trip + lane belongs to the Gaussian whose range started last at or before j: the number of
        // start bits up to j (one LDS word per trip and a popcount) indexes the list of lanes that have candidates -- two
        // dependent LDS reads per trip where a binary search over the prefix sums takes six.
        int before = 0;                                           // start bits of the earlier trips (wave-uniform)
        for (int j0 = 0; j0 < wtot; j0 += 64) {
            const unsigned long long word = s_start[wid][j0 >> 6];
            const int j = j0 + lane;
            const int k = before + __popcll(word & ((2ull << lane) - 1ull)) - 1;
            before += __popcll(word);
            if (j >= wtot) continue;
            const int g = s_nz[wid][k];
            const int local = j - s_pref[wid][g];
            const float4 c0 = s_c0[wid][g], c1 = s_c1[wid][g];
            const unsigned r = s_rect[wid][g];
            const int w = (int)(r >> 22);                         // 1 .. 64
            const int row = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)w));    // local < 64: exact
            const int tx = (int)(r & 2047u) + local - row * w, ty = (int)((r >> 11) & 2047u) + row;
            // (cull_setup's values from the halves and reciprocals formed once per Gaussian)
            CullGauss cg;
            cg.b = c0.z; cg.ha = c1.x; cg.hc = c1.y; cg.ia = c1.z; cg.ic = c1.w;
            cg.X0 = (float)(tx * QED_TILE) + 0.5f - c0.x; cg.Y0 = (float)(ty * QED_TILE) + 0.5f - c0.y;
            const float ax = fmaxf(fabsf(cg.X0), fabsf(cg.X0 + 15.f)), ay = fmaxf(fabsf(cg.Y0), fabsf(cg.Y0 + 15.f));
            cg.thr = c0.w + 1e-3f + 8e-6f * (2.f * (cg.ha * ax * ax + cg.hc * ay * ay) + fabsf(cg.b) * ax * ay);
            if (tile_may_touch(cg)) atomicOr(&s_mask[wid][g], 1ull << local);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (slot < total) {
            unsigned long long mask = ~0ull;
            if (cand > 0) { mask = s_mask[wid][lane]; ntiles = __popcll(mask); }
            // one 16-byte descriptor per slot -- count, rectangle, mask -- is all the emit pass reads of a Gaussian: one
            // gather where count, record slot 11 and mask were three (they are random accesses when the slots are
            // emitted in depth order: config D's binning 1 047 -> 864 us)
            reinterpret_cast<uint4*>(tile_masks)[slot] = make_uint4((unsigned)ntiles, m_rect, (unsigned)mask, (unsigned)(mask >> 32));
        }
    }
    if (slot < total) tiles_per_gauss[slot] = ntiles;
    // block sum of tile counts (input of the intersection scan)
    __shared__ int wsum[4];
    int v = ntiles;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ================================================================================================
// backward
// ================================================================================================
template <int DEG>
__device__ __forceinline__ void sh_bwd(const float* __restrict__ sh0, const float* __restrict__ shN,
                                       const float* dir, const float* v_rgb_in, float* v_coef /*3K, +=*/,
                                       float* v_dir /*3, +=*/) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    const float n2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
    const float inorm = rsqrtf(n2);
    const float x = dir[0] * inorm, y = dir[1] * inorm, z = dir[2] * inorm;
    float b[K];
    sh_basis<DEG>(x, y, z, b);
    float c[3 * K];
    load_sh<K>(sh0, shN, c);
    float col[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        col[0] += b[k] * c[3 * k]; col[1] += b[k] * c[3 * k + 1]; col[2] += b[k] * c[3 * k + 2];
    }
    // clamp_min(colour + 0.5, 0): gradient passes where colour + 0.5 >= 0
    float v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = (col[ch] + 0.5f >= 0.f) ? v_rgb_in[ch] : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v_coef[3 * k] += b[k] * v[0]; v_coef[3 * k + 1] += b[k] * v[1]; v_coef[3 * k + 2] += b[k] * v[2];
    }
    if constexpr (DEG > 0) {
        float bx[K], by[K], bz[K];
        sh_basis_grad<DEG>(x, y, z, bx, by, bz);
        float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const float w = c[3 * k] * v[0] + c[3 * k + 1] * v[1] + c[3 * k + 2] * v[2];
            vx += bx[k] * w; vy += by[k] * w; vz += bz[k] * w;
        }
        // through d = dir / |dir|
        const float dot = vx * x + vy * y + vz * z;
        v_dir[0] += (vx - dot * x) * inorm;
        v_dir[1] += (vy - dot * y) * inorm;
        v_dir[2] += (vz - dot * z) * inorm;
    }
}

// The SH backward from the forward pass's hand-over (sh_color_jac): no coefficient is read.  STREAM: the coefficient
// gradients go straight to memory (one camera); otherwise they are accumulated in v_coef.
template <int DEG, bool STREAM>
__device__ __forceinline__ void sh_bwd_jac(const float* __restrict__ jac /* this slot's first plane entry */, size_t plane,
                                           const float* dir, const float* v_rgb_in, float* __restrict__ o0,
                                           float* __restrict__ oN, float* v_coef, float* v_dir /*3, +=*/, bool compact,
                                           float* stage_b = nullptr /*LDS: this lane's basis row*/,
                                           float* stage_v = nullptr) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    float J[9];
#pragma unroll
    for (int i = 0; i < (DEG > 0 ? 9 : 0); ++i) J[i] = jac[(size_t)i * plane];
    const unsigned mask = __float_as_uint(jac[(size_t)9 * plane]);
    const float n2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
    const float inorm = rsqrtf(n2);
    const float x = dir[0] * inorm, y = dir[1] * inorm, z = dir[2] * inorm;
    float v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = (mask >> ch & 1u) ? v_rgb_in[ch] : 0.f;
    if (STREAM && compact) {
        o0[0] = v[0]; o0[1] = v[1]; o0[2] = v[2];
    } else {
        float b[K];
        sh_basis<DEG>(x, y, z, b);
        if constexpr (STREAM) {
            o0[0] = b[0] * v[0]; o0[1] = b[0] * v[1]; o0[2] = b[0] * v[2];
            if (stage_b != nullptr) {
                // the 3 (K - 1) higher-order gradients leave through LDS, a wave's rows at a time (sh_grad_store_staged)
                constexpr int KS = (K + 3) & ~3;
#pragma unroll
                for (int k4 = 0; k4 < KS / 4; ++k4)
                    reinterpret_cast<float4*>(stage_b)[k4] =
                        make_float4(b[4 * k4], 4 * k4 + 1 < K ? b[4 * k4 + 1] : 0.f, 4 * k4 + 2 < K ? b[4 * k4 + 2] : 0.f,
                                    4 * k4 + 3 < K ? b[4 * k4 + 3] : 0.f);
                *reinterpret_cast<float4*>(stage_v) = make_float4(v[0], v[1], v[2], 0.f);
            } else {
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    oN[3 * (k - 1)] = b[k] * v[0]; oN[3 * (k - 1) + 1] = b[k] * v[1]; oN[3 * (k - 1) + 2] = b[k] * v[2];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                v_coef[3 * k] += b[k] * v[0]; v_coef[3 * k + 1] += b[k] * v[1]; v_coef[3 * k + 2] += b[k] * v[2];
            }
        }
    }
    if constexpr (DEG > 0) {
        const float vx = J[0] * v[0] + J[1] * v[1] + J[2] * v[2];
        const float vy = J[3] * v[0] + J[4] * v[1] + J[5] * v[2];
        const float vz = J[6] * v[0] + J[7] * v[1] + J[8] * v[2];
        const float dot = vx * x + vy * y + vz * z;
        v_dir[0] += (vx - dot * x) * inorm;
        v_dir[1] += (vy - dot * y) * inorm;
        v_dir[2] += (vz - dot * z) * inorm;
    }
}

// Single-camera variant of sh_bwd: the coefficient gradients are final after one camera, so they go
// straight to memory instead of through a 48-register accumulator that stays live across the whole kernel
// (with it the degree-3 kernel needed 256 VGPRs + 82 AGPRs = one wave per SIMD).
template <int DEG>
__device__ __forceinline__ void sh_bwd_stream(const float* __restrict__ sh0, const float* __restrict__ shN,
                                              const float* dir, const float* v_rgb_in, float* __restrict__ o0,
                                              float* __restrict__ oN, float* v_dir /*3, +=*/, bool compact) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    const float n2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
    const float inorm = rsqrtf(n2);
    const float x = dir[0] * inorm, y = dir[1] * inorm, z = dir[2] * inorm;
    float b[K];
    sh_basis<DEG>(x, y, z, b);
    // ONE pass over the 3 K coefficients, each consumed as it arrives (holding all 48 beside the basis and its
    // three derivative tables spilled 25 VGPRs at the 256-register limit): the colour needs sum_k b_k c_k, the
    // direction gradient sum_k db_k (c_k . v) = sum_ch v_ch (sum_k db_k c_k,ch) -- nine sums that do not need v
    float col[3] = {0.f, 0.f, 0.f};
    float S[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};      // [axis][channel]
    {
        float bx[K], by[K], bz[K];
        sh_basis_grad<DEG>(x, y, z, bx, by, bz);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float* ck = k == 0 ? sh0 : shN + 3 * (k - 1);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float c = ck[ch];
                col[ch] += b[k] * c;
                if (k > 0) { S[0][ch] += bx[k] * c; S[1][ch] += by[k] * c; S[2][ch] += bz[k] * c; }
            }
        }
    }
    float v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) v[ch] = (col[ch] + 0.5f >= 0.f) ? v_rgb_in[ch] : 0.f;
    if (compact) {
        // QED_F_SH_GRAD_COMPACT: only the clamp-masked colour gradient (3 floats) leaves the kernel; the 48
        // coefficient gradients b_k(dir) v are rebuilt from it by qed_sh_grad_from_views / qed_adam_step_sh
        o0[0] = v[0]; o0[1] = v[1]; o0[2] = v[2];
    } else {
        o0[0] = b[0] * v[0]; o0[1] = b[0] * v[1]; o0[2] = b[0] * v[2];
#pragma unroll
        for (int k = 1; k < K; ++k) {
            oN[3 * (k - 1)] = b[k] * v[0]; oN[3 * (k - 1) + 1] = b[k] * v[1]; oN[3 * (k - 1) + 2] = b[k] * v[2];
        }
    }
    if constexpr (DEG > 0) {
        const float vx = S[0][0] * v[0] + S[0][1] * v[1] + S[0][2] * v[2];
        const float vy = S[1][0] * v[0] + S[1][1] * v[1] + S[1][2] * v[2];
        const float vz = S[2][0] * v[0] + S[2][1] * v[1] + S[2][2] * v[2];
        const float dot = vx * x + vy * y + vz * z;
        v_dir[0] += (vx - dot * x) * inorm;
        v_dir[1] += (vy - dot * y) * inorm;
        v_dir[2] += (vz - dot * z) * inorm;
    }
}

// The higher-order coefficient gradients b_k(dir) v of a wave's 64 Gaussians, written as the wave's 64 x 3 (K - 1) floats
// in address order (rows `stride` floats apart): every lane storing its own row is 45 store instructions of 64 addresses
// 180 B apart, 16 partial writes to every cache line spread over as many instructions; through LDS each instruction
// covers 256 contiguous bytes.  Lanes whose Gaussian is culled (or beyond N) stage v = 0 and their rows come out as the
// zeros the optimiser expects.  A wave reads only what its own lanes staged.
template <int K>
__device__ __forceinline__ void sh_grad_store_staged(const float* __restrict__ stage /* this wave's 64 (KS + 4) floats */,
                                                     int lane, float* __restrict__ out /* row of the wave's first Gaussian */,
                                                     int stride, int n_rows) {
    constexpr int KS = (K + 3) & ~3, FP = 3 * (K - 1);
    const float* sv = stage + 64 * KS;
    int g = lane / FP, j = lane - g * FP;                        // element `lane` of the wave's 64 FP floats
#pragma unroll 5
    for (int it = 0; it < FP; ++it) {
        const int k = (j * 171) >> 9;                            // j / 3 (j < 384)
        const int ch = j - 3 * k;
        if (g < n_rows) out[(size_t)g * stride + j] = stage[g * KS + k + 1] * sv[4 * g + ch];
        g += 64 / FP; j += 64 % FP;
        if (j >= FP) { j -= FP; ++g; }
    }
}

// JAC: the SH part works from qed_project_fwd's sh_jac hand-over (sh_bwd_jac) instead of the coefficients: no basis-
// derivative tables beside the projection state (183 registers without spills where the degree-3 kernel spills 22 of 256).
template <int DEG, bool ONE_CAM, bool JAC = false>
#ifndef QED_PBWD_WAVES
#define QED_PBWD_WAVES 2
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QED_PBWD_WAVES, QED_PBWD_WAVES)))
project_bwd_kernel(int N, int C, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ opacities,
                   const float* __restrict__ sh0, int sh0_stride, const float* __restrict__ shN, int shN_stride,
                   const float* __restrict__ viewmats, const float* __restrict__ Ks, int width, int height,
                   float eps2d, unsigned flags, const int* __restrict__ radii, const float4* __restrict__ vsplat,
                   float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
                   float* __restrict__ v_opacities, float* __restrict__ v_sh0, int v_sh0_stride,
                   float* __restrict__ v_shN, int v_shN_stride, float* __restrict__ v_viewmats,
                   const float* __restrict__ sh_jac) {
    // DEG = -1: colours pass through (sh_degree None)
    constexpr int K = DEG < 0 ? 1 : (DEG + 1) * (DEG + 1);
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool active = n < N;

    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f}, vo = 0.f;
    // ONE_CAM with SH: coefficient gradients are streamed to memory by sh_bwd_stream (no accumulator)
    constexpr bool kStreamSH = ONE_CAM && DEG >= 0;
    float vcoef[kStreamSH ? 1 : 3 * K];
#pragma unroll
    for (int i = 0; i < (kStreamSH ? 1 : 3 * K); ++i) vcoef[i] = 0.f;
    bool sh_written = false;
    // (hand-over kernel, one camera, all coefficient gradients wanted) they leave through LDS: sh_grad_store_staged
    constexpr bool kStageSH = JAC && kStreamSH && K > 1;
    constexpr int kKS = (K + 3) & ~3;
    constexpr int kStageFloats = kStageSH ? 64 * (kKS + 4) : 1;
    __shared__ __attribute__((aligned(16))) float s_stage[4 * kStageFloats];
    float* const stage_wave = s_stage + (threadIdx.x >> 6) * kStageFloats;
    const bool staged = kStageSH && !(flags & QED_F_SH_GRAD_COMPACT);
    float* const stage_b = staged ? stage_wave + (threadIdx.x & 63) * kKS : nullptr;
    float* const stage_v = staged ? stage_wave + 64 * kKS + 4 * (threadIdx.x & 63) : nullptr;

    float mean[3] = {0.f, 0.f, 0.f}, qraw[4] = {1.f, 0.f, 0.f, 0.f}, q[4], sraw[3] = {0.f, 0.f, 0.f}, s[3];
    float qin = 1.f, oraw = 0.f, oact = 0.f;
    if (active) {
        mean[0] = means[3 * n]; mean[1] = means[3 * n + 1]; mean[2] = means[3 * n + 2];
        qraw[0] = quats[4 * n]; qraw[1] = quats[4 * n + 1]; qraw[2] = quats[4 * n + 2]; qraw[3] = quats[4 * n + 3];
        sraw[0] = scales[3 * n]; sraw[1] = scales[3 * n + 1]; sraw[2] = scales[3 * n + 2];
        oraw = opacities[n];
    }
    qin = rsqrtf(qraw[0] * qraw[0] + qraw[1] * qraw[1] + qraw[2] * qraw[2] + qraw[3] * qraw[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = qraw[i] * qin;
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = (flags & QED_F_LOG_SCALES) ? __expf(sraw[i]) : sraw[i];
    oact = (flags & QED_F_LOGIT_OPAC) ? sigmoidf_dev(oraw) : oraw;

    for (int c = 0; c < C; ++c) {
        const Cam cam = load_cam(viewmats, Ks, c);
        const size_t slot = (size_t)c * N + n;
        const bool vis = active && radii[slot] > 0;
        float vR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, vt[3] = {0.f, 0.f, 0.f};
        if constexpr (kStageSH) {
            if (staged && !vis) {                               // culled (or beyond N): a row of zeros
#pragma unroll
                for (int k4 = 0; k4 < kKS / 4; ++k4) reinterpret_cast<float4*>(stage_b)[k4] = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(stage_v) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (vis) {
            Proj p;
            // radius_clip / near / far already decided in the forward pass (radii > 0)
            project_one(cam, mean, q, s, width, height, eps2d, -3.0e38f, 3.0e38f, -1.f, p);
            const float4 g0 = vsplat[4 * slot], g1 = vsplat[4 * slot + 1], g2 = vsplat[4 * slot + 2];
            const float v_mx = g0.x, v_my = g0.y;
            float v_ca = g1.x, v_cb = g1.y, v_cc = g1.z;
            const float v_op = g1.w;
            const float v_rgb[3] = {g2.x, g2.y, g2.z};
            const float v_depth = (flags & QED_F_DEPTH_CHANNEL) ? g2.w : 0.f;

            // ---- opacity (+ compensation) ----
            float v_comp = 0.f;
            if (flags & QED_F_ANTIALIASED) {
                vo += v_op * p.comp;
                v_comp = v_op * oact;
            } else {
                vo += v_op;
            }
            // ---- conic -> blurred covariance -> T:  H = -X G X,  X = conic, G = [[v_ca, v_cb/2],[v_cb/2, v_cc]], v_T = 2 H T,
            // evaluated in the EIGENBASIS of the covariance: X = U diag(1 / l+, 1 / l-) U^T, so H = -U (G' / (l_a l_b)) U^T
            // with G' = U^T G U.  Formed entry by entry in the pixel basis, the share of H along an elongated Gaussian's
            // long axis (~ 1 / l+^2) is what is left of entries 1e8 times larger: `scales` gradients 6e-2 off on
            // 2 500 : 1 needles.  l+ = bh + r and l- = det / l+ with the cancellation-free det.
            const float g01 = 0.5f * v_cb;
            const float bh = 0.5f * (p.A + p.Cc), hd = 0.5f * (p.A - p.Cc);
            const float rr = __builtin_amdgcn_sqrtf(hd * hd + p.B * p.B);              // (1 ulp forms: the bound is 1e-4)
            const float lp = bh + rr, ilp = __builtin_amdgcn_rcpf(lp), lm = p.det * ilp, ilm = __builtin_amdgcn_rcpf(lm);
            // unit eigenvector of l+: (B, l+ - A) or (l+ - C, B), whichever is formed without a cancelling subtraction
            float ux, uy;
            if (hd >= 0.f) { ux = hd + rr; uy = p.B; }          // l+ - C = (A - C) / 2 + r
            else { ux = p.B; uy = rr - hd; }                    // l+ - A = (C - A) / 2 + r
            const float un = ux * ux + uy * uy;
            if (un > 0.f) { const float ir = __builtin_amdgcn_rsqf(un); ux *= ir; uy *= ir; } else { ux = 1.f; uy = 0.f; }
            const float wx = -uy, wy = ux;                      // eigenvector of l-
            const float gux = v_ca * ux + g01 * uy, guy = g01 * ux + v_cc * uy;       // G u
            const float gwx = v_ca * wx + g01 * wy, gwy = g01 * wx + v_cc * wy;       // G w
            float hpp = -(ux * gux + uy * guy) * ilp * ilp;     // H' in the eigenbasis
            float hpm = -(ux * gwx + uy * gwy) * ilp * ilm;
            float hmm = -(wx * gwx + wy * gwy) * ilm * ilm;
            if (flags & QED_F_ANTIALIASED) {
                // compensation = sqrt(max(0, det_orig / det_blur))  (gsplat add_blur_vjp): + v_sqr (om X - eps det(X) I)
                const float det_conic = 1.f / p.det;
                const float v_sqr = v_comp * 0.5f / (p.comp + 1e-6f);
                const float om = 1.f - p.comp * p.comp;
                hpp += v_sqr * (om * ilp - eps2d * det_conic);
                hmm += v_sqr * (om * ilm - eps2d * det_conic);
            }
            // ---- cov2d = T T^T + eps I  ->  v_T = 2 H T = 2 U H' (U^T T) ----
            float vT[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float pu = ux * p.Tm[j] + uy * p.Tm[3 + j], pw = wx * p.Tm[j] + wy * p.Tm[3 + j];
                const float qu = hpp * pu + hpm * pw, qw = hpm * pu + hmm * pw;
                vT[j] = 2.f * (ux * qu + wx * qw);
                vT[3 + j] = 2.f * (uy * qu + wy * qw);
            }
            // ---- T = J W ----
            const float rz = p.rz, rz2 = rz * rz, rz3 = rz2 * rz;
            const float j00 = cam.fx * rz, j02 = -cam.fx * p.tx * rz2;
            const float j11 = cam.fy * rz, j12 = -cam.fy * p.ty * rz2;
            // v_J = v_T W^T (only the 4 non-constant entries)
            const float vJ00 = vT[0] * p.W[0] + vT[1] * p.W[1] + vT[2] * p.W[2];
            const float vJ02 = vT[0] * p.W[6] + vT[1] * p.W[7] + vT[2] * p.W[8];
            const float vJ11 = vT[3] * p.W[3] + vT[4] * p.W[4] + vT[5] * p.W[5];
            const float vJ12 = vT[3] * p.W[6] + vT[4] * p.W[7] + vT[5] * p.W[8];
            // v_W = J^T v_T
            float vW[9];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                vW[j] = j00 * vT[j];
                vW[3 + j] = j11 * vT[3 + j];
                vW[6 + j] = j02 * vT[j] + j12 * vT[3 + j];
            }
            // ---- camera-space mean ----
            float vx = v_mx * cam.fx * rz;
            float vy = v_my * cam.fy * rz;
            float vz = -v_mx * cam.fx * p.x * rz2 - v_my * cam.fy * p.y * rz2 - vJ00 * cam.fx * rz2 -
                       vJ11 * cam.fy * rz2 + v_depth;
            if (!p.clamp_x) {
                vx += -cam.fx * rz2 * vJ02;
                vz += 2.f * cam.fx * p.tx * rz3 * vJ02;
            } else {
                vz += cam.fx * p.tx * rz3 * vJ02;
            }
            if (!p.clamp_y) {
                vy += -cam.fy * rz2 * vJ12;
                vz += 2.f * cam.fy * p.ty * rz3 * vJ12;
            } else {
                vz += cam.fy * p.ty * rz3 * vJ12;
            }
            // world mean: mean_c = R mean + t
            float vmw[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) vmw[j] = cam.R[j] * vx + cam.R[3 + j] * vy + cam.R[6 + j] * vz;
            // ---- W = R_cam M  ->  v_M = R_cam^T v_W ----
            float vM[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    vM[3 * i + j] = cam.R[i] * vW[j] + cam.R[3 + i] * vW[3 + j] + cam.R[6 + i] * vW[6 + j];
            // ---- SH colour ----
            float vdir[3] = {0.f, 0.f, 0.f};
            if constexpr (DEG >= 0) {
                const float dir[3] = {mean[0] - cam.campos[0], mean[1] - cam.campos[1], mean[2] - cam.campos[2]};
                if constexpr (JAC) {
                    const size_t plane = (size_t)C * N;
                    sh_bwd_jac<(DEG < 0 ? 0 : DEG), kStreamSH>(sh_jac + slot, plane, dir, v_rgb,
                                                                v_sh0 + (size_t)n * v_sh0_stride,
                                                                v_shN + (size_t)n * v_shN_stride, vcoef, vdir,
                                                                (flags & QED_F_SH_GRAD_COMPACT) != 0, stage_b, stage_v);
                    sh_written = true;
                } else if constexpr (kStreamSH) {
                    sh_bwd_stream<(DEG < 0 ? 0 : DEG)>(sh0 + (size_t)n * sh0_stride, shN + (size_t)n * shN_stride, dir,
                                                       v_rgb, v_sh0 + (size_t)n * v_sh0_stride,
                                                       v_shN + (size_t)n * v_shN_stride, vdir,
                                                       (flags & QED_F_SH_GRAD_COMPACT) != 0);
                    sh_written = true;
                } else {
                    sh_bwd<(DEG < 0 ? 0 : DEG)>(sh0 + (size_t)n * sh0_stride, shN + (size_t)n * shN_stride, dir, v_rgb,
                                                vcoef, vdir);
                }
            } else {
                if (flags & QED_F_SIGMOID_COLORS) {
                    const float* cptr = sh0 + (size_t)n * sh0_stride;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float sg = sigmoidf_dev(cptr[ch]);
                        vcoef[ch] += v_rgb[ch] * sg * (1.f - sg);
                    }
                } else {
                    vcoef[0] += v_rgb[0]; vcoef[1] += v_rgb[1]; vcoef[2] += v_rgb[2];
                }
            }
            vm[0] += vmw[0] + vdir[0]; vm[1] += vmw[1] + vdir[1]; vm[2] += vmw[2] + vdir[2];
            // ---- M = Rq diag(s) ----
            float vRq[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    vRq[3 * i + j] = vM[3 * i + j] * s[j];
                    vs[j] += vM[3 * i + j] * p.Rq[3 * i + j];
                }
            // ---- Rq(q) with q normalised ----
            const float w = q[0], x = q[1], y = q[2], z = q[3];
            float vqn[4];
            vqn[0] = 2.f * (x * (vRq[7] - vRq[5]) + y * (vRq[2] - vRq[6]) + z * (vRq[3] - vRq[1]));
            vqn[1] = 2.f * (-2.f * x * (vRq[4] + vRq[8]) + y * (vRq[1] + vRq[3]) + z * (vRq[2] + vRq[6]) +
                            w * (vRq[7] - vRq[5]));
            vqn[2] = 2.f * (x * (vRq[1] + vRq[3]) - 2.f * y * (vRq[0] + vRq[8]) + z * (vRq[5] + vRq[7]) +
                            w * (vRq[2] - vRq[6]));
            vqn[3] = 2.f * (x * (vRq[2] + vRq[6]) + y * (vRq[5] + vRq[7]) - 2.f * z * (vRq[0] + vRq[4]) +
                            w * (vRq[3] - vRq[1]));
            vq[0] += vqn[0]; vq[1] += vqn[1]; vq[2] += vqn[2]; vq[3] += vqn[3];

            if (v_viewmats != nullptr) {
                // v_R = v_meanc mean^T + v_W M^T ; v_t = v_meanc ; campos = -R^T t feeds the SH direction
                const float vmc[3] = {vx, vy, vz};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        vR[3 * i + j] = vmc[i] * mean[j] + vW[3 * i] * p.M[3 * j] + vW[3 * i + 1] * p.M[3 * j + 1] +
                                        vW[3 * i + 2] * p.M[3 * j + 2];
                    }
                    vt[i] = vmc[i];
                }
                // dir = mean - campos, campos_j = -sum_i R_ij t_i  ->  v_campos = -v_dir
                // v_R_ij += -v_campos_j * t_i = v_dir_j * t_i ;  v_t_i += sum_j v_dir_j R_ij
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        vR[3 * i + j] += vdir[j] * cam.t[i];
                        vt[i] += vdir[j] * cam.R[3 * i + j];
                    }
                }
            }
        }
        if constexpr (kStageSH) {
            if (staged) {                                       // (uniform over the launch)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int n0 = blockIdx.x * 256 + (threadIdx.x & ~63);
                sh_grad_store_staged<K>(stage_wave, threadIdx.x & 63, v_shN + (size_t)n0 * v_shN_stride, v_shN_stride,
                                        N - n0);
            }
        }
        if (v_viewmats != nullptr) {
            // block reduction -> 12 atomics per block per camera
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                float v = i < 9 ? vR[i] : vt[i - 9];
                v = wave_sum(v);
                if ((threadIdx.x & 63) == 0 && v != 0.f) {
                    const int row = i < 9 ? i / 3 : i - 9, col = i < 9 ? i % 3 : 3;
                    atomicAdd(&v_viewmats[16 * c + 4 * row + col], v);
                }
            }
        }
    }
    if (!active) return;
    // normalisation q = qraw / |qraw|:  v_qraw = (v_q - (v_q . q) q) / |qraw|
    const float dq = vq[0] * q[0] + vq[1] * q[1] + vq[2] * q[2] + vq[3] * q[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) v_quats[4 * n + i] = (vq[i] - dq * q[i]) * qin;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v_means[3 * n + i] = vm[i];
        v_scales[3 * n + i] = (flags & QED_F_LOG_SCALES) ? vs[i] * s[i] : vs[i];
    }
    v_opacities[n] = (flags & QED_F_LOGIT_OPAC) ? vo * oact * (1.f - oact) : vo;
    float* o0 = v_sh0 + (size_t)n * v_sh0_stride;
    if constexpr (kStreamSH) {
        if (!sh_written) {                              // not visible: zero gradient
            o0[0] = 0.f; o0[1] = 0.f; o0[2] = 0.f;
            if (K > 1 && !(flags & QED_F_SH_GRAD_COMPACT) && !staged) {
                float* oN = v_shN + (size_t)n * v_shN_stride;
#pragma unroll
                for (int i = 0; i < 3 * (K - 1); ++i) oN[i] = 0.f;
            }
        }
    } else {
        o0[0] = vcoef[0]; o0[1] = vcoef[1]; o0[2] = vcoef[2];
        if constexpr (K > 1) {
            float* oN = v_shN + (size_t)n * v_shN_stride;
#pragma unroll
            for (int i = 0; i < 3 * (K - 1); ++i) oN[i] = vcoef[3 + i];
        }
    }
}

// ---- SH coefficient gradients from the per-view colour gradients (data-parallel exchange) -----
// d L / d sh[k] = sum over views of b_k(dir_view) * v_view, where v_view is the clamp-masked colour gradient
// that project_bwd wrote with QED_F_SH_GRAD_COMPACT.  Every rank gathers the 3 floats per Gaussian of every
// view instead of all-reducing 48: 42 MB received instead of 206 MB moved per rank at 8 GPUs / 500 k Gaussians.
template <int DEG>
__global__ void __launch_bounds__(256)
sh_grad_from_views_kernel(int N, int n_views, const float* __restrict__ means, const float* __restrict__ viewmats,
                          long long viewmat_stride, const float* __restrict__ v_views, long long view_stride,
                          float scale, float* __restrict__ v_sh0, int v_sh0_stride, float* __restrict__ v_shN,
                          int v_shN_stride) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float acc[3 * K];
#pragma unroll
    for (int i = 0; i < 3 * K; ++i) acc[i] = 0.f;
    for (int c = 0; c < n_views; ++c) {
        const float* vm = viewmats + viewmat_stride * c;
        const float* vv = v_views + view_stride * c + (size_t)n * 3;
        const float v[3] = {vv[0], vv[1], vv[2]};
        if (v[0] == 0.f && v[1] == 0.f && v[2] == 0.f) continue;        // not visible in this view
        float dir[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)                                      // campos = -R^T t, as load_cam
            dir[j] = mean[j] + (vm[0 + j] * vm[3] + vm[4 + j] * vm[7] + vm[8 + j] * vm[11]);
        const float inorm = rsqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        float b[K];
        sh_basis<DEG>(dir[0] * inorm, dir[1] * inorm, dir[2] * inorm, b);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc[3 * k] += b[k] * v[0]; acc[3 * k + 1] += b[k] * v[1]; acc[3 * k + 2] += b[k] * v[2];
        }
    }
    float* o0 = v_sh0 + (size_t)n * v_sh0_stride;
    o0[0] = acc[0] * scale; o0[1] = acc[1] * scale; o0[2] = acc[2] * scale;
    if constexpr (K > 1) {
        float* oN = v_shN + (size_t)n * v_shN_stride;
#pragma unroll
        for (int i = 0; i < 3 * (K - 1); ++i) oN[i] = acc[3 + i] * scale;
    }
}

// ---- a1 + a3 in one launch: get_viewmat (model.py:22-38) and the intrinsics matrix -----------------
// c2w[C,3,4] (OpenGL camera-to-world) -> viewmats[C,4,4]: flip the y/z columns of R, then the analytic
// rigid inverse (R^T, -R^T t); intr[C,4] = (fx, fy, cx, cy) -> Ks[C,3,3].
__global__ void camera_setup_kernel(int C, const float* __restrict__ c2w, const float* __restrict__ intr,
                                    float* __restrict__ viewmats, float* __restrict__ Ks) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* m = c2w + 12 * c;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        R[3 * i] = m[4 * i];
        R[3 * i + 1] = -m[4 * i + 1];
        R[3 * i + 2] = -m[4 * i + 2];
        t[i] = m[4 * i + 3];
    }
    float* V = viewmats + 16 * c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) V[4 * i + j] = R[3 * j + i];                     // R^T
        V[4 * i + 3] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);           // -R^T t
    }
    V[12] = 0.f; V[13] = 0.f; V[14] = 0.f; V[15] = 1.f;
    float* K = Ks + 9 * c;
    K[0] = intr[4 * c]; K[1] = 0.f; K[2] = intr[4 * c + 2];
    K[3] = 0.f; K[4] = intr[4 * c + 1]; K[5] = intr[4 * c + 3];
    K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
}

}  // namespace qed

using namespace qed;

extern "C" int qed_camera_setup(int32_t C, const float* c2w, const float* intrinsics, float* viewmats, float* Ks,
                                void* stream) {
    QED_REQUIRE(C >= 1 && c2w && intrinsics && viewmats && Ks, "bad arguments");
    hipLaunchKernelGGL(camera_setup_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, c2w, intrinsics,
                       viewmats, Ks);
    return check_launch("qed_camera_setup");
}

extern "C" int qed_project_fwd(int32_t N, int32_t C, const float* means, const float* quats, const float* scales,
                               const float* opacities, const float* sh0, int32_t sh0_stride, const float* shN,
                               int32_t shN_stride, int32_t sh_degree, const float* viewmats, const float* Ks,
                               int32_t width, int32_t height, int32_t tile_w, int32_t tile_h, float eps2d,
                               float near_plane, float far_plane, float radius_clip, uint32_t flags, int32_t* radii,
                               float* means2d, float* depths, float* conics, float* opac_out, float* colors_out,
                               float* splats, int32_t* tiles_per_gauss, uint64_t* tile_masks, int32_t* block_sums,
                               float* viewmats_out, float* Ks_out, float* sh_jac, void* stream) {
    QED_REQUIRE(N >= 0 && C >= 1, "N >= 0 and C >= 1 required");
    QED_REQUIRE(sh_degree <= 3, "SH degree > 3 unsupported (reference config uses sh_degree = 3)");
    QED_REQUIRE(width > 0 && height > 0 && tile_w > 0 && tile_h > 0, "bad image / tile extents");
    QED_REQUIRE((long long)tile_w * tile_h < (1ll << 30), "too many tiles");
    if (N == 0) return QED_OK;
    QED_REQUIRE(means && quats && scales && opacities && sh0 && viewmats && Ks, "null input");
    QED_REQUIRE(sh_degree <= 0 || shN, "shN required for sh_degree > 0");
    QED_REQUIRE(radii && means2d && depths && conics && opac_out && colors_out && splats && tiles_per_gauss &&
                    block_sums, "null output");
    QED_REQUIRE(!(flags & QED_F_CAMERA_C2W) || (viewmats_out && Ks_out), "QED_F_CAMERA_C2W needs viewmats_out and Ks_out");
    QED_REQUIRE(tile_masks == nullptr || (flags & QED_F_TIGHT_TILES), "tile_masks goes with QED_F_TIGHT_TILES");
    QED_REQUIRE(((uintptr_t)tile_masks & 15) == 0, "tile_masks must be 16-byte aligned");
    const long long total = (long long)C * N;
    const unsigned grid = (unsigned)((total + 255) / 256);
#define QED_LAUNCH_PF(ONE)                                                                                           \
    hipLaunchKernelGGL(project_fwd_kernel<ONE>, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, C, means, quats,   \
                       scales, opacities, sh0, sh0_stride, shN, shN_stride, sh_degree, viewmats, Ks, width, height,  \
                       tile_w, tile_h, eps2d, near_plane, far_plane, radius_clip, flags, radii, means2d, depths,     \
                       conics, opac_out, colors_out, (float4*)splats, tiles_per_gauss, block_sums, viewmats_out, Ks_out,    \
                       sh_degree >= 0 ? sh_jac : nullptr, (unsigned long long*)tile_masks)
    if (C == 1) QED_LAUNCH_PF(true);
    else QED_LAUNCH_PF(false);
#undef QED_LAUNCH_PF
    return check_launch("qed_project_fwd");
}

extern "C" int qed_project_bwd(int32_t N, int32_t C, const float* means, const float* quats, const float* scales,
                               const float* opacities, const float* sh0, int32_t sh0_stride, const float* shN,
                               int32_t shN_stride, int32_t sh_degree, const float* viewmats, const float* Ks,
                               int32_t width, int32_t height, float eps2d, uint32_t flags, const int32_t* radii,
                               const float* vsplat, float* v_means, float* v_quats, float* v_scales,
                               float* v_opacities, float* v_sh0, int32_t v_sh0_stride, float* v_shN,
                               int32_t v_shN_stride, float* v_viewmats, const float* sh_jac, void* stream) {
    QED_REQUIRE(N >= 0 && C >= 1, "N >= 0 and C >= 1 required");
    QED_REQUIRE(sh_degree <= 3, "SH degree > 3 unsupported");
    if (N == 0) return QED_OK;
    const bool jac = sh_jac != nullptr && sh_degree >= 0;      // (the coefficients are not read then)
    QED_REQUIRE(means && quats && scales && opacities && (sh0 || jac) && viewmats && Ks && radii && vsplat, "null input");
    QED_REQUIRE(v_means && v_quats && v_scales && v_opacities && v_sh0, "null output");
    QED_REQUIRE(sh_degree <= 0 || ((shN || jac) && v_shN), "shN / v_shN required for sh_degree > 0");
    const unsigned grid = (unsigned)((N + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
#define QED_LAUNCH_BWD_(D, ONE, J)                                                                                 \
    hipLaunchKernelGGL((project_bwd_kernel<D, ONE, J>), dim3(grid), dim3(256), 0, st, N, C, means, quats, scales,  \
                       opacities, sh0, sh0_stride, shN, shN_stride, viewmats, Ks, width, height, eps2d, flags,     \
                       radii, (const float4*)vsplat, v_means, v_quats, v_scales, v_opacities, v_sh0, v_sh0_stride, \
                       v_shN, v_shN_stride, v_viewmats, sh_jac)
#define QED_LAUNCH_BWD(D)                                  \
    do {                                                   \
        if (jac && D >= 0) {                               \
            if (C == 1) QED_LAUNCH_BWD_(D, true, (D >= 0)); \
            else QED_LAUNCH_BWD_(D, false, (D >= 0));      \
        } else if (C == 1) QED_LAUNCH_BWD_(D, true, false); \
        else QED_LAUNCH_BWD_(D, false, false);             \
    } while (0)
    switch (sh_degree) {
        case 0: QED_LAUNCH_BWD(0); break;
        case 1: QED_LAUNCH_BWD(1); break;
        case 2: QED_LAUNCH_BWD(2); break;
        case 3: QED_LAUNCH_BWD(3); break;
        default: QED_LAUNCH_BWD(-1); break;
    }
#undef QED_LAUNCH_BWD
#undef QED_LAUNCH_BWD_
    return check_launch("qed_project_bwd");
}

// The message of the data-parallel exchange BEFORE the projection backward has run: the compositing backward's colour
// gradient (vsplat row slots 8..10) under the clamp mask the forward pass kept (sh_jac plane 9) is exactly what
// qed_project_bwd(QED_F_SH_GRAD_COMPACT) writes as v_sh0 -- packed here into [N,3] so that the all-gather of the colour
// gradients can be on the links while the projection backward runs.
__global__ void __launch_bounds__(256)
pack_color_grad_kernel(int total, const float* __restrict__ vsplat, const float* __restrict__ mask_plane,
                       float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float4 r = *reinterpret_cast<const float4*>(vsplat + (size_t)i * QED_VSPLAT_FLOATS + 8);     // v_r, v_g, v_b, v_depth
    const unsigned m = __float_as_uint(mask_plane[i]);
    // (a slot the camera does not see has an all-zero row -- the compositing backward never touched it -- and an
    // unwritten mask word: 0 either way)
    out[3 * (size_t)i] = (m & 1u) ? r.x : 0.f;
    out[3 * (size_t)i + 1] = (m & 2u) ? r.y : 0.f;
    out[3 * (size_t)i + 2] = (m & 4u) ? r.z : 0.f;
}

extern "C" int qed_pack_color_grad(int32_t total, const float* vsplat, const float* sh_jac, float* out, void* stream) {
    QED_REQUIRE(total >= 0, "bad extent");
    if (total == 0) return QED_OK;
    QED_REQUIRE(vsplat && sh_jac && out, "null buffers");
    QED_REQUIRE(((uintptr_t)vsplat & 15) == 0, "vsplat must be 16-byte aligned");
    hipLaunchKernelGGL(pack_color_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total,
                       vsplat, sh_jac + (size_t)9 * total, out);
    return check_launch("qed_pack_color_grad");
}

extern "C" int qed_sh_grad_from_views(int32_t N, int32_t n_views, const float* means, const float* viewmats,
                                      int64_t viewmat_stride, const float* v_views, int64_t view_stride,
                                      int32_t sh_degree, float scale, float* v_sh0, int32_t v_sh0_stride,
                                      float* v_shN, int32_t v_shN_stride, void* stream) {
    QED_REQUIRE(N >= 0 && n_views >= 1 && sh_degree >= 0 && sh_degree <= 3, "bad arguments");
    QED_REQUIRE(viewmat_stride >= 16 && view_stride >= 3ll * N, "view strides too small");
    if (N == 0) return QED_OK;
    QED_REQUIRE(means && viewmats && v_views && v_sh0, "null buffers");
    QED_REQUIRE(sh_degree == 0 || v_shN, "v_shN required for sh_degree > 0");
    const unsigned grid = (unsigned)((N + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
#define QED_LAUNCH_SHG(D)                                                                                        \
    hipLaunchKernelGGL(sh_grad_from_views_kernel<D>, dim3(grid), dim3(256), 0, st, N, n_views, means, viewmats, \
                       (long long)viewmat_stride, v_views, (long long)view_stride, scale, v_sh0, v_sh0_stride,  \
                       v_shN, v_shN_stride)
    switch (sh_degree) {
        case 0: QED_LAUNCH_SHG(0); break;
        case 1: QED_LAUNCH_SHG(1); break;
        case 2: QED_LAUNCH_SHG(2); break;
        default: QED_LAUNCH_SHG(3); break;
    }
#undef QED_LAUNCH_SHG
    return check_launch("qed_sh_grad_from_views");
}
