// densification / culling driven by the side effects of get_outputs
// (self.xys.absgrad, self.radii, self.last_size; reference model.py:249,289-292).
//
// The callbacks live in the reference's parent class (Nerfstudio SplatfactoModel: after_train,
// refinement_after, split_gaussians, dup_gaussians, cull_gaussians, dup_in_all_optim,
// remove_from_all_optim; un-vendored).  oracle/densify_oracle.py restates them; these kernels do the same
// work on the flat parameter / Adam-moment buffers of this package (group order means, scales, quats,
// opacities, features_dc, features_rest) without ever concatenating or index-selecting per group:
//
//   accumulate  per training step: visibility counts, sum of |absgrad| norms, largest screen radius
//   classify    per refinement: split / duplicate / keep decisions of every Gaussian  -> flags
//   scan        ranks of the splits and output slots of the kept rows (3 launches)   -> pos, totals
//   emit        one pass that writes the new parameter and moment buffers in the reference's order
//               [kept old | children of sample 0 | children of sample 1 | ... | duplicates]
//   reset       opacity clamp + zeroed opacity moments
//
// All of it is HBM streaming (59 floats x 3 buffers per Gaussian in, the same out).
#include "qed_common.h"

namespace qed {

constexpr int kFlagSplit = 1, kFlagDup = 2, kFlagKeepOld = 4, kFlagKeepChild = 8, kFlagKeepDup = 16;
constexpr float kSplitShrink = 1.6f;       // size_fac of split_gaussians

__global__ void __launch_bounds__(256)
densify_accumulate_kernel(int N, const float* __restrict__ absgrad, int stride, const int* __restrict__ radii,
                          float inv_max_dim, float* __restrict__ grad_norm, float* __restrict__ vis_counts,
                          float* __restrict__ max_2d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = absgrad[(size_t)i * stride], gy = absgrad[(size_t)i * stride + 1];
    vis_counts[i] += 1.f;
    grad_norm[i] += sqrtf(gx * gx + gy * gy);
    max_2d[i] = fmaxf(max_2d[i], (float)r * inv_max_dim);
}

struct ClassifyArgs {
    int densify;                 // 0: cull only (post-densification)
    float half_max_dim;          // 0.5 * max(H, W)
    float grad_thresh, size_thresh;
    float split_screen;          // < 0: screen-size split test off (step >= stop_screen_size_at)
    float cull_alpha;
    float cull_scale;            // < 0: "too big" culling off (step <= refine_every * reset_alpha_every)
    float cull_screen;           // < 0: screen-size culling off
};

// shrunk log-scale of a split Gaussian, evaluated exactly as the reference does: log(exp(s) / 1.6)
__device__ __forceinline__ float shrink(float s) { return logf(expf(s) / kSplitShrink); }

__device__ __forceinline__ unsigned classify_one(int i, const float* __restrict__ scales,
                                                 const float* __restrict__ opacities,
                                                 const float* __restrict__ grad_norm,
                                                 const float* __restrict__ vis_counts,
                                                 const float* __restrict__ max_2d, const ClassifyArgs& a) {
    const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    const float m2d = max_2d ? max_2d[i] : 0.f;
    bool split = false, dup = false;
    float smax_after = smax;
    if (a.densify) {
        const float avg = (grad_norm[i] / vis_counts[i]) * a.half_max_dim;
        const bool high = avg > a.grad_thresh;
        split = expf(smax) > a.size_thresh;
        if (a.split_screen >= 0.f) split = split || (m2d > a.split_screen);
        split = split && high;
        if (split) smax_after = shrink(smax);                // in-place shrink precedes the duplicate test
        dup = (expf(smax_after) <= a.size_thresh) && high;
    }
    // cull_gaussians on [old | children | duplicates]; children and duplicates share the parent's opacity
    // and (shrunk) scale and have max_2Dsize 0
    const bool low_alpha = sigmoidf_dev(opacities[i]) < a.cull_alpha;
    const bool big_after = a.cull_scale >= 0.f && expf(smax_after) > a.cull_scale;
    const bool big_screen = a.cull_scale >= 0.f && a.cull_screen >= 0.f && max_2d && m2d > a.cull_screen;
    const bool keep_old = !split && !low_alpha && !big_after && !big_screen;
    const bool keep_new = !low_alpha && !big_after;
    unsigned f = 0;
    if (split) f |= kFlagSplit;
    if (dup) f |= kFlagDup;
    if (keep_old) f |= kFlagKeepOld;
    if (split && keep_new) f |= kFlagKeepChild;
    if (dup && keep_new) f |= kFlagKeepDup;
    return f;
}

// The four scanned 0/1 columns: split rank, slot among kept old rows, slot among kept children (per
// sample), slot among kept duplicates.
__device__ __forceinline__ unsigned scan_bit(int c) {
    return c == 0 ? kFlagSplit : c == 1 ? kFlagKeepOld : c == 2 ? kFlagKeepChild : kFlagKeepDup;
}

// scan step 1: decisions + per-workgroup counts of the four columns: block_counts[c][block]
__global__ void __launch_bounds__(256)
densify_classify_kernel(int N, const float* __restrict__ scales, const float* __restrict__ opacities,
                        const float* __restrict__ grad_norm, const float* __restrict__ vis_counts,
                        const float* __restrict__ max_2d, ClassifyArgs a, unsigned char* __restrict__ flags,
                        int* __restrict__ block_counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned f = 0;
    if (i < N) {
        f = classify_one(i, scales, opacities, grad_norm, vis_counts, max_2d, a);
        flags[i] = (unsigned char)f;
    }
    __shared__ int s_cnt[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned long long m = __ballot((f & scan_bit(c)) != 0);
        if (lane == 0) s_cnt[c][wave] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 4)
        block_counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
            s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
}

// scan step 2: one workgroup turns the per-workgroup counts into exclusive bases (in place) and totals
__global__ void __launch_bounds__(1024)
densify_scan_blocks_kernel(int n_blocks, int* __restrict__ block_counts, int* __restrict__ totals) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = 0; c < 4; ++c) {
        int* col = block_counts + (size_t)c * n_blocks;
        if (tid == 0) s_base = 0;
        __syncthreads();
        for (int start = 0; start < n_blocks; start += 1024) {
            const int i = start + tid;
            const int v = i < n_blocks ? col[i] : 0;
            int incl = v;                                    // inclusive scan within the wave
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d, 64);
                if (lane >= d) incl += t;
            }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            int wbase = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int t = s_wave[w];
                if (w < wave) wbase += t;
                tot += t;
            }
            if (i < n_blocks) col[i] = s_base + wbase + incl - v;
            __syncthreads();
            if (tid == 0) s_base += tot;
            __syncthreads();
        }
        if (tid == 0) totals[c] = s_base;
        __syncthreads();
    }
}

// scan step 3: pos[c][i] = base of i's workgroup + rank of i within it
__global__ void __launch_bounds__(256)
densify_positions_kernel(int N, const unsigned char* __restrict__ flags, const int* __restrict__ block_base,
                         int* __restrict__ pos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned f = i < N ? flags[i] : 0u;
    __shared__ int s_cnt[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int excl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned long long m = __ballot((f & scan_bit(c)) != 0);
        excl[c] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[c][wave] = __popcll(m);
    }
    __syncthreads();
    if (i >= N) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int base = block_base[(size_t)c * gridDim.x + blockIdx.x];
        for (int w = 0; w < wave; ++w) base += s_cnt[c][w];
        pos[(size_t)c * N + i] = base + excl[c];
    }
}

struct EmitLayout {
    long long old_begin[7], new_begin[7];   // element offsets of the six groups (+ end) in the flat buffers
    int width[6];                           // floats per Gaussian in each group
    int total_width;
    int k_old, k_child, n_split;            // kept old rows, kept children per sample, all splits
    int n_samples;
};

__device__ __forceinline__ void quat_rotate(const float* __restrict__ q4, float vx, float vy, float vz, float* out) {
    float w = q4[0], x = q4[1], y = q4[2], z = q4[3];
    const float inv = 1.f / sqrtf(w * w + x * x + y * y + z * z);
    w *= inv; x *= inv; y *= inv; z *= inv;
    out[0] = (1.f - 2.f * (y * y + z * z)) * vx + 2.f * (x * y - w * z) * vy + 2.f * (x * z + w * y) * vz;
    out[1] = 2.f * (x * y + w * z) * vx + (1.f - 2.f * (x * x + z * z)) * vy + 2.f * (y * z - w * x) * vz;
    out[2] = 2.f * (x * z - w * y) * vx + 2.f * (y * z + w * x) * vy + (1.f - 2.f * (x * x + y * y)) * vz;
}

// 64 lanes per Gaussian (lane = float index within the 59-float row, looping if wider), 4 Gaussians per block
__global__ void __launch_bounds__(256)
densify_emit_kernel(int N, const unsigned char* __restrict__ flags, const int* __restrict__ pos,
                    const float* __restrict__ samples, const float* __restrict__ old_p, const float* __restrict__ old_m,
                    const float* __restrict__ old_v, float* __restrict__ new_p, float* __restrict__ new_m,
                    float* __restrict__ new_v, EmitLayout L) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const unsigned f = flags[i];
    if ((f & (kFlagKeepOld | kFlagKeepChild | kFlagKeepDup)) == 0) return;
    const int lane = threadIdx.x & 63;
    const int split_rank = pos[i], slot_old = pos[(size_t)N + i], slot_child = pos[(size_t)2 * N + i],
              slot_dup = pos[(size_t)3 * N + i];
    for (int c = lane; c < L.total_width; c += 64) {
        int g = 0, k = c;
        while (k >= L.width[g]) { k -= L.width[g]; ++g; }
        const int wg = L.width[g];
        const size_t src = (size_t)L.old_begin[g] + (size_t)i * wg + k;
        const float p = old_p[src];
        if (f & kFlagKeepOld) {
            const size_t dst = (size_t)L.new_begin[g] + (size_t)slot_old * wg + k;
            new_p[dst] = p; new_m[dst] = old_m[src]; new_v[dst] = old_v[src];
        }
        float p_new = p;                                     // value the children / the duplicate carry
        if (g == 1 && (f & kFlagSplit)) p_new = shrink(p);
        if (f & kFlagKeepChild) {
            for (int s = 0; s < L.n_samples; ++s) {
                float val = p_new;
                if (g == 0) {                                // mean + R(q) (exp(scale) * sample), ORIGINAL scale
                    const float* sc = old_p + L.old_begin[1] + (size_t)i * 3;
                    const float* z = samples + ((size_t)s * L.n_split + split_rank) * 3;
                    float r[3];
                    quat_rotate(old_p + L.old_begin[2] + (size_t)i * 4, expf(sc[0]) * z[0], expf(sc[1]) * z[1],
                                expf(sc[2]) * z[2], r);
                    val = r[k] + p;
                }
                const size_t row = (size_t)L.k_old + (size_t)s * L.k_child + slot_child;
                const size_t dst = (size_t)L.new_begin[g] + row * wg + k;
                new_p[dst] = val; new_m[dst] = 0.f; new_v[dst] = 0.f;      // dup_in_all_optim: zero moments
            }
        }
        if (f & kFlagKeepDup) {
            const size_t row = (size_t)L.k_old + (size_t)L.n_samples * L.k_child + slot_dup;
            const size_t dst = (size_t)L.new_begin[g] + row * wg + k;
            new_p[dst] = p_new; new_m[dst] = 0.f; new_v[dst] = 0.f;
        }
    }
}

__global__ void __launch_bounds__(256)
densify_reset_opacity_kernel(int N, float* __restrict__ opac, float* __restrict__ m, float* __restrict__ v, float max_logit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    opac[i] = fminf(opac[i], max_logit);
    m[i] = 0.f; v[i] = 0.f;
}

}  // namespace qed

using namespace qed;

extern "C" int qed_densify_accumulate(int32_t N, const float* absgrad, int32_t stride_floats, const int32_t* radii,
                                      float inv_max_dim, float* xys_grad_norm, float* vis_counts, float* max_2Dsize,
                                      void* stream) {
    QED_REQUIRE(N >= 0 && stride_floats >= 2, "bad arguments");
    if (N == 0) return QED_OK;
    QED_REQUIRE(absgrad && radii && xys_grad_norm && vis_counts && max_2Dsize, "null buffers");
    hipLaunchKernelGGL(densify_accumulate_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, absgrad,
                       stride_floats, radii, inv_max_dim, xys_grad_norm, vis_counts, max_2Dsize);
    return check_launch("qed_densify_accumulate");
}

extern "C" int64_t qed_densify_pos_ints(int32_t N) {
    if (N < 0) return QED_E_INVALID_ARG;
    return 4ll * N + 4ll * ((N + 255) / 256);
}

extern "C" int qed_densify_classify(int32_t N, const float* scales, const float* opacities, const float* xys_grad_norm,
                                    const float* vis_counts, const float* max_2Dsize, int32_t densify,
                                    float half_max_dim, float densify_grad_thresh, float densify_size_thresh,
                                    float split_screen_size, float cull_alpha_thresh, float cull_scale_thresh,
                                    float cull_screen_size, uint8_t* flags, int32_t* pos, int32_t* totals,
                                    void* stream) {
    QED_REQUIRE(N >= 1 && scales && opacities && flags && pos && totals, "bad arguments");
    QED_REQUIRE(!densify || (xys_grad_norm && vis_counts && max_2Dsize), "densification needs the accumulated statistics");
    ClassifyArgs a{densify, half_max_dim, densify_grad_thresh, densify_size_thresh, split_screen_size,
                   cull_alpha_thresh, cull_scale_thresh, cull_screen_size};
    hipStream_t st = (hipStream_t)stream;
    const int nb = (N + 255) / 256;
    int* block_counts = pos + (size_t)4 * N;                 // scratch behind the four position columns
    hipLaunchKernelGGL(densify_classify_kernel, dim3(nb), dim3(256), 0, st, N, scales, opacities, xys_grad_norm,
                       vis_counts, max_2Dsize, a, flags, block_counts);
    hipLaunchKernelGGL(densify_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, nb, block_counts, totals);
    hipLaunchKernelGGL(densify_positions_kernel, dim3(nb), dim3(256), 0, st, N, flags, block_counts, pos);
    return check_launch("qed_densify_classify");
}

extern "C" int qed_densify_emit(int32_t N, int32_t n_samples, const uint8_t* flags, const int32_t* pos,
                                const int32_t* h_totals, const float* samples, const float* old_params,
                                const float* old_exp_avg, const float* old_exp_avg_sq, const int64_t* h_old_begin,
                                float* new_params, float* new_exp_avg, float* new_exp_avg_sq,
                                const int64_t* h_new_begin, void* stream) {
    QED_REQUIRE(N >= 1 && n_samples >= 1 && flags && pos && h_totals && h_old_begin && h_new_begin, "bad arguments");
    QED_REQUIRE(old_params && old_exp_avg && old_exp_avg_sq, "null source buffers");
    const int n_split = h_totals[0], k_old = h_totals[1], k_child = h_totals[2], k_dup = h_totals[3];
    const long long n_new = (long long)k_old + (long long)n_samples * k_child + k_dup;
    QED_REQUIRE(n_split == 0 || samples, "samples required when anything splits");
    if (n_new == 0) return QED_OK;
    QED_REQUIRE(new_params && new_exp_avg && new_exp_avg_sq, "null destination buffers");
    EmitLayout L;
    L.total_width = 0;
    for (int g = 0; g < 6; ++g) {
        const long long w = (h_old_begin[g + 1] - h_old_begin[g]) / N;
        QED_REQUIRE(w * N == h_old_begin[g + 1] - h_old_begin[g], "old group sizes must be multiples of N");
        QED_REQUIRE(h_new_begin[g + 1] - h_new_begin[g] == w * n_new, "new group sizes must be width x N'");
        L.width[g] = (int)w;
        L.total_width += (int)w;
    }
    QED_REQUIRE(L.width[0] == 3 && L.width[1] == 3 && L.width[2] == 4 && L.width[3] == 1,
                "group order: means, scales, quats, opacities, features_dc, features_rest");
    for (int g = 0; g < 7; ++g) { L.old_begin[g] = h_old_begin[g]; L.new_begin[g] = h_new_begin[g]; }
    L.k_old = k_old; L.k_child = k_child; L.n_split = n_split; L.n_samples = n_samples;
    hipLaunchKernelGGL(densify_emit_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, flags, pos, samples,
                       old_params, old_exp_avg, old_exp_avg_sq, new_params, new_exp_avg, new_exp_avg_sq, L);
    return check_launch("qed_densify_emit");
}

extern "C" int qed_densify_reset_opacity(int32_t N, float* opacities, float* exp_avg, float* exp_avg_sq, float max_logit,
                                         void* stream) {
    QED_REQUIRE(N >= 0, "bad arguments");
    if (N == 0) return QED_OK;
    QED_REQUIRE(opacities && exp_avg && exp_avg_sq, "null buffers");
    hipLaunchKernelGGL(densify_reset_opacity_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, opacities,
                       exp_avg, exp_avg_sq, max_logit);
    return check_launch("qed_densify_reset_opacity");
}
