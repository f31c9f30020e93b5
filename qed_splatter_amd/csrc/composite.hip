// K6 / K7: per-tile alpha compositing, forward and backward, for gfx950 (wave64).
//
// Replaces gsplat rasterize_to_pixels fwd/bwd behind model.py:267-288.
//
// One 256-thread workgroup per 16x16 tile = 4 waves, each wave owning one 8x8 pixel quadrant (so a
// wave's early-out and "nobody touches this Gaussian" tests see a compact pixel block).  The
// tile's run of the depth-sorted list is streamed in batches of 256: every thread gathers one
// 48-byte splat record (three 16-byte loads) for the NEXT batch into registers while the current
// batch, staged in LDS, is consumed with wave-uniform (broadcast) LDS reads.
//
// Backward: per-pixel gradients of one Gaussian are reduced over the 64 lanes with
// v_permlane32_swap / v_permlane16_swap (which halve the number of live values at each level)
// plus one DPP row reduction, then accumulated per tile in LDS and flushed with ONE 64-byte-row
// atomic add per (tile, Gaussian).
#include "qed_common.h"

namespace qed {

constexpr int kBatch = 256;

__device__ __forceinline__ void pixel_of_thread(int tid, int& lx, int& ly) {
    // wave w -> quadrant (w & 1, w >> 1); lane l -> (l & 7, l >> 3) inside the quadrant
    const int w = tid >> 6, l = tid & 63;
    lx = ((w & 1) << 3) | (l & 7);
    ly = ((w >> 1) << 3) | (l >> 3);
}


// ---- exact-conservative quadrant culling ---------------------------------------------------------
// A Gaussian contributes to a pixel only if alpha = min(.999, o e^-sigma) >= 1/255, i.e.
// sigma <= tau = ln(255 o).  For each 8x8 quadrant (= one wave) the staging thread minimises the
// quadratic form sigma over the quadrant's pixel-centre rectangle (convex: the minimum is 0 if the
// mean lies inside, otherwise it is on one of the four edges) and drops the Gaussian for that wave
// when sigma_min > tau + margin.  The margin covers fp32 rounding of both this bound and the
// per-pixel evaluation, so a culled Gaussian is one every pixel of the quadrant would have
// skipped anyway: results are bit-identical to the unculled loop.
__device__ __forceinline__ float rect_min_sigma(float a, float b, float c, float x0, float x1, float y0, float y1,
                                                float& scale) {
    const float ax = fmaxf(fabsf(x0), fabsf(x1)), ay = fmaxf(fabsf(y0), fabsf(y1));
    scale = a * ax * ax + c * ay * ay + fabsf(b) * ax * ay;
    if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return 0.f;
    const float ia = 1.f / a, ic = 1.f / c;
    float m;
    {
        const float y = fminf(fmaxf(-b * x0 * ic, y0), y1);
        m = 0.5f * (a * x0 * x0 + c * y * y) + b * x0 * y;
    }
    {
        const float y = fminf(fmaxf(-b * x1 * ic, y0), y1);
        m = fminf(m, 0.5f * (a * x1 * x1 + c * y * y) + b * x1 * y);
    }
    {
        const float x = fminf(fmaxf(-b * y0 * ia, x0), x1);
        m = fminf(m, 0.5f * (a * x * x + c * y0 * y0) + b * x * y0);
    }
    {
        const float x = fminf(fmaxf(-b * y1 * ia, x0), x1);
        m = fminf(m, 0.5f * (a * x * x + c * y1 * y1) + b * x * y1);
    }
    return m;
}

// Ballot, for each of the 4 quadrants of the tile at pixel origin (ox, oy), which of this wave's 64
// staged records can contribute; lane 0 stores the 4 masks to s_mask[q][wave].
__device__ __forceinline__ void stage_cull_masks(bool present, const float4& r0, const float4& r1, float tau,
                                                 float ox, float oy, unsigned long long (*s_mask)[4], int wid,
                                                 int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // pixel centres of quadrant q relative to the mean
        const float x0 = ox + (float)((q & 1) << 3) + 0.5f - r0.x, x1 = x0 + 7.f;
        const float y0 = oy + (float)((q >> 1) << 3) + 0.5f - r0.y, y1 = y0 + 7.f;
        float scale;
        const float smin = rect_min_sigma(r0.z, r0.w, r1.x, x0, x1, y0, y1, scale);
        const bool keep = present && !(smin > tau + 1e-3f + 8e-6f * scale);
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_mask[q][wid] = m;
    }
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// ================================================================================================
// forward
// ================================================================================================
template <int CH>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(int C, const float4* __restrict__ splats, const int* __restrict__ flatten_ids,
                     const int* __restrict__ offsets, int width, int height, int tile_w, int tile_h,
                     const float* __restrict__ backgrounds, float* __restrict__ render, float* __restrict__ alpha_out,
                     int* __restrict__ last_ids) {
    __shared__ float4 s_q0[kBatch];   // x, y, conic_a, conic_b
    __shared__ float4 s_q1[kBatch];   // conic_c, opacity, r, g
    __shared__ float2 s_q2[kBatch];   // b, depth
    __shared__ unsigned long long s_mask[4][4];       // [quadrant = consuming wave][staging wave]
    __shared__ int s_done[4];

    const int tile = blockIdx.x;                      // cam * T + ty * tile_w + tx
    const int n_tiles = tile_w * tile_h;
    const int cam = tile / n_tiles;
    const int t_in = tile - cam * n_tiles;
    const int ty = t_in / tile_w, tx = t_in - ty * tile_w;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    int lx, ly;
    pixel_of_thread(tid, lx, ly);
    const int ix = tx * QED_TILE + lx, iy = ty * QED_TILE + ly;
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const bool inside = ix < width && iy < height;
    const float ox = (float)(tx * QED_TILE), oy = (float)(ty * QED_TILE);

    const int start = offsets[tile], end = offsets[tile + 1];
    const int nb = (end - start + kBatch - 1) / kBatch;

    float T = 1.f;
    float out[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) out[k] = 0.f;
    int cur = 0;
    bool done = !inside;

    // prefetch batch 0
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    bool present = false;
    {
        const int idx = start + tid;
        present = idx < end;
        if (present) {
            const size_t g = (size_t)flatten_ids[idx];
            r0 = splats[3 * g]; r1 = splats[3 * g + 1]; r2 = splats[3 * g + 2];
        }
    }
    for (int b = 0; b < nb; ++b) {
        __syncthreads();                               // LDS of the previous batch fully consumed
        s_q0[tid] = r0; s_q1[tid] = r1; s_q2[tid] = make_float2(r2.x, r2.y);
        stage_cull_masks(present, r0, r1, r2.z, ox, oy, s_mask, wid, lane);
        const bool wave_done = __all(done);
        if (lane == 0) s_done[wid] = wave_done;
        __syncthreads();
        if (s_done[0] && s_done[1] && s_done[2] && s_done[3]) break;
        // issue the gather of the next batch; it lands while this batch is composited
        if (b + 1 < nb) {
            const int idx = start + (b + 1) * kBatch + tid;
            present = idx < end;
            if (present) {
                const size_t g = (size_t)flatten_ids[idx];
                r0 = splats[3 * g]; r1 = splats[3 * g + 1]; r2 = splats[3 * g + 2];
            }
        }
        if (wave_done) continue;
        const int batch_start = start + b * kBatch;
        bool wave_finished = false;
#pragma unroll 1
        for (int sw = 0; sw < 4 && !wave_finished; ++sw) {
          unsigned long long m = uniform_u64(s_mask[wid][sw]);
          while (m) {
            const int t = (sw << 6) + __builtin_ctzll(m);
            m &= m - 1;
            const float4 q0 = s_q0[t];
            const float4 q1 = s_q1[t];
            const float dx = q0.x - px, dy = q0.y - py;
            const float sigma = 0.5f * (q0.z * dx * dx + q1.x * dy * dy) + q0.w * dx * dy;
            const float a = fminf(kAlphaMax, q1.y * __expf(-sigma));
            const bool ok = !done && sigma >= 0.f && a >= kAlphaMin;
            if (!__any(ok)) continue;
            const float nT = T * (1.f - a);
            const bool term = ok && nT <= kTMin;
            done = done || term;
            const bool acc = ok && !term;
            const float w = acc ? a * T : 0.f;
            const float2 q2 = s_q2[t];
            out[0] += q1.z * w; out[1] += q1.w * w; out[2] += q2.x * w;
            if constexpr (CH == 4) out[3] += q2.y * w;
            T = acc ? nT : T;
            cur = acc ? batch_start + t : cur;
            if (__all(done)) { wave_finished = true; break; }
          }
        }
    }
    if (inside) {
        const size_t pix = ((size_t)cam * height + iy) * width + ix;
        if (backgrounds != nullptr) {
#pragma unroll
            for (int k = 0; k < CH; ++k) out[k] += T * backgrounds[cam * CH + k];
        }
        if constexpr (CH == 4) {
            *reinterpret_cast<float4*>(render + 4 * pix) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
            render[3 * pix] = out[0]; render[3 * pix + 1] = out[1]; render[3 * pix + 2] = out[2];
        }
        alpha_out[pix] = 1.f - T;
        last_ids[pix] = cur;
    }
}

// ================================================================================================
// backward
// ================================================================================================
__device__ __forceinline__ void swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

// Reduce 12 per-lane values over the 64 lanes.  On return w[j] (j = 0..2) holds, in EVERY lane of
// DPP row r (lanes 16r .. 16r+15), the wave total of value index kRowValue[r] + 4 j, with
// kRowValue = {0, 2, 1, 3}.
__device__ __forceinline__ void wave_reduce12(const float* v, float* w) {
    float u[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        swap32(a, b);              // a = [a.lo | b.lo], b = [a.hi | b.hi]
        u[i] = a + b;              // lanes 0-31: value 2i ; lanes 32-63: value 2i+1
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float a = u[2 * j], b = u[2 * j + 1];
        swap16(a, b);              // a = [a.r0, b.r0, a.r2, b.r2], b = [a.r1, b.r1, a.r3, b.r3]
        w[j] = row16_sum(a + b);   // row0: 4j, row1: 4j+2, row2: 4j+1, row3: 4j+3
    }
}

// vsplat row layout (QED_VSPLAT_FLOATS = 16):
//  0 v_x  1 v_y  2 |v_x|  3 |v_y|  4 v_conic_a  5 v_conic_b  6 v_conic_c  7 v_opacity  8 v_r  9 v_g  10 v_b  11 v_depth
template <int CH>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(int C, const float4* __restrict__ splats, const int* __restrict__ flatten_ids,
                     const int* __restrict__ offsets, int width, int height, int tile_w, int tile_h,
                     const float* __restrict__ backgrounds, const float* __restrict__ render_alpha,
                     const int* __restrict__ last_ids, const float* __restrict__ v_render,
                     const float* __restrict__ v_alpha, float* __restrict__ vsplat) {
    __shared__ float4 s_q0[kBatch];
    __shared__ float4 s_q1[kBatch];
    __shared__ float2 s_q2[kBatch];
    __shared__ int s_id[kBatch];
    __shared__ float s_acc[kBatch][12];
    __shared__ int s_touched[kBatch];
    __shared__ unsigned long long s_mask[4][4];       // [quadrant = consuming wave][staging wave]
    __shared__ int s_wmax[4];

    const int tile = blockIdx.x;
    const int n_tiles = tile_w * tile_h;
    const int cam = tile / n_tiles;
    const int t_in = tile - cam * n_tiles;
    const int ty = t_in / tile_w, tx = t_in - ty * tile_w;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    int lx, ly;
    pixel_of_thread(tid, lx, ly);
    const int ix = tx * QED_TILE + lx, iy = ty * QED_TILE + ly;
    const float px = (float)ix + 0.5f, py = (float)iy + 0.5f;
    const bool inside = ix < width && iy < height;
    const size_t pix = ((size_t)cam * height + (inside ? iy : 0)) * width + (inside ? ix : 0);
    const float ox = (float)(tx * QED_TILE), oy = (float)(ty * QED_TILE);

    const int start = offsets[tile], end = offsets[tile + 1];
    if (end <= start) return;

    float T_final = 1.f, vra = 0.f;
    float vr[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) vr[k] = 0.f;
    int bin_final = -1;
    if (inside) {
        T_final = 1.f - render_alpha[pix];
        bin_final = last_ids[pix];
        vra = v_alpha[pix];
        if constexpr (CH == 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(v_render + 4 * pix);
            vr[0] = t4.x; vr[1] = t4.y; vr[2] = t4.z; vr[3] = t4.w;
        } else {
            vr[0] = v_render[3 * pix]; vr[1] = v_render[3 * pix + 1]; vr[2] = v_render[3 * pix + 2];
        }
        if (backgrounds != nullptr) {
            // render = sum + T_final * bg  ->  d render / d T_final folds into the alpha gradient
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < CH; ++k) acc += backgrounds[cam * CH + k] * vr[k];
            vra -= acc;
        }
    }
    // a pixel that composited nothing has last_id 0 and T_final 1: it only "owns" index 0
    const int wave_last = wave_max_i(bin_final);
    if (lane == 0) s_wmax[wid] = wave_last;
    __syncthreads();
    const int tile_last = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    const int eff_end = min(end, tile_last + 1);
    if (eff_end <= start) return;
    const int nb = (eff_end - start + kBatch - 1) / kBatch;

    float T = T_final;
    float buf[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) buf[k] = 0.f;

    // batches run back to front; inside a batch slot t holds sorted index (batch_hi - t)
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    int rid = -1;
    {
        const int idx = eff_end - 1 - tid;
        if (idx >= start) {
            rid = flatten_ids[idx];
            r0 = splats[3 * (size_t)rid]; r1 = splats[3 * (size_t)rid + 1]; r2 = splats[3 * (size_t)rid + 2];
        }
    }
    for (int b = 0; b < nb; ++b) {
        __syncthreads();                               // previous batch consumed and flushed
        s_q0[tid] = r0; s_q1[tid] = r1; s_q2[tid] = make_float2(r2.x, r2.y); s_id[tid] = rid;
#pragma unroll
        for (int k = 0; k < 12; ++k) s_acc[tid][k] = 0.f;
        s_touched[tid] = 0;
        stage_cull_masks(rid >= 0, r0, r1, r2.z, ox, oy, s_mask, wid, lane);
        __syncthreads();
        const int batch_hi = eff_end - 1 - b * kBatch;            // sorted index of slot 0
        const int bn = min(kBatch, batch_hi - start + 1);
        if (b + 1 < nb) {
            const int idx = batch_hi - kBatch - tid;
            rid = -1;
            if (idx >= start) {
                rid = flatten_ids[idx];
                r0 = splats[3 * (size_t)rid]; r1 = splats[3 * (size_t)rid + 1]; r2 = splats[3 * (size_t)rid + 2];
            }
        }
        // slots whose index is beyond every pixel of this wave can be skipped wholesale
        const int t0 = max(0, batch_hi - wave_last);
#pragma unroll 1
        for (int sw = t0 >> 6; sw < 4; ++sw) {
          unsigned long long m = uniform_u64(s_mask[wid][sw]);
          if (sw == (t0 >> 6)) m &= ~0ull << (t0 & 63);
          while (m) {
            const int t = (sw << 6) + __builtin_ctzll(m);
            m &= m - 1;
            const int idx = batch_hi - t;
            const float4 q0 = s_q0[t];
            const float4 q1 = s_q1[t];
            const float dx = q0.x - px, dy = q0.y - py;
            const float sigma = 0.5f * (q0.z * dx * dx + q1.x * dy * dy) + q0.w * dx * dy;
            const float vis = __expf(-sigma);
            const float opv = q1.y * vis;
            const float a = fminf(kAlphaMax, opv);
            const bool valid = idx <= bin_final && sigma >= 0.f && a >= kAlphaMin;
            if (!__any(valid)) continue;
            const float2 q2 = s_q2[t];
            float g[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) g[k] = 0.f;
            if (valid) {
                const float ra = 1.f / (1.f - a);
                T *= ra;
                const float fac = a * T;
                float col[4] = {q1.z, q1.w, q2.x, q2.y};
                float v_a = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    g[8 + k] = fac * vr[k];
                    v_a += (col[k] * T - buf[k] * ra) * vr[k];
                    buf[k] += col[k] * fac;
                }
                v_a += T_final * ra * vra;
                if (opv <= kAlphaMax) {
                    const float v_sigma = -opv * v_a;
                    g[4] = 0.5f * v_sigma * dx * dx;
                    g[5] = v_sigma * dx * dy;
                    g[6] = 0.5f * v_sigma * dy * dy;
                    g[0] = v_sigma * (q0.z * dx + q0.w * dy);
                    g[1] = v_sigma * (q0.w * dx + q1.x * dy);
                    g[2] = fabsf(g[0]);
                    g[3] = fabsf(g[1]);
                    g[7] = vis * v_a;
                }
            }
            float w[3];
            wave_reduce12(g, w);
            // lanes 0,16,32,48 publish: row r holds value (r==0?0 : r==1?2 : r==2?1 : 3) + 4j
            if ((lane & 15) == 0) {
                const int r = lane >> 4;
                const int vbase = ((r & 1) << 1) | (r >> 1);
                atomicAdd(&s_acc[t][vbase], w[0]);
                atomicAdd(&s_acc[t][vbase + 4], w[1]);
                atomicAdd(&s_acc[t][vbase + 8], w[2]);
                if (lane == 0) s_touched[t] = 1;
            }
          }
        }
        __syncthreads();
        // flush: 16 lanes per Gaussian -> one 64-byte row per atomic request
        for (int t = tid >> 4; t < bn; t += 16) {
            if (!s_touched[t]) continue;
            const int k = tid & 15;
            if (k < 12) {
                const float v = s_acc[t][k];
                if (v != 0.f) atomicAdd(&vsplat[(size_t)s_id[t] * QED_VSPLAT_FLOATS + k], v);
            }
        }
    }
}

}  // namespace qed

using namespace qed;

extern "C" int qed_composite_fwd(int32_t C, int32_t N, const float* splats, const int32_t* flatten_ids,
                                 const int32_t* offsets, int32_t width, int32_t height, int32_t tile_w,
                                 int32_t tile_h, int32_t channels, const float* backgrounds, float* render,
                                 float* alpha, int32_t* last_ids, void* stream) {
    QED_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "bad extents");
    QED_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (RGB) or 4 (RGB+D)");
    QED_REQUIRE(tile_w == (width + QED_TILE - 1) / QED_TILE && tile_h == (height + QED_TILE - 1) / QED_TILE,
                "tile grid does not match the image (tile size is 16)");
    QED_REQUIRE(offsets && render && alpha && last_ids, "null buffers");
    // flatten_ids may be NULL when the sorted list is empty (offsets are then all zero)
    QED_REQUIRE(N == 0 || splats, "null splat buffer");
    const long long grid = (long long)C * tile_w * tile_h;
    QED_REQUIRE(grid < (1ll << 31), "too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (channels == 4)
        hipLaunchKernelGGL(composite_fwd_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render, alpha, last_ids);
    else
        hipLaunchKernelGGL(composite_fwd_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render, alpha, last_ids);
    return check_launch("qed_composite_fwd");
}

extern "C" int qed_composite_bwd(int32_t C, int32_t N, const float* splats, const int32_t* flatten_ids,
                                 const int32_t* offsets, int32_t width, int32_t height, int32_t tile_w,
                                 int32_t tile_h, int32_t channels, const float* backgrounds,
                                 const float* render_alpha, const int32_t* last_ids, const float* v_render,
                                 const float* v_alpha, float* vsplat, void* stream) {
    QED_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "bad extents");
    QED_REQUIRE(channels == 3 || channels == 4, "channels must be 3 (RGB) or 4 (RGB+D)");
    QED_REQUIRE(tile_w == (width + QED_TILE - 1) / QED_TILE && tile_h == (height + QED_TILE - 1) / QED_TILE,
                "tile grid does not match the image (tile size is 16)");
    QED_REQUIRE(offsets && render_alpha && last_ids && v_render && v_alpha, "null buffers");
    if (N == 0) return QED_OK;
    QED_REQUIRE(splats && vsplat, "null splat buffers");
    const long long grid = (long long)C * tile_w * tile_h;
    QED_REQUIRE(grid < (1ll << 31), "too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (channels == 4)
        hipLaunchKernelGGL(composite_bwd_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render_alpha, last_ids,
                           v_render, v_alpha, vsplat);
    else
        hipLaunchKernelGGL(composite_bwd_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, C, (const float4*)splats,
                           flatten_ids, offsets, width, height, tile_w, tile_h, backgrounds, render_alpha, last_ids,
                           v_render, v_alpha, vsplat);
    return check_launch("qed_composite_bwd");
}
