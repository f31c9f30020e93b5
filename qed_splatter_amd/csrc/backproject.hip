// SURVEY 8(f) rank 4 (second half): depth back-projection for the initial point cloud.
//
// The reference's `qed-init-pc` (create_init_pointcloud.py:148-196) cleans a depth frame (non-finite and
// non-positive values -> 0), converts the OpenGL camera-to-world pose to an OpenCV world-to-camera
// extrinsic (:61-70) and hands both to Open3D's PointCloud.create_from_depth_image(depth, K, w2c,
// depth_scale = 1, depth_max, stride) on the CPU.  Open3D is not vendored; its unprojection is restated in
// oracle/backproject_oracle.py: every stride-th pixel (u, v) with 0 < d < depth_max gives the camera point
// ((u - cx) d / fx, (v - cy) d / fy, d), which the inverse extrinsic carries to the world.  With the
// OpenGL c2w = [R | t] that inverse is R diag(1, -1, -1) p + t, so no matrix is inverted here.
//
// Three launches: count the valid pixels per 256-pixel chunk, scan the chunk counts (qed_isect_scan: it
// also reports overflow of the caller's buffer), then write the points in row-major pixel order
// (deterministic, unlike an atomic-counter compaction).  One read of the depth image, 12 B per point out.
#include "qed_common.h"

namespace qed {

struct BackprojectArgs {
    int H, W, stride, gw, gh;        // gw x gh = the strided pixel grid
    float fx, fy, cx, cy, depth_max;
    float R[9], t[3];                // OpenGL camera-to-world
};

__device__ __forceinline__ bool bp_valid(const float* __restrict__ depth, const BackprojectArgs& a, long long i, int& u,
                                         int& v, float& d) {
    if (i >= (long long)a.gw * a.gh) return false;
    const int gy = (int)(i / a.gw), gx = (int)(i - (long long)gy * a.gw);
    u = gx * a.stride; v = gy * a.stride;
    d = depth[(size_t)v * a.W + u];
    return isfinite(d) && d > 0.f && d < a.depth_max;
}

__global__ void __launch_bounds__(256)
backproject_count_kernel(const float* __restrict__ depth, BackprojectArgs a, int* __restrict__ block_counts) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    int u, v; float d;
    const bool ok = bp_valid(depth, a, i, u, v, d);
    __shared__ int s_cnt[4];
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ void __launch_bounds__(256)
backproject_write_kernel(const float* __restrict__ depth, BackprojectArgs a, const int* __restrict__ block_offsets,
                         const int* __restrict__ n_points, float* __restrict__ points) {
    if (n_points[0] == 0) return;                       // nothing valid, or the buffer was too small
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    int u = 0, v = 0; float d = 0.f;
    const bool ok = bp_valid(depth, a, i, u, v, d);
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(ok);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (!ok) return;
    int rank = block_offsets[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += s_cnt[w];
    const float px = ((float)u - a.cx) * d / a.fx, py = -((float)v - a.cy) * d / a.fy, pz = -d;    // OpenGL camera axes
    points[3 * (size_t)rank] = a.R[0] * px + a.R[1] * py + a.R[2] * pz + a.t[0];
    points[3 * (size_t)rank + 1] = a.R[3] * px + a.R[4] * py + a.R[5] * pz + a.t[1];
    points[3 * (size_t)rank + 2] = a.R[6] * px + a.R[7] * py + a.R[8] * pz + a.t[2];
}

}  // namespace qed

using namespace qed;

extern "C" int qed_isect_scan(const int32_t* block_sums, int32_t n_blocks, int32_t* block_offsets, int32_t* n_isect,
                              int64_t capacity, int32_t* status, void* stream);

extern "C" int64_t qed_backproject_workspace_ints(int32_t height, int32_t width, int32_t stride) {
    if (height <= 0 || width <= 0 || stride <= 0) return QED_E_INVALID_ARG;
    const long long g = (long long)((width + stride - 1) / stride) * ((height + stride - 1) / stride);
    return 2 * ((g + 255) / 256) + 8;
}

extern "C" int qed_backproject_depth(int32_t height, int32_t width, const float* depth, float fx, float fy, float cx,
                                     float cy, const float* h_c2w_opengl, float depth_max, int32_t stride,
                                     int64_t capacity, float* points, int32_t* n_points, int32_t* workspace,
                                     int32_t* status, void* stream) {
    QED_REQUIRE(height > 0 && width > 0 && stride > 0 && capacity >= 0, "bad extents");
    QED_REQUIRE(depth && h_c2w_opengl && n_points && workspace && status && (points || capacity == 0), "null buffers");
    QED_REQUIRE(fx != 0.f && fy != 0.f, "zero focal length");
    BackprojectArgs a;
    a.H = height; a.W = width; a.stride = stride;
    a.gw = (width + stride - 1) / stride; a.gh = (height + stride - 1) / stride;
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.depth_max = depth_max;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) a.R[3 * i + j] = h_c2w_opengl[4 * i + j];
        a.t[i] = h_c2w_opengl[4 * i + 3];
    }
    const long long g = (long long)a.gw * a.gh;
    const int nb = (int)((g + 255) / 256);
    int* block_counts = workspace;
    int* block_offsets = workspace + nb;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(backproject_count_kernel, dim3(nb), dim3(256), 0, st, depth, a, block_counts);
    const int rc = qed_isect_scan(block_counts, nb, block_offsets, n_points, capacity, status, stream);
    if (rc != QED_OK) return rc;
    hipLaunchKernelGGL(backproject_write_kernel, dim3(nb), dim3(256), 0, st, depth, a, (const int*)block_offsets,
                       (const int*)n_points, points);
    return check_launch("qed_backproject_depth");
}
