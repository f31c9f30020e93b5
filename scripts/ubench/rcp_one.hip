// Is v_rcp_f32 exact where composite_bwd_kernel relies on it?  bwd_quadrant folds "this pixel is not valid for the Gaussian"
// into alpha = 0 and then computes T * v_rcp_f32(1 - 0): T survives bit for bit only if v_rcp_f32(1.0f) == 1.0f.
// Prints the bit pattern of v_rcp_f32(1.0f) and checks every power of two 2^-126 .. 2^126 as well.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/rcp_one scripts/ubench/rcp_one.hip && scripts/ubench/rcp_one
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

__global__ void rcp_kernel(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcpf(in[i]);
}

int main() {
    const int n = 253;
    float h_in[n], h_out[n];
    for (int i = 0; i < n; ++i) h_in[i] = std::ldexp(1.0f, i - 126);
    float *d_in, *d_out;
    if (hipMalloc(&d_in, sizeof(h_in)) != hipSuccess || hipMalloc(&d_out, sizeof(h_out)) != hipSuccess) return 2;
    (void)hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(rcp_kernel, dim3(1), dim3(256), 0, 0, d_in, d_out, n);
    (void)hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    unsigned bits;
    std::memcpy(&bits, &h_out[126], 4);
    std::printf("v_rcp_f32(1.0f) = %.9g (0x%08x) %s\n", h_out[126], bits, bits == 0x3f800000u ? "exact" : "NOT exact");
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (h_out[i] != std::ldexp(1.0f, 126 - i)) ++bad;
    std::printf("powers of two 2^-126 .. 2^126: %d of %d inexact\n", bad, n);
    return bits == 0x3f800000u ? 0 : 1;
}
