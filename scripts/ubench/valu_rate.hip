// Micro-benchmark: wave64 VALU issue rate on gfx950 as a function of waves per SIMD and of the
// instruction mix (independent v_fma chains, dependent chain, v_exp, v_cndmask with SGPR mask,
// v_readlane).  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.0001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent fma chains
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
            }
        } else if (MODE == 1) {  // one dependent chain
#pragma unroll
            for (int j = 0; j < 64; ++j) a0 = __builtin_fmaf(a0, b, c);
        } else if (MODE == 2) {  // v_exp chain (8 independent)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
            }
        } else if (MODE == 3) {  // packed fma: 4 independent float2 chains
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
            const f2 bb = {b, b}, cc = {c, c};
#pragma unroll
            for (int j = 0; j < 16; ++j) { p0 = p0 * bb + cc; p1 = p1 * bb + cc; p2 = p2 * bb + cc; p3 = p3 * bb + cc; }
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (MODE == 4) {  // v_cmp -> SGPR mask -> v_cndmask pairs
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned long long m = __ballot(a0 > a1);
                float d; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a2), "v"(a3), "s"(m));
                a0 = a0 + d; 
                unsigned long long m2 = __ballot(a4 > a5);
                float e; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(e) : "v"(a6), "v"(a7), "s"(m2));
                a4 = a4 + e;
            }
        } else if (MODE == 5) {  // v_readlane x 8 + 8 fma using the SGPRs
            const int t = i & 63;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), t));
                float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), t));
                a2 = __builtin_fmaf(a2, s0, c); a3 = __builtin_fmaf(a3, s1, c);
                a4 = __builtin_fmaf(a4, s0, c); a5 = __builtin_fmaf(a5, s1, c);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
double run(int waves_per_simd, int iters, int ops_per_iter) {
    const int blocks = 256 * 4 * waves_per_simd;   // one wave per block
    float* out; hipMalloc(&out, sizeof(float) * blocks * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    // wave-instructions per SIMD per second
    const double winstr = (double)waves_per_simd * iters * ops_per_iter;
    return winstr / (ms * 1e-3);
}

int main() {
    const int iters = 20000;
    printf("mode waves/SIMD  Gwave-instr/s/SIMD  (cycles per wave-instr at 2.4 GHz)\n");
    const char* names[] = {"fma x8 indep", "fma dependent", "v_exp x8 indep", "v_pk_fma x4 indep", "cmp+cndmask+add", "readlane+fma"};
    const int ops[] = {64, 64, 64, 64, 64, 48};
    for (int w : {1, 2, 4, 8}) {
        double r[6] = {run<0>(w, iters, ops[0]), run<1>(w, iters, ops[1]), run<2>(w, iters, ops[2]), run<3>(w, iters, ops[3]),
                       run<4>(w, iters, ops[4]), run<5>(w, iters, ops[5])};
        for (int m = 0; m < 6; ++m) printf("%-18s %d  %8.3f  (%.2f)\n", names[m], w, r[m] / 1e9, 2.4e9 / r[m]);
    }
    return 0;
}
