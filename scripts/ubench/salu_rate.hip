// Micro-benchmark: scalar-ALU and branch issue rates per CU on gfx950 (is the SALU shared by the 4 SIMDs?)
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void k(int* out, int iters, int seed) {
    int a = seed, b = seed + 1, c = seed + 2, d = seed + 3;
    float v = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 64 independent-ish scalar ops
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("s_add_u32 %0, %0, %4\n s_xor_b32 %1, %1, %5\n s_add_u32 %2, %2, %6\n s_xor_b32 %3, %3, %7"
                             : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : "s"(b), "s"(c), "s"(d), "s"(a) : "scc");
        } else if (MODE == 1) { // 32 scalar + 32 vector interleaved
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("s_add_u32 %0, %0, %2\n s_xor_b32 %1, %1, %3" : "+s"(a), "+s"(b) : "s"(c), "s"(d) : "scc");
                v = __builtin_fmaf(v, 1.0001f, 0.5f);
                v = __builtin_fmaf(v, 0.9999f, 0.25f);
            }
        } else if (MODE == 2) { // 16 x (s_cmp + taken-or-not uniform branch) 
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if ((a >> j) & 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(b) :: "scc");
                else asm volatile("s_add_u32 %0, %0, 5" : "+s"(c) :: "scc");
            }
            a += i;
        } else if (MODE == 3) { // 64-bit mask ops as in the compositing loop: v_cmp -> s_and -> s_andn2 -> s_or -> v_cndmask
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned long long m = __ballot(v > 1.0f);
                unsigned long long m2 = m & 0x5555555555555555ull;
                unsigned long long m3 = m2 & ~(unsigned long long)a;
                float t; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(t) : "v"(v), "v"(v * 0.5f), "s"(m3));
                v = t + 1.0f;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + (int)v;
}

template <int MODE>
double run(int waves_per_simd, int iters, int ops) {
    const int blocks = 256 * 4 * waves_per_simd;
    int* out; (void)hipMalloc(&out, sizeof(int) * blocks * 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return (double)waves_per_simd * 4 * iters * ops / (ms * 1e-3);   // instr per CU per second
}

int main() {
    const int iters = 4000; setvbuf(stdout, nullptr, _IONBF, 0);
    const char* names[] = {"64 SALU", "32 SALU + 32 VALU", "16 x (shift,and,cmp,branch,add)", "8 x mask chain (cmp,2 s_and,cndmask,mul,add)"};
    const int ops[] = {64, 64, 80, 48};
    printf("mode  waves/SIMD  Ginstr/s per CU  (cycles per instr per CU at 2.4 GHz)\n");
    for (int w : {1, 2, 4, 8}) {
        double r[4] = {run<0>(w, iters, ops[0]), run<1>(w, iters, ops[1]), run<2>(w, iters, ops[2]), run<3>(w, iters, ops[3])};
        for (int m = 0; m < 4; ++m) printf("%-48s %d  %8.3f  (%.2f)\n", names[m], w, r[m] / 1e9, 2.4e9 / r[m]);
    }
    return 0;
}
