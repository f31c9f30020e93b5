// Micro-benchmark: cycles per wave64 VALU instruction per SIMD on gfx950, counted IN the kernel with s_memtime
// (shader cycles) so that no clock assumption enters, plus the clock itself from s_memrealtime (100 MHz).
// Answers: does a SIMD issue a wave64 VALU op every 2 or every 4 cycles when several waves are resident, and
// what do packed / transcendental / readlane ops cost?
// Build: hipcc --offload-arch=gfx950 -O3 valu_cycles.hip -o valu_cycles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* stamps, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 64 fma, 8 independent chains
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
            }
        } else if (MODE == 1) {  // 64 fma, one dependent chain
#pragma unroll
            for (int j = 0; j < 64; ++j) a0 = __builtin_fmaf(a0, b, c);
        } else if (MODE == 2) {  // 64 v_exp_f32, 8 independent
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
            }
        } else if (MODE == 3) {  // 64 v_pk_fma_f32, 4 independent float2 chains
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
            const f2 bb = {b, b}, cc = {c, c};
#pragma unroll
            for (int j = 0; j < 16; ++j) { p0 = p0 * bb + cc; p1 = p1 * bb + cc; p2 = p2 * bb + cc; p3 = p3 * bb + cc; }
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (MODE == 4) {  // 16 x (v_cmp -> SGPR mask, v_cndmask, add) x2 = 96 VALU
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned long long m = __ballot(a0 > a1);
                float d; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a2), "v"(a3), "s"(m));
                a0 = a0 + d;
                unsigned long long m2 = __ballot(a4 > a5);
                float e; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(e) : "v"(a6), "v"(a7), "s"(m2));
                a4 = a4 + e;
            }
        } else if (MODE == 5) {  // 8 x (2 v_readlane + 4 fma with the SGPRs) = 48 VALU
            const int t = i & 63;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), t));
                float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), t));
                a2 = __builtin_fmaf(a2, s0, c); a3 = __builtin_fmaf(a3, s1, c);
                a4 = __builtin_fmaf(a4, s0, c); a5 = __builtin_fmaf(a5, s1, c);
            }
        } else if (MODE == 6) {  // 64 DPP adds (row_ror), 8 independent
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#define DPPADD(x) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false))
                DPPADD(a0); DPPADD(a1); DPPADD(a2); DPPADD(a3); DPPADD(a4); DPPADD(a5); DPPADD(a6); DPPADD(a7);
            }
        } else if (MODE == 7) {  // 32 v_permlane32_swap + 32 add
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
                a0 = __uint_as_float(r[0]) + c; a1 = __uint_as_float(r[1]) + c;
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(a3), false, false);
                a2 = __uint_as_float(r[0]) + c; a3 = __uint_as_float(r[1]) + c;
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a4), __float_as_uint(a5), false, false);
                a4 = __uint_as_float(r[0]) + c; a5 = __uint_as_float(r[1]) + c;
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a6), __float_as_uint(a7), false, false);
                a6 = __uint_as_float(r[0]) + c; a7 = __uint_as_float(r[1]) + c;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

struct Res { double cyc_per_instr_simd, ghz, wall_cyc_per_instr; };

template <int MODE>
Res run(int waves_per_simd, int iters, int ops_per_iter) {
    const int blocks = 256 * 4 * waves_per_simd;   // one wave per block: exactly fills the chip once
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 64);
    unsigned long long* st; (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, 10, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> cyc(blocks), clk(blocks);
    for (int i = 0; i < blocks; ++i) { cyc[i] = (double)h[2 * i]; clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    (void)hipFree(out); (void)hipFree(st);
    const double instr_per_wave = (double)iters * ops_per_iter;
    Res r;
    // a SIMD hosts waves_per_simd waves that each took cyc cycles for instr_per_wave instructions
    r.cyc_per_instr_simd = cyc[blocks / 2] / (instr_per_wave * waves_per_simd);
    r.ghz = clk[blocks / 2];
    r.wall_cyc_per_instr = (ms * 1e-3) * r.ghz * 1e9 / (instr_per_wave * waves_per_simd);
    return r;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int iters = 20000;
    const char* names[] = {"v_fma x8 indep", "v_fma dependent", "v_exp x8 indep", "v_pk_fma x4 indep", "cmp+cndmask+add",
                           "2 readlane + 4 fma", "v_add_dpp x8 indep", "permlane32_swap + 2 add"};
    const int ops[] = {64, 64, 64, 64, 96, 48, 64, 96};
    printf("%-26s waves/SIMD  cycles per wave-instr per SIMD (in-kernel)  clock GHz  (from wall time)\n", "mode");
    for (int w : {1, 2, 3, 4, 8}) {
        Res r[8] = {run<0>(w, iters, ops[0]), run<1>(w, iters, ops[1]), run<2>(w, iters, ops[2]), run<3>(w, iters, ops[3]),
                    run<4>(w, iters, ops[4]), run<5>(w, iters, ops[5]), run<6>(w, iters, ops[6]), run<7>(w, iters, ops[7])};
        for (int m = 0; m < 8; ++m)
            printf("%-26s %d  %7.3f  %6.3f  (%7.3f)\n", names[m], w, r[m].cyc_per_instr_simd, r[m].ghz, r[m].wall_cyc_per_instr);
    }
    return 0;
}
