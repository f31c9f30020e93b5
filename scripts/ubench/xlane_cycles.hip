// Micro-benchmark: issue cost of the cross-lane instructions the compositing backward uses for its wave reduction
// (v_permlane32_swap, v_permlane16_swap, v_add_f32_dpp, v_mov_b32_dpp, v_readlane, ds_bpermute, ds_swizzle), each
// written as the exact instruction (inline asm), 8 independent chains, wall-clock cycles per wave-instruction per SIMD
// with the clock taken from s_memtime / s_memrealtime.
// Build: hipcc --offload-arch=gfx950 -O3 xlane_cycles.hip -o xlane_cycles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define REP8(X) X(a0, a1) X(a2, a3) X(a4, a5) X(a6, a7) X(a0, a1) X(a2, a3) X(a4, a5) X(a6, a7)
#define REP8S(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)

#define SWAP32(x, y) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(x), "+v"(y));
#define SWAP16(x, y) asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(x), "+v"(y));
#define DPPADD(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
#define DPPADD2(x, y) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(y));
#define DPPMOV(x, y) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(y));
#define BCAST15(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
#define PLAINADD(x) asm volatile("v_add_f32_e32 %0, %0, %0" : "+v"(x));
#define PLAINADD2(x, y) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "v"(y));

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, unsigned long long* stamps, int iters, float seed, const float4* recs) {
    // (MODE 13) ids: a pseudo-random record per lane, different per wave, within a 96 MB table
    const int ids = (int)(((unsigned)(blockIdx.x * 64 + threadIdx.x) * 2654435761u) >> 11);
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int addr = ((threadIdx.x + 1) & 63) * 4;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, c0 = 0.25f * threadIdx.x, c1 = 0.125f * threadIdx.x;
    __shared__ __attribute__((aligned(16))) float lds[64 * 12];
    for (int q = 0; q < 12; ++q) lds[12 * threadIdx.x + q] = 1.f + 1e-6f * (float)(q + threadIdx.x);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) { REP8(SWAP32) }                                   // 8 swaps
            else if (MODE == 1) { REP8(SWAP16) }
            else if (MODE == 2) { REP8S(DPPADD) }                             // 8 dpp adds, dst = src (dependent per register)
            else if (MODE == 3) { REP8(DPPADD2) }                             // 8 dpp adds, independent source
            else if (MODE == 4) { REP8(DPPMOV) }
            else if (MODE == 5) { REP8S(PLAINADD) }
            else if (MODE == 6) {                                             // 8 ds_bpermute
                a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a0)));
                a1 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a1)));
                a2 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a2)));
                a3 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a3)));
                a4 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a4)));
                a5 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a5)));
                a6 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a6)));
                a7 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a7)));
            } else if (MODE == 7) {                                           // 4 x (swap32, 2 adds): the reduce pattern
                SWAP32(a0, a1) SWAP32(a2, a3) SWAP32(a4, a5) SWAP32(a6, a7)
                PLAINADD2(a0, a1) PLAINADD2(a2, a3) PLAINADD2(a4, a5) PLAINADD2(a6, a7)
            } else if (MODE == 8) { REP8S(BCAST15) }
            else if (MODE == 9) {                                             // 8 v_readlane + 8 adds of the SGPR
                const int t = i & 63;
                float s;
#define RL(x) s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), t)); asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "s"(s));
                REP8S(RL)
            } else if (MODE == 10) {                                          // 10 v_readlane back to back, then 10 fma that use them
                const int t = (i + j) & 63;
                float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), t));
                float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), t));
                float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), t));
                float s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a3), t));
                float s4 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a4), t));
                float s5 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a5), t));
                float s6 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a6), t));
                float s7 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a7), t));
                float s8 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b0), t));
                float s9 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b1), t));
                asm volatile("" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7), "+s"(s8), "+s"(s9));
                c0 = __builtin_fmaf(c0, s0, s1); c1 = __builtin_fmaf(c1, s2, s3); c0 = __builtin_fmaf(c0, s4, s5);
                c1 = __builtin_fmaf(c1, s6, s7); c0 = __builtin_fmaf(c0, s8, s9); c1 = __builtin_fmaf(c1, s0, s2);
                c0 = __builtin_fmaf(c0, s1, s3); c1 = __builtin_fmaf(c1, s4, s6); c0 = __builtin_fmaf(c0, s5, s7);
                c1 = __builtin_fmaf(c1, s8, s9);
            } else if (MODE == 11) {                                          // the same operands from LDS: 3 broadcast ds_read_b128
                const int t = (i + j) & 63;
                const float4* rec = reinterpret_cast<const float4*>(lds + 12 * t);
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                c0 = __builtin_fmaf(c0, q0.x, q0.y); c1 = __builtin_fmaf(c1, q0.z, q0.w); c0 = __builtin_fmaf(c0, q1.x, q1.y);
                c1 = __builtin_fmaf(c1, q1.z, q1.w); c0 = __builtin_fmaf(c0, q2.x, q2.y); c1 = __builtin_fmaf(c1, q0.x, q0.z);
                c0 = __builtin_fmaf(c0, q0.y, q0.w); c1 = __builtin_fmaf(c1, q1.x, q1.z); c0 = __builtin_fmaf(c0, q1.y, q1.w);
                c1 = __builtin_fmaf(c1, q2.x, q2.y);
            } else if (MODE == 13) {                                          // the same operands by scalar loads (s_load_dwordx4 x 3) + 1 readlane for the id
                const int t = (i + j) & 63;
                const int id = __builtin_amdgcn_readlane(ids, t);
                typedef float v4f __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(4))) v4f* cptr;
                cptr rec = (cptr)(recs) + 3 * (size_t)id;
                const v4f q0 = rec[0], q1 = rec[1], q2 = rec[2];
                c0 = __builtin_fmaf(c0, q0.x, q0.y); c1 = __builtin_fmaf(c1, q0.z, q0.w); c0 = __builtin_fmaf(c0, q1.x, q1.y);
                c1 = __builtin_fmaf(c1, q1.z, q1.w); c0 = __builtin_fmaf(c0, q2.x, q2.y); c1 = __builtin_fmaf(c1, q0.x, q0.z);
                c0 = __builtin_fmaf(c0, q0.y, q0.w); c1 = __builtin_fmaf(c1, q1.x, q1.z); c0 = __builtin_fmaf(c0, q1.y, q1.w);
                c1 = __builtin_fmaf(c1, q2.x, q2.y);
            } else if (MODE == 12) {                                          // 10 fma only (baseline of modes 10 / 11)
                c0 = __builtin_fmaf(c0, a0, a1); c1 = __builtin_fmaf(c1, a2, a3); c0 = __builtin_fmaf(c0, a4, a5);
                c1 = __builtin_fmaf(c1, a6, a7); c0 = __builtin_fmaf(c0, b0, b1); c1 = __builtin_fmaf(c1, a0, a2);
                c0 = __builtin_fmaf(c0, a1, a3); c1 = __builtin_fmaf(c1, a4, a6); c0 = __builtin_fmaf(c0, a5, a7);
                c1 = __builtin_fmaf(c1, b0, b1);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0 + c1;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int waves_per_simd, int iters, int ops_per_iter) {
    const int blocks = 256 * 4 * waves_per_simd;
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 64);
    unsigned long long* st; (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    static float4* recs = nullptr;
    if (!recs) { (void)hipMalloc(&recs, (size_t)3 * 16 * (1u << 21)); (void)hipMemset(recs, 0, (size_t)3 * 16 * (1u << 21)); }
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, 10, 1.0f, recs);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, iters, 1.0f, recs);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> clk(blocks);
    for (int i = 0; i < blocks; ++i) clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
    std::sort(clk.begin(), clk.end());
    (void)hipFree(out); (void)hipFree(st);
    const double ghz = clk[blocks / 2];
    const double cyc = (ms * 1e-3) * ghz * 1e9 / ((double)iters * ops_per_iter * waves_per_simd);
    printf("%-44s %d  %7.2f cycles per wave-instr per SIMD  (clock %.2f GHz)\n", name, waves_per_simd, cyc, ghz);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int iters = 5000;
    for (int w : {1, 2, 4, 8}) {
        run<5>("v_add_f32 (plain, x8 indep)", w, iters, 32);
        run<0>("v_permlane32_swap x8", w, iters, 32);
        run<1>("v_permlane16_swap x8", w, iters, 32);
        run<2>("v_add_f32_dpp row_ror (dst = src)", w, iters, 32);
        run<3>("v_add_f32_dpp row_ror (separate source)", w, iters, 32);
        run<4>("v_mov_b32_dpp row_ror", w, iters, 32);
        run<8>("v_add_f32_dpp row_bcast:15", w, iters, 32);
        run<6>("ds_bpermute_b32 x8", w, iters, 32);
        run<7>("4 x (swap32 + add) = 8 instr", w, iters, 32);
        run<9>("8 x (v_readlane + v_add sgpr) = 16 instr", w, iters, 64);
        run<12>("10 fma (baseline), per group of 10", w, iters, 4);
        run<10>("10 v_readlane + 10 fma, per group", w, iters, 4);
        run<11>("3 ds_read_b128 broadcast + 10 fma, per group", w, iters, 4);
        run<13>("readlane id + 3 s_load_dwordx4 + 10 fma, per group", w, iters, 4);
    }
    return 0;
}
