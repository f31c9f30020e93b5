// Micro-benchmark: what the LDS access patterns of ssim.hip (and two reference patterns) cost on gfx950, in shader cycles
// per wave-instruction measured IN the kernel (s_memtime), and -- under `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT
// SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS` -- what the bank-conflict counter reports for each of them.  Round 2 left open
// whether the 44-50 % "bank conflict" share of the SSIM kernels' LDS-active cycles is real (re-layout worth ~10 us) or
// how the counter charges 16-byte accesses: every pattern below runs in a kernel of its own (its own dispatch in the
// counter output), so the counter can be read per pattern and set against the measured cycles.
//
// Patterns (SsimTile<32,32>: patch 42 x 42, patch row stride SP = 44 floats, row-pass output stride SH = 36 floats):
//   0 ds_read_b32   lane * 4 B                 (conflict-free reference)
//   1 ds_read_b64   lane * 8 B                 (conflict-free reference)
//   2 ds_read_b128  lane * 16 B                (conflict-free reference)
//   3 ds_read_b128  ssim row pass:     item i = tid -> (py = i % 42, gx = i / 42), address (py * 44 + 4 gx + 4 j) floats
//   4 ds_read_b32   ssim column pass:  ((ty0 + j) * 36 + tx) floats, tx = tid % 32, ty0 = 4 (tid / 32)
//   5 ds_write_b128 lane * 16 B                (conflict-free reference)
//   6 ds_write_b128 ssim row-pass output: (py * 36 + 4 gx) floats
//   7 ds_write_b32  ssim staging:      (spy * 44 + spx) floats, spy = tid / 42, spx = tid % 42
//   8 ds_read_b128  lane * 256 B               (every lane the same four banks: the worst case, for scale)
//   9 ds_read_b32   pseudo-random dword        (what data-dependent counters / scatters look like: the per-tile sort)
// Build: hipcc --offload-arch=gfx950 -O3 lds_patterns.hip -o lds_patterns ; run: ./lds_patterns [waves_per_block=4]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

constexpr int kLdsFloats = 16384;          // 64 KB
constexpr int kUnroll = 16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) float s[kLdsFloats];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kLdsFloats; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    // byte offset of this lane's access for step j of the unrolled group
    auto addr = [&](int j) -> int {
        if (MODE == 0) return 4 * lane + 256 * j;
        if (MODE == 1) return 8 * lane + 512 * j;
        if (MODE == 2 || MODE == 5) return 16 * lane + 1024 * j;
        if (MODE == 3) { const int i = tid % 336, py = i % 42, gx = i / 42; return 4 * (py * 44 + 4 * gx + 4 * (j & 3)) + 8192 * (j >> 2); }
        if (MODE == 4) { const int tx = tid & 31, ty0 = 4 * (tid >> 5); return 4 * ((ty0 + (j % 14)) * 36 + tx) + 8192 * (j / 14); }
        if (MODE == 6) { const int i = tid % 336, py = i % 42, gx = i / 42; return 4 * (py * 36 + 4 * gx) + 8192 * (j & 3); }
        if (MODE == 7) { const int spy = tid / 42, spx = tid % 42; return 4 * ((spy + 6 * (j % 7)) * 44 + spx); }
        if (MODE == 8) return (256 * lane) & (4 * kLdsFloats - 1);
        return 4 * ((unsigned)(tid * 2654435761u + j * 40503u) >> 18 & (kLdsFloats - 1));
    };
    constexpr int kAlign = (MODE == 0 || MODE == 4 || MODE == 7 || MODE == 9) ? 4 : (MODE == 1 ? 8 : 16);
    int a[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) a[j] = addr(j) & (4 * kLdsFloats - kAlign);
    float acc = 0.f;
    const char* base = reinterpret_cast<const char*>(s);
    char* wbase = reinterpret_cast<char*>(s);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2 || MODE == 3 || MODE == 8) {
            // sixteen 16-byte reads in flight, one wait (inline asm: the compiler must not merge or reorder them)
            f4 v[kUnroll];
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a[j]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) acc += v[j].x;
        } else if constexpr (MODE == 5 || MODE == 6) {
            const f4 v = {acc, 1.f, 2.f, 3.f};
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) asm volatile("ds_write_b128 %0, %1" :: "v"(a[j]), "v"(v) : "memory");
        } else if constexpr (MODE == 1) {
            f2 v[kUnroll];
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(a[j]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) acc += v[j].x;
        } else if constexpr (MODE == 7) {
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) asm volatile("ds_write_b32 %0, %1" :: "v"(a[j]), "v"(acc) : "memory");
        } else {
            float v[kUnroll];
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"(a[j]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) acc += v[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    (void)base; (void)wbase;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = acc + s[tid];
    if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves, float* out, unsigned long long* stamps, int n_blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(n_blocks), dim3(64 * waves), 0, 0, out, stamps, 50);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(n_blocks), dim3(64 * waves), 0, 0, out, stamps, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(n_blocks);
    hipMemcpy(h.data(), stamps, n_blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[n_blocks / 2];
    // per CU: `waves` waves issue kUnroll instructions per iteration each; one workgroup per CU
    printf("%-52s %7.2f cycles per wave-instruction per CU (%d waves: %6.2f cycles of CU time each)\n", name,
           cyc / ((double)iters * kUnroll), waves, cyc / ((double)iters * kUnroll * waves));
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4;
    const int n_blocks = 256;                      // one workgroup per CU (64 KB of LDS each)
    float* out; unsigned long long* stamps;
    hipMalloc(&out, n_blocks * 256 * sizeof(float));
    hipMalloc(&stamps, n_blocks * sizeof(unsigned long long));
    printf("lds_patterns: %d waves per workgroup, one workgroup per CU\n", waves);
    run<0>("0 ds_read_b32   conflict-free", waves, out, stamps, n_blocks);
    run<1>("1 ds_read_b64   conflict-free", waves, out, stamps, n_blocks);
    run<2>("2 ds_read_b128  conflict-free", waves, out, stamps, n_blocks);
    run<3>("3 ds_read_b128  ssim row pass (stride 44 floats)", waves, out, stamps, n_blocks);
    run<4>("4 ds_read_b32   ssim column pass (stride 36 floats)", waves, out, stamps, n_blocks);
    run<5>("5 ds_write_b128 conflict-free", waves, out, stamps, n_blocks);
    run<6>("6 ds_write_b128 ssim row-pass output (stride 36)", waves, out, stamps, n_blocks);
    run<7>("7 ds_write_b32  ssim staging (42-wide rows, stride 44)", waves, out, stamps, n_blocks);
    run<8>("8 ds_read_b128  every lane the same banks", waves, out, stamps, n_blocks);
    run<9>("9 ds_read_b32   pseudo-random dwords", waves, out, stamps, n_blocks);
    return 0;
}
