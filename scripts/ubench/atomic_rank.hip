// Micro-benchmark for a binning variant: every (Gaussian, tile) entry takes its rank inside the tile with a RETURNING
// global atomic on a per-tile counter, then a second pass scatters the entries to offsets[tile] + rank.  Question: what
// do 3.3 M returning atomics on 8 160 counters cost (counters packed, or padded to one per 64 / 128 bytes), and what
// does the scatter cost, against two stable radix passes (2 x 36 us) + the offsets pass (9 us) of the tile-sort
// pipeline?
// Build: hipcc --offload-arch=gfx950 -O3 atomic_rank.hip -o atomic_rank
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one thread per Gaussian, as the emit kernel: its tiles are a small rectangle, written to consecutive slots
template <bool RETURNING>
__global__ void __launch_bounds__(256) rank_kernel(int n, const int* __restrict__ rect, const int* __restrict__ off, int tile_w,
                                                   int pad, int* __restrict__ cnt, unsigned* __restrict__ keys) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int4 r = reinterpret_cast<const int4*>(rect)[g];          // x0, y0, w, h
    int o = off[g];
    for (int dy = 0; dy < r.w; ++dy)
        for (int dx = 0; dx < r.z; ++dx) {
            const int t = (r.y + dy) * tile_w + r.x + dx;
            if (RETURNING) {
                const int rk = atomicAdd(&cnt[(size_t)t * pad], 1);
                keys[o++] = ((unsigned)t << 16) | (unsigned)(rk & 0xffff);
            } else {
                atomicAdd(&cnt[(size_t)t * pad], 1);
                keys[o++] = (unsigned)t << 16;
            }
        }
}

__global__ void __launch_bounds__(256) scatter_kernel(int m, const unsigned* __restrict__ keys, const int* __restrict__ gid,
                                                      const int* __restrict__ tile_off, int* __restrict__ out_gid,
                                                      float* __restrict__ out_depth) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const unsigned k = keys[i];
    const int pos = tile_off[k >> 16] + (int)(k & 0xffff);
    const int g = gid[i];
    out_gid[pos] = g;
    out_depth[pos] = (float)g * 1e-3f;
}

int main() {
    const int n = 500000, tile_w = 120, tile_h = 68, n_tiles = tile_w * tile_h;
    std::vector<int> rect(4 * n), off(n + 1);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    int m = 0;
    for (int g = 0; g < n; ++g) {
        const int w = 1 + rnd() % 3, h = 1 + rnd() % 4;              // 1..3 x 1..4 tiles: 5 on average (config B: 6.7)
        const int w2 = w + (rnd() % 3 == 0), h2 = h + (rnd() % 3 == 0);
        rect[4 * g] = rnd() % (tile_w - w2 + 1); rect[4 * g + 1] = rnd() % (tile_h - h2 + 1);
        rect[4 * g + 2] = w2; rect[4 * g + 3] = h2;
        off[g] = m; m += w2 * h2;
    }
    off[n] = m;
    std::vector<int> gid(m);
    for (int g = 0; g < n; ++g) for (int i = off[g]; i < off[g + 1]; ++i) gid[i] = g;
    printf("N=%d tiles=%d entries=%d (%.2f per Gaussian, %.0f per tile)\n", n, n_tiles, m, (double)m / n, (double)m / n_tiles);
    int *d_rect, *d_off, *d_cnt, *d_gid, *d_tile_off, *d_out_gid;
    unsigned* d_keys;
    float* d_out_depth;
    CHECK(hipMalloc(&d_rect, rect.size() * 4)); CHECK(hipMalloc(&d_off, off.size() * 4));
    CHECK(hipMalloc(&d_cnt, (size_t)n_tiles * 32 * 4)); CHECK(hipMalloc(&d_gid, (size_t)m * 4));
    CHECK(hipMalloc(&d_keys, (size_t)m * 4)); CHECK(hipMalloc(&d_tile_off, (size_t)(n_tiles + 1) * 4));
    CHECK(hipMalloc(&d_out_gid, (size_t)m * 4)); CHECK(hipMalloc(&d_out_depth, (size_t)m * 4));
    CHECK(hipMemcpy(d_rect, rect.data(), rect.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_gid, gid.data(), (size_t)m * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 20;
    for (int returning = 0; returning < 2; ++returning)
        for (int pad : {1, 16, 32}) {
            std::vector<float> ts;
            for (int it = 0; it < reps; ++it) {
                CHECK(hipMemsetAsync(d_cnt, 0, (size_t)n_tiles * 32 * 4, 0));
                CHECK(hipEventRecord(e0, 0));
                if (returning) hipLaunchKernelGGL(rank_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_rect, d_off, tile_w, pad, d_cnt, d_keys);
                else hipLaunchKernelGGL(rank_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_rect, d_off, tile_w, pad, d_cnt, d_keys);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            printf("rank pass, %s atomics, counter stride %3d B: median %.1f us, min %.1f us\n", returning ? "returning" : "plain    ",
                   pad * 4, ts[reps / 2], ts[0]);
        }
    // tile offsets from the last (returning, pad 32) counters, on the host; ranks < 65536 by construction here
    std::vector<int> cnt((size_t)n_tiles * 32), tile_off(n_tiles + 1);
    CHECK(hipMemcpy(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
    int acc = 0;
    for (int t = 0; t < n_tiles; ++t) { tile_off[t] = acc; acc += cnt[(size_t)t * 32]; }
    tile_off[n_tiles] = acc;
    if (acc != m) { printf("count mismatch %d vs %d\n", acc, m); return 1; }
    CHECK(hipMemcpy(d_tile_off, tile_off.data(), tile_off.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ts;
    for (int it = 0; it < reps; ++it) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(scatter_kernel, dim3((m + 255) / 256), dim3(256), 0, 0, m, d_keys, d_gid, d_tile_off, d_out_gid, d_out_depth);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("scatter pass (2 x 4 B per entry to offsets[tile] + rank): median %.1f us, min %.1f us\n", ts[reps / 2], ts[0]);
    // every output slot written exactly once?
    std::vector<int> out(m);
    CHECK(hipMemcpy(out.data(), d_out_gid, (size_t)m * 4, hipMemcpyDeviceToHost));
    std::vector<int> back(out);
    std::sort(back.begin(), back.end());
    std::vector<int> ref(gid);
    std::sort(ref.begin(), ref.end());
    printf("scatter is a permutation: %s\n", back == ref ? "yes" : "NO");
    return 0;
}
