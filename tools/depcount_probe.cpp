// Probe for the next tuning round: can a per-row-tile dependency counter replace the launch boundary between two LSTM layers?
// One grid of 2 x 256 workgroups at one workgroup per CU (98 KB of LDS, like rc_gemm_kernel). Producers (blocks 0..255) do
// ~T_us of work, publish 1 KB each, release (agent scope) and bump the counter of their row tile (32 producers per tile).
// Consumers (blocks 256..511) wait -- BOUNDED spin -- for their row tile's counter, acquire, read the 32 published blocks and
// check them. Compared with the same work as two dependent launches. Every spin is bounded: a wrong assumption about
// dispatch order shows up as a timeout count, not as a hang.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_depcount depcount_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void work(float* sink, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 123.456f) *sink = a;
}

__global__ __launch_bounds__(256, 1) void k_fused(float* pub, unsigned* cnt, unsigned gen, int iters, long long* stamps, unsigned* fails, float* sink) {
    __shared__ float big[24000];                       // 96 KB: one workgroup per CU
    big[threadIdx.x] = 0.f;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) stamps[b * 3 + 0] = wall_clock64();
    if (b < 256) {
        work(sink, iters);
        pub[b * 256 + threadIdx.x] = (float)(gen * 1000 + b);            // what the next layer would read (h of this tile)
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                               // release, agent scope
            __hip_atomic_fetch_add(&cnt[b / 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            stamps[b * 3 + 1] = wall_clock64();
        }
    } else {
        const int rt = (b - 256) / 32;
        __shared__ int ok;
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(&cnt[rt], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 32u * gen && spins < 2000000) { __builtin_amdgcn_s_sleep(2); ++spins; }
            ok = spins < 2000000;
            __threadfence();                                               // acquire
            stamps[b * 3 + 1] = wall_clock64();
            if (!ok) atomicAdd(&fails[0], 1u);
        }
        __syncthreads();
        // read the 32 published blocks of the row tile; count stale values
        unsigned stale = 0;
        for (int p = 0; p < 32; ++p) {
            const float v = __builtin_nontemporal_load(&pub[(rt * 32 + p) * 256 + threadIdx.x]);
            if (v != (float)(gen * 1000 + rt * 32 + p)) ++stale;
        }
        if (stale) atomicAdd(&fails[1], stale);
        work(sink, iters);
    }
    if (threadIdx.x == 0) stamps[b * 3 + 2] = wall_clock64();
    if (big[(threadIdx.x * 7) % 24000] == 42.f) *sink = 1.f;
}

__global__ __launch_bounds__(256, 1) void k_half(float* pub, unsigned gen, int iters, int consumer, unsigned* fails, float* sink) {
    __shared__ float big[24000];
    big[threadIdx.x] = 0.f;
    const int b = blockIdx.x;
    if (!consumer) {
        work(sink, iters);
        pub[b * 256 + threadIdx.x] = (float)(gen * 1000 + b);
    } else {
        const int rt = b / 32;
        unsigned stale = 0;
        for (int p = 0; p < 32; ++p) if (pub[(rt * 32 + p) * 256 + threadIdx.x] != (float)(gen * 1000 + rt * 32 + p)) ++stale;
        if (stale) atomicAdd(&fails[1], stale);
        work(sink, iters);
    }
    if (big[(threadIdx.x * 7) % 24000] == 42.f) *sink = 1.f;
}

int main() {
    float *pub, *sink; unsigned *cnt, *fails; long long* stamps;
    CK(hipMalloc(&pub, 256 * 256 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&cnt, 8 * 4)); CK(hipMalloc(&fails, 8)); CK(hipMalloc(&stamps, 512 * 3 * 8));
    CK(hipMemset(cnt, 0, 32)); CK(hipMemset(fails, 0, 8)); CK(hipMemset(pub, 0, 256 * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 50;
    for (int iters : {2000, 20000, 60000}) {
        float ms_sep = 0, ms_fused = 0;
        unsigned gen = 0;
        for (int warm = 0; warm < 2; ++warm) {
            CK(hipEventRecord(a, 0));
            for (int r = 0; r < reps; ++r) {
                ++gen;
                hipLaunchKernelGGL(k_half, dim3(256), dim3(256), 0, 0, pub, gen, iters, 0, fails, sink);
                hipLaunchKernelGGL(k_half, dim3(256), dim3(256), 0, 0, pub, gen, iters, 1, fails, sink);
            }
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_sep, a, b));
        }
        CK(hipMemset(cnt, 0, 32));
        unsigned g2 = 0;
        for (int warm = 0; warm < 2; ++warm) {
            CK(hipEventRecord(a, 0));
            for (int r = 0; r < reps; ++r) {
                ++g2;   // counters accumulate: tile ready when cnt >= 32 * generation
                hipLaunchKernelGGL(k_fused, dim3(512), dim3(256), 0, 0, pub, cnt, g2, iters, stamps, fails, sink);
            }
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_fused, a, b));
        }
        unsigned f[2]; CK(hipMemcpy(f, fails, 8, hipMemcpyDeviceToHost));
        std::vector<long long> st(512 * 3); CK(hipMemcpy(st.data(), stamps, 512 * 3 * 8, hipMemcpyDeviceToHost));
        long long t0 = st[0]; for (int i = 0; i < 512; ++i) t0 = st[i * 3] < t0 ? st[i * 3] : t0;
        double prod_end = 0, cons_start_min = 1e18, cons_ready = 0, end = 0, wait_sum = 0;
        for (int i = 0; i < 256; ++i) prod_end = (st[i * 3 + 2] - t0) > prod_end ? (st[i * 3 + 2] - t0) : prod_end;
        for (int i = 256; i < 512; ++i) {
            cons_start_min = (st[i * 3] - t0) < cons_start_min ? (st[i * 3] - t0) : cons_start_min;
            cons_ready = (st[i * 3 + 1] - t0) > cons_ready ? (st[i * 3 + 1] - t0) : cons_ready;
            end = (st[i * 3 + 2] - t0) > end ? (st[i * 3 + 2] - t0) : end;
            wait_sum += (double)(st[i * 3 + 1] - st[i * 3]);
        }
        printf("work iters %6d: two launches %7.2f us | one grid + counters %7.2f us | timeouts %u stale %u | last producer end %.1f, first consumer start %.1f, last consumer ready %.1f, end %.1f us, mean consumer wait %.2f us\n",
               iters, ms_sep * 1e3 / reps, ms_fused * 1e3 / reps, f[0], f[1], prod_end * 0.01, cons_start_min * 0.01, cons_ready * 0.01, end * 0.01, wait_sum / 256 * 0.01);
    }
    return 0;
}
