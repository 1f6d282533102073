// micro-benchmark: cost of a grid-wide barrier on gfx950 (cooperative groups vs a hand-rolled atomic barrier),
// compared with the cost of a dependent empty kernel launch. Build: hipcc --offload-arch=gfx950 -O3 -o probe_gridsync gridsync_probe.cpp
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <chrono>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1; }

__global__ void k_cg(int n, float* buf) {
    cg::grid_group g = cg::this_grid();
    float v = 0.f;
    for (int i = 0; i < n; ++i) {
        buf[blockIdx.x * 256 + threadIdx.x] = v + i;      // a write other workgroups read after the barrier
        g.sync();
        v += buf[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
    }
    if (v == -1.f) buf[0] = v;
}

// sense-reversing barrier: one device-scope atomic per workgroup + spin on a generation word
__global__ void k_atomic(int n, float* buf, unsigned* bar) {
    float v = 0.f;
    const unsigned nb = gridDim.x;
    for (int i = 0; i < n; ++i) {
        buf[blockIdx.x * 256 + threadIdx.x] = v + i;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                   // release: make this workgroup's writes visible device-wide
            const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nb - 1) {
                __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&bar[1], gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
            }
            __threadfence();
        }
        __syncthreads();
        v += __builtin_nontemporal_load(&buf[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x]);
    }
    if (v == -1.f) buf[0] = v;
}

int main(int argc, char** argv) {
    const int n = 200;
    float* buf; unsigned* bar;
    CK(hipMalloc(&buf, 4096 * 256 * 4)); CK(hipMalloc(&bar, 8)); CK(hipMemset(bar, 0, 8)); CK(hipMemset(buf, 0, 4096 * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    // dependent empty launches
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, nullptr);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("empty dependent launch: %.2f us each\n", ms * 1e3 / n);
    for (int grid : {256, 512, 1024}) {
        int nn = n; void* args[] = {&nn, &buf};
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a, 0));
            hipError_t e = hipLaunchCooperativeKernel((void*)k_cg, dim3(grid), dim3(256), args, 0, 0);
            if (e != hipSuccess) { printf("cooperative launch grid %d: %s\n", grid, hipGetErrorString(e)); (void)hipGetLastError(); ms = -1; break; }
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        if (ms >= 0) printf("cg grid.sync   grid %4d: %.2f us per barrier\n", grid, ms * 1e3 / n);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(bar, 0, 8));
            CK(hipEventRecord(a, 0));
            void* args2[] = {&nn, &buf, &bar};
            hipError_t e = hipLaunchCooperativeKernel((void*)k_atomic, dim3(grid), dim3(256), args2, 0, 0);
            if (e != hipSuccess) { printf("cooperative launch (atomic) grid %d: %s\n", grid, hipGetErrorString(e)); (void)hipGetLastError(); ms = -1; break; }
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        if (ms >= 0) printf("atomic barrier grid %4d: %.2f us per barrier\n", grid, ms * 1e3 / n);
    }
    return 0;
}
