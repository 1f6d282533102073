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

// XCD-hierarchical barrier (MI355X_MICROARCH.md price list, row "barrier-xcd"): arrivals are counted per XCC (the
// members of one XCD share its L2), the LAST arriver of an XCD is its leader: it alone writes the XCD's L2 back
// (release fence), arrives on the top counter, and the last leader bumps every XCC's generation word; everybody else
// polls its own XCC's generation (one relaxed sc1 load + s_sleep) and ends with an agent-scope acquire (L1 invalidate).
// Membership per XCC is counted at run time (placement is not promised). Every spin is bounded: a timeout sets bar[ERR].
#define W(i) ((i) * 32)          // one 128-byte line per word
enum { CNT = 0, GEN = 8, TOP = 16, MEM = 17, INIT = 25, ERR = 26, NWORDS = 27 };
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ unsigned ld(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool spin_until_changed(unsigned* p, unsigned old, unsigned* err) {
    for (int it = 0; it < (1 << 22); ++it) {
        if (ld(p) != old) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    st(err, 1u);
    return false;
}
__global__ void k_xcd(int n, float* buf, unsigned* bar) {
    float v = 0.f;
    const unsigned nb = gridDim.x;
    __shared__ unsigned s_x, s_m, s_nx;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        __hip_atomic_fetch_add(&bar[W(MEM + x)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&bar[W(INIT)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int it = 0; it < (1 << 22) && ld(&bar[W(INIT)]) < nb; ++it) __builtin_amdgcn_s_sleep(1);
        if (ld(&bar[W(INIT)]) < nb) st(&bar[W(ERR)], 1u);
        unsigned nx = 0;
        for (int y = 0; y < 8; ++y) nx += ld(&bar[W(MEM + y)]) ? 1u : 0u;
        s_x = x; s_m = ld(&bar[W(MEM + x)]); s_nx = nx;
    }
    __syncthreads();
    const unsigned x = s_x, m = s_m, nx = s_nx;
    for (int i = 0; i < n; ++i) {
        buf[blockIdx.x * 256 + threadIdx.x] = (float)i;   // a stale read of the previous round changes the sum
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's stores have reached the XCD's L2
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned gen = ld(&bar[W(GEN + x)]);
            if (ld(&bar[W(ERR)])) break;
            if (__hip_atomic_fetch_add(&bar[W(CNT + x)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == m - 1) {
                st(&bar[W(CNT + x)], 0u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // leader: write this XCD's L2 back once
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (__hip_atomic_fetch_add(&bar[W(TOP)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nx - 1) {
                    st(&bar[W(TOP)], 0u);
                    for (int y = 0; y < 8; ++y) st(&bar[W(GEN + y)], gen + 1);
                } else {
                    spin_until_changed(&bar[W(GEN + x)], gen, &bar[W(ERR)]);
                }
            } else {
                spin_until_changed(&bar[W(GEN + x)], gen, &bar[W(ERR)]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        v += buf[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
    }
    if (v == -1.f) buf[0] = v;
    // correctness witness: after n barriers block 0 has added 0 + 1 + ... + (n - 1), its neighbour's round numbers
    if (threadIdx.x == 0 && blockIdx.x == 0) buf[4096 * 256 - 1] = v;
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
        unsigned* xbar;
        CK(hipMalloc(&xbar, NWORDS * 128));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(xbar, 0, NWORDS * 128));
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(k_xcd, dim3(grid), dim3(256), 0, 0, nn, buf, xbar);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        unsigned hb[NWORDS * 32]; float wit = 0.f;
        CK(hipMemcpy(hb, xbar, sizeof(hb), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&wit, buf + 4096 * 256 - 1, 4, hipMemcpyDeviceToHost));
        // expected witness: every round adds the neighbour's freshly written round number
        float ev = 0.f; for (int i = 0; i < n; ++i) ev += (float)i;
        printf("xcd barrier    grid %4d: %.2f us per barrier (incl. one %d-block membership pass; members/XCC", grid, ms * 1e3 / n, grid);
        for (int y = 0; y < 8; ++y) printf(" %u", hb[W(MEM + y)]);
        printf("; timeout flag %u; witness %s)\n", hb[W(ERR)], wit == ev ? "ok" : "MISMATCH");
        CK(hipFree(xbar));
    }
    return 0;
}
