// Micro-probe of a CU's vector-memory path next to the matrix pipe (gfx950), one wave per SIMD like the wide gate-GEMM tiles:
//   loads   : 1-KiB global_load_dwordx4 wave-instructions per cycle from an L1- / L2-resident window, nothing else
//   lds_dma : the same bytes through global_load_lds_dwordx4
//   mix<N>  : N v_mfma_f32_16x16x32_bf16 per load, loads as above -- does the load stream hide behind the MFMAs?
//   mixv<N> : the same plus 2 VALU per MFMA (the operand split's density)
// hipcc --offload-arch=gfx950 -O3 -o l1_probe tools/l1_probe.cpp && ./l1_probe      (results: profiles/r02_l1_probe.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// window: bytes per wave that the loads cycle through (1 KiB steps). 4 waves x 4 KiB: L1 hits; 4 x 16 KiB x 32 CUs = 2 MiB per
// XCD: L2 hits; 4 x 128 KiB x 256 CUs = 128 MiB: Infinity-Cache hits. The loads and their waits are inline asm (DEPTH loads in
// flight per wave, s_waitcnt vmcnt(DEPTH - 1) in front of the use of the oldest): hipcc's own waits would be vmcnt(0) here.
#define STR_(x) #x
#define STR(x) STR_(x)
typedef float f32x16 __attribute__((ext_vector_type(16)));
// the same with v_mfma_f32_32x32x16_bf16 (twice the MACs and cycles per instruction: half the MFMA issue slots per MAC)
template <int MFMA_PER_LOAD, int VALU_PER_MFMA, int DEPTH, int NWAVE = 4>
__global__ __launch_bounds__(64 * NWAVE) void k_mix32(const u32x4* __restrict__ src, long long window_u4, int iters, unsigned long long* cyc, float* sink) {
    __shared__ float pad[24 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = src + ((long long)blockIdx.x * NWAVE + wave) * window_u4 + lane;
    u32x4 buf[DEPTH ? DEPTH : 1];
#pragma unroll
    for (int u = 0; u < (DEPTH ? DEPTH : 1); ++u) buf[u] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float v[4] = {1.f, 2.f, 3.f, 4.f};
    long long off = 0;
    if (DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[u]) : "v"(p) : "memory");
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < (DEPTH ? DEPTH : 8); ++u) {
            const int ub = DEPTH ? u : 0;
            if (DEPTH) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(buf[ub]) : "n"(DEPTH ? DEPTH - 1 : 0));
#pragma unroll
            for (int m = 0; m < MFMA_PER_LOAD; ++m) {
                acc[(m + u) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, buf[ub]), __builtin_bit_cast(bf16x8, buf[ub]), acc[(m + u) & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < VALU_PER_MFMA; ++q) v[q & 3] = v[q & 3] * 1.0001f + 0.5f;
            }
            if (DEPTH) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[ub]) : "v"(p + off), "v"(acc[(MFMA_PER_LOAD - 1 + u) & 3]) : "memory");
                off += 64;
                if (off >= window_u4) off = 0;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][1];
#pragma unroll
    for (int u = 0; u < (DEPTH ? DEPTH : 1); ++u) s += __uint_as_float(buf[u][1] & 0x3fffffffu);
    if (s == 12345.678f) sink[0] = s + pad[lane];
    if (lane == 0) cyc[blockIdx.x * NWAVE + wave] = t1 - t0;
}

template <int MFMA_PER_LOAD, int VALU_PER_MFMA, int DEPTH, int NWAVE = 4>
__global__ __launch_bounds__(64 * NWAVE) void k_mix(const u32x4* __restrict__ src, long long window_u4, int iters, unsigned long long* cyc, float* sink) {
    __shared__ float pad[24 * 1024];                   // 96 KiB: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = src + ((long long)blockIdx.x * NWAVE + wave) * window_u4 + lane;
    u32x4 buf[DEPTH ? DEPTH : 1];
#pragma unroll
    for (int u = 0; u < (DEPTH ? DEPTH : 1); ++u) buf[u] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    f32x4 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[4] = {1.f, 2.f, 3.f, 4.f};
    long long off = 0;
    if (DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[u]) : "v"(p) : "memory");
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < (DEPTH ? DEPTH : 8); ++u) {
            const int ub = DEPTH ? u : 0;
            if (DEPTH) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(buf[ub]) : "n"(DEPTH ? DEPTH - 1 : 0));
            if (MFMA_PER_LOAD == 0) acc[u & 7][0] += __uint_as_float(buf[ub][0] & 0x3fffffffu);
#pragma unroll
            for (int m = 0; m < MFMA_PER_LOAD; ++m) {
                acc[(m + u) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, buf[ub]), __builtin_bit_cast(bf16x8, buf[ub]), acc[(m + u) & 7], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < VALU_PER_MFMA; ++q) v[q & 3] = v[q & 3] * 1.0001f + 0.5f;
            }
            if (DEPTH) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(buf[ub]) : "v"(p + off), "v"(acc[(MFMA_PER_LOAD - 1 + u) & 7]) : "memory");
                off += 64;
                if (off >= window_u4) off = 0;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1];
#pragma unroll
    for (int u = 0; u < (DEPTH ? DEPTH : 1); ++u) s += __uint_as_float(buf[u][1] & 0x3fffffffu);
    if (s == 12345.678f) sink[0] = s + pad[lane];
    if (lane == 0) cyc[blockIdx.x * NWAVE + wave] = t1 - t0;
}

// the same stream through the LDS: global_load_lds_dwordx4 into a ring of DEPTH 1-KiB slots per wave, one ds_read_b128 per slot,
// MFMAs on the fragment read one step earlier
template <int MFMA_PER_LOAD, int VALU_PER_MFMA, int DEPTH>
__global__ __launch_bounds__(256) void k_dma(const u32x4* __restrict__ src, long long window_u4, int iters, unsigned long long* cyc, float* sink) {
    __shared__ u32x4 stage[4][DEPTH][64];              // DEPTH KiB per wave
    __shared__ float pad[8 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = src + ((long long)blockIdx.x * 4 + wave) * window_u4 + lane;
    f32x4 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[4] = {1.f, 2.f, 3.f, 4.f};
    u32x4 x[2] = {u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}};
    long long off = 0;
#define DMA(SLOT, SRC) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), \
                                                        (__attribute__((address_space(3))) void*)(&stage[wave][SLOT][0]), 16, 0, 0)
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) DMA(u, p);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[(u + 1) & 1]));        // the previous step's fragment is in, its slot is free
            DMA((u + DEPTH - 1) % DEPTH, p + off);
            off += 64;
            if (off >= window_u4) off = 0;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");     // slot u has landed
            const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&stage[wave][u][lane]);
            asm volatile("ds_read_b128 %0, %1" : "=v"(x[u & 1]) : "v"(la) : "memory");
            const u32x4 xp = x[(u + 1) & 1];
            if (MFMA_PER_LOAD == 0) acc[u & 7][0] += __uint_as_float(xp[0] & 0x3fffffffu);
#pragma unroll
            for (int m = 0; m < MFMA_PER_LOAD; ++m) {
                acc[(m + u) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xp), __builtin_bit_cast(bf16x8, xp), acc[(m + u) & 7], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < VALU_PER_MFMA; ++q) v[q & 3] = v[q & 3] * 1.0001f + 0.5f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]));
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1];
    s += __uint_as_float((x[0][1] ^ x[1][1]) & 0x3fffffffu);
    if (s == 12345.678f) sink[0] = s + pad[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static hipDeviceProp_t prop;
static u32x4* d_src;
static unsigned long long* d_cyc;
static float* d_sink;
static int n_wg;
static int n_wave = 4;

template <typename F>
static void run(const char* name, F launch, int iters, double loads_per_iter, double mfma_per_iter) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();                                           // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(n_wg * n_wave);
    CK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto x : c) avg += (double)x;
    avg /= c.size();
    const double cyc_it = avg / iters;
    printf("%-34s %8.3f ms  %9.0f cyc/wave", name, ms, avg);
    if (loads_per_iter > 0) printf("  %6.1f cyc per 1-KiB load and wave = %5.1f B/clk/CU", cyc_it / loads_per_iter, n_wave * 1024.0 * loads_per_iter / cyc_it);
    if (mfma_per_iter > 0) printf("  %5.1f cyc/MFMA and SIMD", cyc_it / mfma_per_iter / (n_wave / 4));
    printf("  (clock %.2f GHz)\n", avg / (ms * 1e6));
}

int main() {
    CK(hipGetDeviceProperties(&prop, 0));
    n_wg = prop.multiProcessorCount;
    const long long max_window = 256 * 1024;            // bytes per wave
    CK(hipMalloc(&d_src, (size_t)n_wg * 8 * max_window));
    CK(hipMemset(d_src, 0x3f, (size_t)n_wg * 8 * max_window));
    CK(hipMalloc(&d_cyc, n_wg * 8 * 8));
    CK(hipMalloc(&d_sink, 64));
    printf("%s, %d CUs; one 256-thread workgroup per CU (one wave per SIMD)\n", prop.name, n_wg);
    const int iters = 1000;
    char nm[128];
#define VG(N, V, D, WIN)                                                                                                     \
    snprintf(nm, sizeof nm, "vgpr %2d in flight, %d MFMA+%dv/load, %3lldK", D, N, V, (long long)(WIN) / 1024);                 \
    run(nm, [&] { hipLaunchKernelGGL((k_mix<N, V, D>), dim3(n_wg), dim3(256), 0, 0, d_src, (long long)(WIN) / 16, iters, d_cyc, d_sink); }, iters, D, (double)D * N);
#define LD(N, V, D, WIN)                                                                                                     \
    snprintf(nm, sizeof nm, "lds  %2d in flight, %d MFMA+%dv/load, %3lldK", D, N, V, (long long)(WIN) / 1024);                 \
    run(nm, [&] { hipLaunchKernelGGL((k_dma<N, V, D>), dim3(n_wg), dim3(256), 0, 0, d_src, (long long)(WIN) / 16, iters, d_cyc, d_sink); }, iters, D, (double)D * N);
    run("mfma only", [&] { hipLaunchKernelGGL((k_mix<5, 0, 0>), dim3(n_wg), dim3(256), 0, 0, d_src, 256ll, iters, d_cyc, d_sink); }, iters, 0, 40);
    run("mfma + 2 VALU each", [&] { hipLaunchKernelGGL((k_mix<5, 2, 0>), dim3(n_wg), dim3(256), 0, 0, d_src, 256ll, iters, d_cyc, d_sink); }, iters, 0, 40);
    for (long long win : {4096ll, 16384ll, 131072ll}) {
        VG(0, 0, 8, win) VG(0, 0, 24, win) LD(0, 0, 8, win) LD(0, 0, 24, win)
        VG(5, 0, 24, win) LD(5, 0, 24, win) VG(5, 2, 24, win) LD(5, 2, 24, win) VG(8, 0, 24, win) LD(8, 0, 24, win)
    }
    // 32x32x16 MFMAs (32 cycles each): the same filler densities per MAC are half as many per instruction gap
    printf("-- v_mfma_f32_32x32x16_bf16, one wave per SIMD (cycles per MFMA: floor 32)\n");
#define VG32(N, V, D, WIN)                                                                                                   \
    snprintf(nm, sizeof nm, "mfma32 %2d in flight, %d MFMA+%dv/load, %3lldK", D, N, V, (long long)(WIN) / 1024);               \
    run(nm, [&] { hipLaunchKernelGGL((k_mix32<N, V, D>), dim3(n_wg), dim3(256), 0, 0, d_src, (long long)(WIN) / 16, iters, d_cyc, d_sink); }, iters, D, (double)(D ? D : 8) * N);
    VG32(4, 0, 0, 16384) VG32(4, 2, 0, 16384) VG32(4, 3, 0, 16384) VG32(4, 4, 0, 16384)
    VG32(2, 0, 24, 16384) VG32(3, 0, 24, 16384) VG32(2, 2, 24, 16384) VG32(3, 2, 24, 16384) VG32(3, 3, 24, 16384) VG32(2, 3, 24, 16384) VG32(2, 4, 24, 16384)
    // two waves per SIMD (512-thread workgroups, 256 VGPRs per wave): does the second wave's issue hide behind the first's MFMAs?
    n_wave = 8;
    printf("-- two waves per SIMD\n");
#define VG8(N, V, D, WIN)                                                                                                    \
    snprintf(nm, sizeof nm, "vgpr8 %2d in flight, %d MFMA+%dv/load, %3lldK", D, N, V, (long long)(WIN) / 1024);                \
    run(nm, [&] { hipLaunchKernelGGL((k_mix<N, V, D, 8>), dim3(n_wg), dim3(512), 0, 0, d_src, (long long)(WIN) / 16, iters, d_cyc, d_sink); }, iters, D, (double)D * N);
    run("mfma only", [&] { hipLaunchKernelGGL((k_mix<5, 0, 0, 8>), dim3(n_wg), dim3(512), 0, 0, d_src, 256ll, iters, d_cyc, d_sink); }, iters, 0, 40);
    run("mfma + 2 VALU each", [&] { hipLaunchKernelGGL((k_mix<5, 2, 0, 8>), dim3(n_wg), dim3(512), 0, 0, d_src, 256ll, iters, d_cyc, d_sink); }, iters, 0, 40);
    for (long long win : {4096ll, 16384ll, 131072ll}) {
        VG8(0, 0, 8, win) VG8(0, 0, 24, win) VG8(5, 0, 8, win) VG8(5, 2, 8, win) VG8(5, 0, 24, win) VG8(5, 2, 24, win) VG8(8, 2, 24, win)
    }
    return 0;
}
