// Micro-probe (gfx950): the K loop of the split-bf16 gate GEMM in two constructions, one 256-thread workgroup per CU, operands from an
// L2-resident window, no epilogue -- cycles per k-block (32 k) and per v_mfma_f32_16x16x32_bf16:
//   cur   : the product kernel's 64 x 128 tile: K split over the 4 waves, every wave streams its own operands straight into VGPRs
//           (8 activation + 24 weight-plane 1-KiB loads, 144 VALU of operand split, 192 MFMAs per k-block and wave; two buffers)
//   lds   : a 128 x 128 tile, waves as 2 x 2 quadrants of 64 x 64, NO K split: the workgroup stages one k-block (activations split into
//           planes by the waves, 2 row blocks each; weight planes by global_load_lds_dwordx4) in a 3-deep LDS ring; every wave reads
//           its 4 + 4 blocks of planes (24 ds_read_b128) for 96 MFMAs -- 40 KB of global operands per 384 MFMAs instead of 128 KB
//   ldsp  : the same with activations already stored as planes by their producer (no split in the loop)
// hipcc --offload-arch=gfx950 -O3 -o lds_gemm_probe tools/lds_gemm_probe.cpp && ./lds_gemm_probe     (results: profiles/r03_lds_gemm_probe.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
    const f32x2 a[4] = {f32x2{x0[0], x0[1]}, f32x2{x0[2], x0[3]}, f32x2{x1[0], x1[1]}, f32x2{x1[2], x1[3]}};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const u32x2 ua = __builtin_bit_cast(u32x2, a[d]);
        const f32x2 r1 = a[d] - __builtin_bit_cast(f32x2, ua & 0xffff0000u);
        const u32x2 um = __builtin_bit_cast(u32x2, r1);
        const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, um & 0xffff0000u);
        const u32x2 ul = __builtin_bit_cast(u32x2, r2);
        h[d] = __builtin_amdgcn_perm(ua[1], ua[0], 0x07060302u);
        m[d] = __builtin_amdgcn_perm(um[1], um[0], 0x07060302u);
        l[d] = __builtin_amdgcn_perm(ul[1], ul[0], 0x07060302u);
    }
}

#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

// ------------------------------------------------------------------------------------------------ current construction
template <int MR, int NC>
struct Frag { f32x4 a0[MR], a1[MR]; u32x4 b[NC][3]; };

template <int MR, int NC>
__device__ __forceinline__ void load_frag(Frag<MR, NC>& f, const u32x4* g, long long off, long long win) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        f.a0[r] = __builtin_bit_cast(f32x4, g[(off + (2 * r) * 64) & (win - 1)]);
        f.a1[r] = __builtin_bit_cast(f32x4, g[(off + (2 * r + 1) * 64) & (win - 1)]);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) f.b[j][p] = g[(off + (2 * MR + 3 * j + p) * 64) & (win - 1)];
}
template <int MR, int NC>
__device__ __forceinline__ void mma_frag(const Frag<MR, NC>& f, f32x4 (&acc)[MR][NC]) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        u32x4 h, m, l;
        split3(f.a0[r], f.a1[r], h, m, l);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(l, f.b[j][0], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(h, f.b[j][2], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(m, f.b[j][1], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(m, f.b[j][0], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(h, f.b[j][1], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(h, f.b[j][0], acc[r][j]);
    }
}

template <int MR, int NC>
__global__ __launch_bounds__(256, 1) void k_cur(const u32x4* __restrict__ src, long long win_u4, int kblocks, unsigned long long* cyc, float* sink) {
    __shared__ float pad[30 * 1024];                   // 120 KiB: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* g = src + ((long long)blockIdx.x * 4 + wave) * win_u4 + lane;
    const long long step = (2 * MR + 3 * NC) * 64;
    f32x4 acc[MR][NC];
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag<MR, NC> fa = {}, fb = {};
    long long off = 0;
    load_frag<MR, NC>(fa, g, off, win_u4); off = (off + step) & (win_u4 - 1);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q + 2 <= kblocks; q += 2) {
        load_frag<MR, NC>(fb, g, off, win_u4); off = (off + step) & (win_u4 - 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag<MR, NC>(fa, acc);
        __builtin_amdgcn_sched_barrier(0);
        load_frag<MR, NC>(fa, g, off, win_u4); off = (off + step) & (win_u4 - 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag<MR, NC>(fb, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) s += acc[r][j][0] + acc[r][j][1] + acc[r][j][2] + acc[r][j][3];
    s += fa.a0[0][0];
    if (s == 12345.678f) sink[0] = s + pad[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// the current construction with the WEIGHTS streamed as fp32 too (4 B instead of 6 B per weight) and split in the loop, once per k-block
template <int MR, int NC>
struct FragF { f32x4 a0[MR], a1[MR], b0[NC], b1[NC]; };
template <int MR, int NC>
__device__ __forceinline__ void load_fragf(FragF<MR, NC>& f, const u32x4* g, long long off, long long win) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        f.a0[r] = __builtin_bit_cast(f32x4, g[(off + (2 * r) * 64) & (win - 1)]);
        f.a1[r] = __builtin_bit_cast(f32x4, g[(off + (2 * r + 1) * 64) & (win - 1)]);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        f.b0[j] = __builtin_bit_cast(f32x4, g[(off + (2 * MR + 2 * j) * 64) & (win - 1)]);
        f.b1[j] = __builtin_bit_cast(f32x4, g[(off + (2 * MR + 2 * j + 1) * 64) & (win - 1)]);
    }
}
template <int MR, int NC>
__device__ __forceinline__ void mma_fragf(const FragF<MR, NC>& f, f32x4 (&acc)[MR][NC]) {
    u32x4 bh[NC], bm[NC], bl[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) split3(f.b0[j], f.b1[j], bh[j], bm[j], bl[j]);
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        u32x4 h, m, l;
        split3(f.a0[r], f.a1[r], h, m, l);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(l, bh[j], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(h, bl[j], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(m, bm[j], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(m, bh[j], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(h, bm[j], acc[r][j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) MFMA(h, bh[j], acc[r][j]);
    }
}
template <int MR, int NC, int NBUF>
__global__ __launch_bounds__(256, 1) void k_curf(const u32x4* __restrict__ src, long long win_u4, int kblocks, unsigned long long* cyc, float* sink) {
    __shared__ float pad[30 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* g = src + ((long long)blockIdx.x * 4 + wave) * win_u4 + lane;
    const long long step = (2 * MR + 2 * NC) * 64;
    f32x4 acc[MR][NC];
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    FragF<MR, NC> fa = {}, fb = {}, fc = {};
    long long off = 0;
#define LD_(F) do { load_fragf<MR, NC>(F, g, off, win_u4); off = (off + step) & (win_u4 - 1); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MM_(F) do { mma_fragf<MR, NC>(F, acc); __builtin_amdgcn_sched_barrier(0); } while (0)
    LD_(fa);
    if (NBUF == 3) LD_(fb);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (NBUF == 3) {
        for (int q = 0; q + 3 <= kblocks; q += 3) { LD_(fc); MM_(fa); LD_(fa); MM_(fb); LD_(fb); MM_(fc); }
    } else {
        for (int q = 0; q + 2 <= kblocks; q += 2) { LD_(fb); MM_(fa); LD_(fa); MM_(fb); }
    }
#undef LD_
#undef MM_
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) s += acc[r][j][0] + acc[r][j][1] + acc[r][j][2] + acc[r][j][3];
    s += fa.a0[0][0] + fb.a0[0][0] + fc.a0[0][0];
    if (s == 12345.678f) sink[0] = s + pad[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ LDS-staged construction
#define STAGE_U4 3072                                  // per stage: A planes [8 row blocks][3][64] + B planes [8 column blocks][3][64] u32x4 = 48 KiB
#define DMA(LDSPTR, GPTR) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(GPTR), \
                                                           (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

template <bool PRESPLIT>
__global__ __launch_bounds__(256, 1) void k_lds(const u32x4* __restrict__ src, long long win_u4, int kblocks, unsigned long long* cyc, float* sink) {
    __shared__ u32x4 lds[3 * STAGE_U4];                // 144 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    const u32x4* g = src + (long long)blockIdx.x * win_u4 + lane;
    // global operand bytes of one k-block and workgroup: B 24 pieces of 1 KiB, A 16 pieces (fp32) or 24 (planes)
    const long long step = (24 + (PRESPLIT ? 24 : 16)) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 areg[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) { areg[r][0] = f32x4{1.f, 1.f, 1.f, 1.f}; areg[r][1] = areg[r][0]; }
    long long off = 0;
    // fills of one k-block into stage s: every wave's share (weight planes 6 pieces; activations 4 fp32 pieces into registers, or 6 plane pieces)
#define FILL(S)                                                                                                          \
    do {                                                                                                                 \
        if (!PRESPLIT) {                                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                              \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg[r][0]) : "v"(g + ((off + (24 + wave * 4 + 2 * r) * 64) & (win_u4 - 1))) : "memory");      \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg[r][1]) : "v"(g + ((off + (24 + wave * 4 + 2 * r + 1) * 64) & (win_u4 - 1))) : "memory");  \
            }                                                                                                            \
        } else {                                                                                                         \
            u32x4* sa_ = lds + (S) * STAGE_U4;                                                                           \
            _Pragma("unroll") for (int p = 0; p < 6; ++p) DMA(sa_ + (wave * 6 + p) * 64, g + ((off + (24 + wave * 6 + p) * 64) & (win_u4 - 1)));  \
        }                                                                                                                \
        u32x4* sb_ = lds + (S) * STAGE_U4 + 1536;                                                                        \
        _Pragma("unroll") for (int p = 0; p < 6; ++p) DMA(sb_ + (wave * 6 + p) * 64, g + ((off + (wave * 6 + p) * 64) & (win_u4 - 1)));   \
        off = (off + step) & (win_u4 - 1);                                                                               \
    } while (0)
    // the wave's two row blocks of the activations in registers -> planes in stage s
#define WRITE_A(S)                                                                                                       \
    do {                                                                                                                 \
        u32x4* sa = lds + (S) * STAGE_U4;                                                                                \
        _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                                  \
            u32x4 h, m, l;                                                                                               \
            split3(areg[r][0], areg[r][1], h, m, l);                                                                     \
            sa[((wave * 2 + r) * 3 + 0) * 64 + lane] = h;                                                                \
            sa[((wave * 2 + r) * 3 + 1) * 64 + lane] = m;                                                                \
            sa[((wave * 2 + r) * 3 + 2) * 64 + lane] = l;                                                                \
        }                                                                                                                \
    } while (0)
    constexpr int PER = PRESPLIT ? 12 : 10;            // vector-memory operations of one FILL and wave
    FILL(0);
    if (!PRESPLIT) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(areg[0][0]), "+v"(areg[0][1]), "+v"(areg[1][0]), "+v"(areg[1][1]) :: "memory");
        WRITE_A(0);
    }
    FILL(1);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < kblocks; ++q) {
        const int s = q % 3, s1 = (q + 1) % 3, s2 = (q + 2) % 3;
        if (PER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // block q has landed (the newest FILL may still be in flight)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // not __syncthreads(): its fence waits vmcnt(0), i.e. for the prefetched blocks too
        if (!PRESPLIT) {                               // block q + 1's activations (requested one iteration ago, in front of its weight pieces)
            asm volatile("s_waitcnt vmcnt(6)" : "+v"(areg[0][0]), "+v"(areg[0][1]), "+v"(areg[1][0]), "+v"(areg[1][1]) :: "memory");
            WRITE_A(s1);
        }
        FILL(s2);                                      // block q + 2: its stage was last read in block q - 1
        const u32x4* sa = lds + s * STAGE_U4 + (wr * 4) * 192 + lane;
        const u32x4* sb = lds + s * STAGE_U4 + 1536 + (wc * 4) * 192 + lane;
        u32x4 b[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = sb[(j * 3 + p) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u32x4 h = sa[(r * 3 + 0) * 64], m = sa[(r * 3 + 1) * 64], l = sa[(r * 3 + 2) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) MFMA(l, b[j][0], acc[r][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) MFMA(h, b[j][2], acc[r][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) MFMA(m, b[j][1], acc[r][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) MFMA(m, b[j][0], acc[r][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) MFMA(h, b[j][1], acc[r][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) MFMA(h, b[j][0], acc[r][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[r][j][0] + acc[r][j][1] + acc[r][j][2] + acc[r][j][3];
    s += areg[0][0][0] + areg[1][1][1];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static int n_wg;
static u32x4* d_src;
static unsigned long long* d_cyc;
static float* d_sink;

template <typename F>
static void run(const char* name, F launch, int kblocks, int mfma_per_kblock_wave, double kb_per_kblock_wg) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(n_wg * 4);
    CK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0, mx = 0;
    for (auto x : c) { avg += (double)x; if ((double)x > mx) mx = (double)x; }
    avg /= c.size();
    const double cyc_kb = avg / kblocks;
    printf("%-44s %8.3f ms  %8.0f cyc per k-block and wave  %5.1f cyc/MFMA (%4.1f %% of the 16-cycle issue rate)  %5.1f B/clk/CU of global operands  clock %.2f GHz\n",
           name, ms, cyc_kb, cyc_kb / mfma_per_kblock_wave, 1600.0 / (cyc_kb / mfma_per_kblock_wave), kb_per_kblock_wg * 1024.0 / cyc_kb, mx / (ms * 1e6));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    n_wg = prop.multiProcessorCount;
    const long long max_win = 4ll << 20;               // bytes per workgroup
    CK(hipMalloc(&d_src, (size_t)n_wg * max_win));
    CK(hipMemset(d_src, 0x3f, (size_t)n_wg * max_win));
    CK(hipMalloc(&d_cyc, n_wg * 4 * 8));
    CK(hipMalloc(&d_sink, 64));
    printf("%s, %d CUs; one 256-thread workgroup per CU\n", prop.name, n_wg);
    const int kb = 960;                                // k-blocks per launch (a multiple of 2 and 3)
    for (long long win : {64ll << 10, 512ll << 10, 4ll << 20}) {       // per workgroup: L2-resident (16 MiB in all), Infinity Cache (128 MiB), HBM (1 GiB)
        char nm[128];
        printf("-- operand window %lld KiB per workgroup\n", win >> 10);
        snprintf(nm, sizeof nm, "cur 64x128 (K split, straight to VGPRs)");
        run(nm, [&] { hipLaunchKernelGGL((k_cur<4, 8>), dim3(n_wg), dim3(256), 0, 0, d_src, win / 16 / 4, kb / 4, d_cyc, d_sink); }, kb / 4, 192, 4 * 32.0);
        snprintf(nm, sizeof nm, "cur 64x80");
        run(nm, [&] { hipLaunchKernelGGL((k_cur<4, 5>), dim3(n_wg), dim3(256), 0, 0, d_src, win / 16 / 4, kb / 4, d_cyc, d_sink); }, kb / 4, 120, 4 * 23.0);
        snprintf(nm, sizeof nm, "cur 64x128, fp32 weights split in the loop");
        run(nm, [&] { hipLaunchKernelGGL((k_curf<4, 8, 2>), dim3(n_wg), dim3(256), 0, 0, d_src, win / 16 / 4, kb / 4, d_cyc, d_sink); }, kb / 4, 192, 4 * 24.0);
        snprintf(nm, sizeof nm, "cur 64x80, fp32 weights, 3 buffers");
        run(nm, [&] { hipLaunchKernelGGL((k_curf<4, 5, 3>), dim3(n_wg), dim3(256), 0, 0, d_src, win / 16 / 4, kb / 4, d_cyc, d_sink); }, kb / 4, 120, 4 * 18.0);
        snprintf(nm, sizeof nm, "cur 64x80, fp32 weights, 2 buffers");
        run(nm, [&] { hipLaunchKernelGGL((k_curf<4, 5, 2>), dim3(n_wg), dim3(256), 0, 0, d_src, win / 16 / 4, kb / 4, d_cyc, d_sink); }, kb / 4, 120, 4 * 18.0);
        snprintf(nm, sizeof nm, "lds 128x128, split in the loop");
        run(nm, [&] { hipLaunchKernelGGL((k_lds<false>), dim3(n_wg), dim3(256), 0, 0, d_src, win / 16, kb, d_cyc, d_sink); }, kb, 96, 40.0);
        snprintf(nm, sizeof nm, "lds 128x128, activations stored as planes");
        run(nm, [&] { hipLaunchKernelGGL((k_lds<true>), dim3(n_wg), dim3(256), 0, 0, d_src, win / 16, kb, d_cyc, d_sink); }, kb, 96, 48.0);
    }
    return 0;
}
