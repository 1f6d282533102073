// Micro-probe (gfx950), round 6: the K loop of the split-bf16 gate GEMM with the WEIGHTS shared by the workgroup.
//   w256 : tile 256 rows x (16 NC) columns, 4 waves, wave w owns rows 64 w .. 64 w + 63 and ALL columns (M split, no K split):
//          the weight planes of a k-block are staged ONCE per workgroup in a 3-deep LDS ring by global_load_lds_dwordx4 and
//          read back with ds_read_b128 (one column block ahead of its MFMAs); every wave streams its own 64 rows of fp32
//          activations straight into VGPRs (two buffers) and splits the NEXT k-block's activations between the MFMAs of the
//          current one. The K range is still summed as FOUR chains (quarters, the K split of the product kernel's four waves),
//          added in the order of that kernel's LDS reduction: per element the arithmetic is gemm_tile's, bit for bit (SUM = 1).
// One workgroup per CU; A (256 x K fp32, rc_pk order) is shared by every workgroup (L2 / Infinity Cache resident, as in a tick:
// all column tiles of a layer step read the same activations), the weight planes are a stream of their own per workgroup (HBM).
// hipcc --offload-arch=gfx950 -O3 -o lds256_probe tools/lds256_probe.cpp && ./lds256_probe     (results: profiles/r06_lds256_probe.txt)
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

#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

// one float2 pair of the operand split (rc_gemm.hip: split3): a = hi + mid + lo exactly, each a truncated bf16
__device__ __forceinline__ void split_pair(const f32x2 a, unsigned& h, unsigned& m, unsigned& l) {
    const u32x2 ua = __builtin_bit_cast(u32x2, a);
    const f32x2 r1 = a - __builtin_bit_cast(f32x2, ua & 0xffff0000u);
    const u32x2 um = __builtin_bit_cast(u32x2, r1);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, um & 0xffff0000u);
    const u32x2 ul = __builtin_bit_cast(u32x2, r2);
    h = __builtin_amdgcn_perm(ua[1], ua[0], 0x07060302u);
    m = __builtin_amdgcn_perm(um[1], um[0], 0x07060302u);
    l = __builtin_amdgcn_perm(ul[1], ul[0], 0x07060302u);
}
// half of a row block's split: pairs 2 * HALF and 2 * HALF + 1 of (x0, x1)
template <int HALF>
__device__ __forceinline__ void split_half(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
    const f32x4& x = HALF ? x1 : x0;
    unsigned hh, mm, ll;
    split_pair(f32x2{x[0], x[1]}, hh, mm, ll); h[2 * HALF] = hh; m[2 * HALF] = mm; l[2 * HALF] = ll;
    split_pair(f32x2{x[2], x[3]}, hh, mm, ll); h[2 * HALF + 1] = hh; m[2 * HALF + 1] = mm; l[2 * HALF + 1] = ll;
}

__device__ __forceinline__ void gload(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload1k(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(d) : "v"(p) : "memory"); }
template <int OFF>
__device__ __forceinline__ void glds(const u32x4* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void dsread(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }

template <int WR, int NC, bool SUM, bool SPLITPIPE>
__global__ __launch_bounds__(1024 / WR, 1) void k_w256(const float* __restrict__ A, long long a_rb_stride, const u32x4* __restrict__ W, int Qs, int n_wrap,
                                                 float* out, unsigned long long* cyc) {
    constexpr int NW = 16 / WR;                // waves: wave w owns rows 16 WR w .. 16 WR (w + 1) - 1
    constexpr int PW = NC * 3 / NW;            // glds pieces (1 KiB) per wave and k-block
    constexpr int NA = 2 * WR;                 // activation loads per wave and k-block
    constexpr int STAGE = NC * 3 * 1024;       // bytes of one k-block of weight planes: [column block][plane][lane] x 16 B
    __shared__ __attribute__((aligned(1024))) unsigned char ring[3 * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ring0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)ring;
    const int n_tile = blockIdx.x % n_wrap;

    const float* pa[WR];
#pragma unroll
    for (int r = 0; r < WR; ++r) pa[r] = A + (long long)(wave * WR + r) * a_rb_stride + lane * 4;
    // this wave's pieces of a k-block: NC = 8: column blocks 2 w, 2 w + 1 (three planes each); NC = 4: column block w
    constexpr int CBW = NC / NW;               // column blocks per wave
    const u32x4* pw[CBW];
#pragma unroll
    for (int c = 0; c < CBW; ++c) pw[c] = W + ((long long)(n_tile * NC + wave * CBW + c) * Qs) * 192 + lane;
    const unsigned my_lds = ring0 + (unsigned)(wave * CBW) * 3072u;   // + stage * STAGE + (c * 3 + plane) * 1024

    f32x4 acc[WR][NC];
    float sum[WR][NC][4];                       // the quarters' running sum lives in accumulation registers: touched by asm with "a" constraints only
#pragma unroll
    for (int r = 0; r < WR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(sum[r][j][e]));
        }
    f32x4 raw[2][WR][2];
    u32x4 pl[2][WR][3];
    u32x4 bf[2][3];

#define ISSUE_A(BUF, QI)                                                                                  \
    do {                                                                                                  \
        const long long ko_ = (long long)min((QI), Qs - 1) * 512;                                         \
        _Pragma("unroll") for (int r = 0; r < WR; ++r) { gload(raw[BUF][r][0], pa[r] + ko_); gload1k(raw[BUF][r][1], pa[r] + ko_); }   \
    } while (0)
#define ISSUE_B(QI)                                                                                       \
    do {                                                                                                  \
        const int qq_ = min((QI), Qs - 1);                                                                \
        const unsigned dst_ = my_lds + (unsigned)((QI) % 3) * STAGE;                                      \
        _Pragma("unroll") for (int c = 0; c < CBW; ++c) {                                                 \
            const u32x4* g_ = pw[c] + (long long)qq_ * 192;                                               \
            glds<0>(g_, dst_ + c * 3072); glds<1024>(g_, dst_ + c * 3072); glds<2048>(g_, dst_ + c * 3072);   \
        }                                                                                                 \
    } while (0)
#define WAIT_RAW(BUF, N)                                                                                  \
    do { if constexpr (WR == 4)                                                                           \
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(raw[BUF][0][0]), "+v"(raw[BUF][0][1]), "+v"(raw[BUF][1][0]), "+v"(raw[BUF][1][1]),   \
                     "+v"(raw[BUF][WR - 2][0]), "+v"(raw[BUF][WR - 2][1]), "+v"(raw[BUF][WR - 1][0]), "+v"(raw[BUF][WR - 1][1]) : "n"(N) : "memory");  \
      else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(raw[BUF][0][0]), "+v"(raw[BUF][0][1]), "+v"(raw[BUF][1][0]), "+v"(raw[BUF][1][1]) : "n"(N) : "memory"); } while (0)
#define SPLIT_ALL(BUF)                                                                                    \
    do { _Pragma("unroll") for (int r = 0; r < WR; ++r) {                                                  \
            split_half<0>(raw[BUF][r][0], raw[BUF][r][1], pl[BUF][r][0], pl[BUF][r][1], pl[BUF][r][2]);   \
            split_half<1>(raw[BUF][r][0], raw[BUF][r][1], pl[BUF][r][0], pl[BUF][r][1], pl[BUF][r][2]); } } while (0)

    // ---- prologue
    ISSUE_A(0, 0); ISSUE_B(0); ISSUE_A(1, 1); ISSUE_B(1);
    WAIT_RAW(0, NA + PW);
    SPLIT_ALL(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
        const unsigned a_ = ring0 + lane * 16;
        dsread<0>(bf[0][0], a_); dsread<1024>(bf[0][1], a_); dsread<2048>(bf[0][2], a_);
    }
    const int Qq = Qs / 4;
    int qnext = Qq;
    const unsigned long long t0 = __builtin_readcyclecounter();

#define SLOT(CUR, J, Q)                                                                                                     \
    do {                                                                                                                    \
        constexpr int bi_ = (J) & 1;                                                                                        \
        if ((J) == NC - 1) {                                                                                                \
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(NA + PW) : "memory");             \
            const unsigned a_ = ring0 + (unsigned)(((Q) + 1) % 3) * STAGE + lane * 16;                                      \
            dsread<0>(bf[bi_ ^ 1][0], a_); dsread<1024>(bf[bi_ ^ 1][1], a_); dsread<2048>(bf[bi_ ^ 1][2], a_);              \
        } else {                                                                                                            \
            const unsigned a_ = ring0 + (unsigned)((Q) % 3) * STAGE + lane * 16;                                            \
            dsread<((J) + 1) * 3072>(bf[bi_ ^ 1][0], a_); dsread<((J) + 1) * 3072 + 1024>(bf[bi_ ^ 1][1], a_);              \
            dsread<((J) + 1) * 3072 + 2048>(bf[bi_ ^ 1][2], a_);                                                            \
        }                                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[bi_][0]), "+v"(bf[bi_][1]), "+v"(bf[bi_][2]) :: "memory");            \
        if (SPLITPIPE) {    /* the next k-block's activations: row block J / (NC / 4) ... */                                \
            constexpr int per_ = NC / WR;      /* slots per row block */                                                    \
            constexpr int rr_ = (J) / per_, ph_ = (J) % per_;                                                               \
            if (per_ == 4) { if (ph_ == 0) split_half<0>(raw[CUR ^ 1][rr_][0], raw[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][0], pl[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][2]);  \
                             if (ph_ == 2) split_half<1>(raw[CUR ^ 1][rr_][0], raw[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][0], pl[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][2]); }  \
            else if (per_ == 1) { split_half<0>(raw[CUR ^ 1][rr_][0], raw[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][0], pl[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][2]);   \
                             split_half<1>(raw[CUR ^ 1][rr_][0], raw[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][0], pl[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][2]); }  \
            else if (ph_ == 0) split_half<0>(raw[CUR ^ 1][rr_][0], raw[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][0], pl[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][2]);  \
            else split_half<1>(raw[CUR ^ 1][rr_][0], raw[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][0], pl[CUR ^ 1][rr_][1], pl[CUR ^ 1][rr_][2]);                \
        }                                                                                                                   \
        _Pragma("unroll") for (int r = 0; r < WR; ++r) MFMA(pl[CUR][r][2], bf[bi_][0], acc[r][J]);                           \
        _Pragma("unroll") for (int r = 0; r < WR; ++r) MFMA(pl[CUR][r][0], bf[bi_][2], acc[r][J]);                           \
        _Pragma("unroll") for (int r = 0; r < WR; ++r) MFMA(pl[CUR][r][1], bf[bi_][1], acc[r][J]);                           \
        _Pragma("unroll") for (int r = 0; r < WR; ++r) MFMA(pl[CUR][r][1], bf[bi_][0], acc[r][J]);                           \
        _Pragma("unroll") for (int r = 0; r < WR; ++r) MFMA(pl[CUR][r][0], bf[bi_][1], acc[r][J]);                           \
        _Pragma("unroll") for (int r = 0; r < WR; ++r) MFMA(pl[CUR][r][0], bf[bi_][0], acc[r][J]);                           \
        if (SPLITPIPE) {    /* ... is done HERE (else hipcc sinks the split in front of the planes' first use) */           \
            constexpr int rr2_ = (J) / (NC / WR);                                                                            \
            asm volatile("" : "+v"(pl[CUR ^ 1][rr2_][0]), "+v"(pl[CUR ^ 1][rr2_][1]), "+v"(pl[CUR ^ 1][rr2_][2]));          \
        }                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)

#define ITER(CUR, Q)                                                                                                        \
    do {                                                                                                                    \
        if (SUM && (Q) == qnext) {             /* quarter boundary: the chain of the next K quarter starts from zero */      \
            qnext += Qq;                                                                                                    \
            _Pragma("unroll") for (int r = 0; r < WR; ++r) _Pragma("unroll") for (int j = 0; j < NC; ++j) {                  \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) { float t_;                                                   \
                    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_add_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1" : "+a"(sum[r][j][e]), "=&v"(t_) : "v"(acc[r][j][e])); }   \
                acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }                                                                    \
        }                                                                                                                   \
        WAIT_RAW(CUR ^ 1, PW);                 /* A(Q + 1) has landed (B(Q + 1) may still be in flight) */                  \
        if (!SPLITPIPE) { }                                                                                                 \
        ISSUE_A(CUR, (Q) + 2);                                                                                              \
        ISSUE_B((Q) + 2);                                                                                                   \
        if constexpr (NC == 8) { SLOT(CUR, 0, Q); SLOT(CUR, 1, Q); SLOT(CUR, 2, Q); SLOT(CUR, 3, Q); SLOT(CUR, 4, Q); SLOT(CUR, 5, Q); SLOT(CUR, 6, Q); SLOT(CUR, 7, Q); }   \
        else { SLOT(CUR, 0, Q); SLOT(CUR, 1, Q); SLOT(CUR, 2, Q); SLOT(CUR, 3, Q); }                                         \
        if (!SPLITPIPE) SPLIT_ALL(CUR ^ 1);                                                                                 \
    } while (0)

    for (int q = 0; q < Qs; q += 2) {
        ITER(0, q);
        ITER(1, q + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < WR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float t_; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t_) : "a"(sum[r][j][e])); s += t_ + acc[r][j][e]; }
        }
    s += raw[0][0][0][0] + raw[1][0][0][0] + __builtin_bit_cast(float, bf[0][0][0]) + __builtin_bit_cast(float, bf[1][0][0]);
    if (s == 12345.678f) out[0] = s;
    if (lane == 0 && wave < 4) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static int n_wg;

template <typename F>
static void run(const char* name, F launch, unsigned long long* d_cyc, int kblocks, int mfma_per_kblock_wave, double kb_per_kblock_wg) {
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
    printf("%-58s %8.3f ms  %7.0f cyc per k-block  %5.1f cyc/MFMA (%4.1f %% of the 16-cycle issue rate)  %5.1f B/clk/CU global  clock %.2f GHz\n",
           name, ms, cyc_kb, cyc_kb / mfma_per_kblock_wave, 1600.0 / (cyc_kb / mfma_per_kblock_wave), kb_per_kblock_wg * 1024.0 / cyc_kb, mx / (ms * 1e6));
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    n_wg = prop.multiProcessorCount;
    const int K = 2048, Qs = K / 32;                    // an rnn6 layer step: K' = 2 H = 2048
    float* dA; u32x4* dW; float* d_out; unsigned long long* d_cyc;
    CK(hipMalloc(&dA, (size_t)256 * K * 4));
    CK(hipMemset(dA, 0x3c, (size_t)256 * K * 4));
    const size_t w_bytes = (size_t)n_wg * 128 * K * 6;  // a weight stream of its own for every workgroup: 403 MB at 256 CUs
    CK(hipMalloc(&dW, w_bytes));
    CK(hipMemset(dW, 0x3c, w_bytes));
    CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_cyc, n_wg * 4 * 8));
    printf("%s, %d CUs; one 256-thread workgroup per CU; K = %d\n", prop.name, n_wg, K);
    const long long rbs = 16ll * K;
    for (int wrap : {n_wg, 8}) {                        // every workgroup its own slice (HBM stream) | 8 slices in all (L2-resident)
        printf("-- weight slices: %d (%s)\n", wrap, wrap == n_wg ? "HBM stream" : "L2-resident");
        run("w256 256x128, quarter sums, split between the MFMAs", [&] { hipLaunchKernelGGL((k_w256<4, 8, true, true>), dim3(n_wg), dim3(256), 0, 0, dA, rbs, dW, Qs, wrap, d_out, d_cyc); }, d_cyc, Qs, 192, 4 * 8.0 + 24.0);
        run("w256 256x128, quarter sums, split at the k-block's end", [&] { hipLaunchKernelGGL((k_w256<4, 8, true, false>), dim3(n_wg), dim3(256), 0, 0, dA, rbs, dW, Qs, wrap, d_out, d_cyc); }, d_cyc, Qs, 192, 4 * 8.0 + 24.0);
        run("w256 256x128, one chain, split between the MFMAs", [&] { hipLaunchKernelGGL((k_w256<4, 8, false, true>), dim3(n_wg), dim3(256), 0, 0, dA, rbs, dW, Qs, wrap, d_out, d_cyc); }, d_cyc, Qs, 192, 4 * 8.0 + 24.0);
        run("w256 8 waves x (32 x 128), quarter sums, split between", [&] { hipLaunchKernelGGL((k_w256<2, 8, true, true>), dim3(n_wg), dim3(512), 0, 0, dA, rbs, dW, Qs, wrap, d_out, d_cyc); }, d_cyc, Qs, 96, 4 * 8.0 + 24.0);
        run("w256 8 waves x (32 x 128), quarter sums, split at the end", [&] { hipLaunchKernelGGL((k_w256<2, 8, true, false>), dim3(n_wg), dim3(512), 0, 0, dA, rbs, dW, Qs, wrap, d_out, d_cyc); }, d_cyc, Qs, 96, 4 * 8.0 + 24.0);
        run("w256 8 waves x (32 x 128), one chain, split at the end", [&] { hipLaunchKernelGGL((k_w256<2, 8, false, false>), dim3(n_wg), dim3(512), 0, 0, dA, rbs, dW, Qs, wrap, d_out, d_cyc); }, d_cyc, Qs, 96, 4 * 8.0 + 24.0);
        run("w256 256x64, quarter sums, split between the MFMAs", [&] { hipLaunchKernelGGL((k_w256<4, 4, true, true>), dim3(n_wg), dim3(256), 0, 0, dA, rbs, dW, Qs, wrap * 2, d_out, d_cyc); }, d_cyc, Qs, 96, 4 * 8.0 + 12.0);
    }
    return 0;
}
