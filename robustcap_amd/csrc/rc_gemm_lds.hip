// Gate GEMM of the LSTM layer steps with the WEIGHTS SHARED BY THE WORKGROUP (round 6) -- gfx950 only.
//
// Replaces, for contexts of >= RC_LDS_MIN_ROWS rows in split-product mode, the 64-row tiles of rc_gemm.hip on the twelve LSTM
// layer steps of a frame (net/sig_mp.py:126-129 -> articulate/utils/torch/rnn.py:129-133, aten::lstm; 96 % of the path's FLOPs).
//
// Why. rc_gemm.hip splits K over the four waves of a 64 x 128 tile; nothing is shared inside a workgroup, so every weight byte is
// pulled through FOUR CUs' vector-memory paths (the four row tiles of a batch of 256) and each wave issues its MFMAs, its operand
// split and 32 global loads per k-block in order: 39 cycles per MFMA against a 16-cycle issue rate. Here one
// workgroup owns ALL rows of a 128-column slice (up to 256: one row tile of the batch):
//   * 8 waves (two per SIMD), wave w = rows 32 w .. 32 w + 31 of the tile's compacted row list x all 128 columns: M split, no K
//     split inside the workgroup;
//   * the weight planes of a k-block (24 KiB) are staged ONCE per workgroup in a 4-deep LDS ring by global_load_lds_dwordx4
//     (each wave moves one column block's three planes) and read back with ds_read_b128 one column block ahead of its MFMAs;
//   * every wave streams its own 32 rows of fp32 activations straight into VGPRs (two buffers, two k-blocks ahead) and splits them
//     into bf16 planes once per k-block; the second wave of the SIMD issues MFMAs meanwhile;
//   * two s_barriers per k-block with the two waves of a SIMD half a k-block apart (one splits operands while the other multiplies),
//     counted vmcnt waits (the ring stays ahead across the barriers).
// Measured on the K loop alone (tools/lds256_probe.cpp, profiles/r06_lds256_probe*.txt): 22.5 cycles per MFMA and SIMD with the
// weights streaming from HBM (71 % of the issue rate) against 39 in the product's 64 x 128 tile.
//
// K split ACROSS workgroups. A workgroup computes ONE HALF of K -- seg[0] (the layer's input x) or seg[1] (its own h(t - 1)) --
// as two chains (the quarters the four waves of a gemm_tile own), s = p_a + p_b. The two halves of a tile are neighbouring
// workgroups of the launch; whichever finishes LAST adds the other's half sum (a 128 KiB slab in device memory, handed over with
// one agent-scope release / acquire pair and a ticket: no spinning, any placement of the two on XCDs / CUs) and runs the epilogue:
//   gates = (p0 + p1) + (p2 + p3) + bias        -- the sum gemm_tile's LDS reduction forms, in its order (fp32 add commutes),
// so per element the result is BITWISE that of every other tile shape: which kernel a row's step runs on never shows.
// (ksplit = 1: one workgroup runs both halves one after the other and parks the first half sum in the slab itself.)
// What the halving buys: 544 work items of 16 / 32 / 40 k-blocks per tick instead of 272 of 32 / 64 / 80 -- the chain
// h(t) -> h(t + 1) of rnn4 (K' = 2560) is one item long, and 256 CUs end a tick level.
//
// Epilogue in registers: the 16 x 16 accumulator block of a lane holds 4 rows of ONE column; the four gates of a unit are four
// neighbouring columns (pack_weights_split), i.e. the four lanes of a quad -- a 4 x 4 transpose on DPP quad permutes gives every
// lane the (i, f, g, o) of one (row, unit): no LDS round trip, the ring is all the LDS the kernel uses.
#include "rc_internal.h"
#include <hip/hip_ext.h>
#include "rc_gates.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// -DRC_TRACE_TILES (tools/lds_trace.py): every workgroup appends {H | half << 16 | ksplit << 20 | last << 24, CU, 5 wall-clock stamps, rows}
// to a host-provided buffer. Not compiled into the product library.
#ifdef RC_TRACE_TILES
__device__ unsigned long long* g_lds_trace_buf = nullptr;
__device__ unsigned long long g_lds_trace_cap = 0;
__device__ unsigned long long g_lds_trace_n = 0;
extern "C" int rc_trace_lds_set(unsigned long long* buf, unsigned long long cap_records) {
    const unsigned long long zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lds_trace_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lds_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lds_trace_cap), &cap_records, sizeof(cap_records)) == hipSuccess ? 0 : -1;
}
#define LDS_T(i) do { if (threadIdx.x == 0) trace_t[i] = wall_clock64(); } while (0)
#define LDS_TRACE_OUT(LAST)                                                                                                  \
    do { if (threadIdx.x == 0 && g_lds_trace_buf) {                                                                          \
        const unsigned long long idx = atomicAdd(&g_lds_trace_n, 1ull);                                                      \
        if (idx < g_lds_trace_cap) { unsigned long long* r_ = g_lds_trace_buf + idx * 8;                                     \
            r_[0] = (unsigned long long)P.H | ((unsigned long long)kh0 << 16) | ((unsigned long long)P.ksplit << 20) | ((unsigned long long)(LAST) << 24) | (1ull << 32);  \
            r_[1] = __smid(); r_[2] = trace_t[0]; r_[3] = trace_t[1]; r_[4] = trace_t[2]; r_[5] = trace_t[3]; r_[6] = wall_clock64(); r_[7] = nrows; } } } while (0)
#else
#define LDS_T(i) do { } while (0)
#define LDS_TRACE_OUT(LAST) do { } while (0)
#endif

#ifndef RC_LDS_PHASE
#define RC_LDS_PHASE 0        // 1: two barriers per k-block, waves 4-7 half a k-block behind waves 0-3 (measured slower: profiles/r06_lds_kernel_notes.txt)
#endif
#ifdef RC_LDS_NODEP           // TIMING EXPERIMENT ONLY (wrong results): every MFMA of a slot on another accumulator -- no dependent chain inside a slot
#define RC_DJ(J, P) (((J) + (P)) & 7)
#else
#define RC_DJ(J, P) (J)
#endif
#ifndef RC_LDS_ABL
#define RC_LDS_ABL 0          // TIMING EXPERIMENTS ONLY (wrong results), bit mask: 1 no operand split, 2 no barrier, 4 no ds_read, 8 no activation loads, 16 no weight DMA
#endif
#ifndef RC_LDS_PKSUB
#define RC_LDS_PKSUB 1        // operand split on v_pk_add_f32 (rc_gemm.hip's form); 0: plain v_sub_f32
#endif

namespace {

constexpr int kWaves = 8;                     // waves per workgroup
constexpr int kStage = 8 * 3 * 1024;          // bytes of one k-block of weight planes in the ring: [column block][plane][lane] x 16 B
constexpr int kRing = 4;                      // stages (a power of two): see the phase shift of the two wave groups below
constexpr int kNA = 4;                        // activation loads per wave and k-block (2 row blocks x 2 chunks of 16 k)
constexpr int kPW = 3;                        // LDS-DMA pieces per wave and k-block (one column block's three planes)

#define RC_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

// one float2 pair of the operand split (rc_gemm.hip: split3): a = hi + mid + lo exactly, each a truncated bf16
__device__ __forceinline__ float sub1(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_pair(const f32x2 a, unsigned& h, unsigned& m, unsigned& l) {
    const u32x2 ua = __builtin_bit_cast(u32x2, a);
#if RC_LDS_PKSUB
    const f32x2 r1 = a - __builtin_bit_cast(f32x2, ua & 0xffff0000u);
    const u32x2 um = __builtin_bit_cast(u32x2, r1);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, um & 0xffff0000u);
#else
    const f32x2 r1 = f32x2{sub1(a[0], __builtin_bit_cast(float, ua[0] & 0xffff0000u)), sub1(a[1], __builtin_bit_cast(float, ua[1] & 0xffff0000u))};
    const u32x2 um = __builtin_bit_cast(u32x2, r1);
    const f32x2 r2 = f32x2{sub1(r1[0], __builtin_bit_cast(float, um[0] & 0xffff0000u)), sub1(r1[1], __builtin_bit_cast(float, um[1] & 0xffff0000u))};
#endif
    const u32x2 ul = __builtin_bit_cast(u32x2, r2);
    h = __builtin_amdgcn_perm(ua[1], ua[0], 0x07060302u);
    m = __builtin_amdgcn_perm(um[1], um[0], 0x07060302u);
    l = __builtin_amdgcn_perm(ul[1], ul[0], 0x07060302u);
}
__device__ __forceinline__ void split_block(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
    const f32x2 a[4] = {f32x2{x0[0], x0[1]}, f32x2{x0[2], x0[3]}, f32x2{x1[0], x1[1]}, f32x2{x1[2], x1[3]}};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned hh, mm, ll;
        split_pair(a[d], hh, mm, ll);
        h[d] = hh; m[d] = mm; l[d] = ll;
    }
}

// Loads the compiler must not count: it would drain the ring (vmcnt(0)) at the first use of an activation. Completion is waited
// for by hand (WAIT_RAW / the slot's vmcnt) and tied to the destination registers with "+v" operands.
__device__ __forceinline__ void gload(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload1k(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(d) : "v"(p) : "memory"); }
template <int OFF>   // the instruction offset moves the global AND the LDS address: plane p of a column block sits 1 KiB further in both
__device__ __forceinline__ void glds(const u32x4* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void dsread(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }

template <int CTRL>
__device__ __forceinline__ float quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// 4 x 4 transpose inside every quad of lanes: in v[e] = (gate q, row e) on quad lane q; out g[k] = (gate k, row q)
__device__ __forceinline__ void quad_transpose(const f32x4& v, const int q, f32x4& g) {
    const bool odd = q & 1, hi = q & 2;
    const float x0 = quad<0xB1>(v[0]), x1 = quad<0xB1>(v[1]), x2 = quad<0xB1>(v[2]), x3 = quad<0xB1>(v[3]);   // quad lane ^ 1
    // pairs of gates (2 (q / 2), + 1) for row (q & 1) in a0, a1 and for row 2 + (q & 1) in a2, a3
    const float a0 = odd ? x1 : v[0], a1 = odd ? v[1] : x0, a2 = odd ? x3 : v[2], a3 = odd ? v[3] : x2;
    const float y0 = quad<0x4E>(a0), y1 = quad<0x4E>(a1), y2 = quad<0x4E>(a2), y3 = quad<0x4E>(a3);           // quad lane ^ 2
    g[0] = hi ? y2 : a0; g[1] = hi ? y3 : a1; g[2] = hi ? a2 : y0; g[3] = hi ? a3 : y1;
}

__device__ __forceinline__ bool locate(const LdsLaunch& L, int& pi, int& m_tile, int& n_tile, int& kh0, int& kh1) {
    pi = 0;
#pragma unroll
    for (int q = 1; q < RC_LDS_MAXP; ++q)
        if (q < L.n && (int)blockIdx.x >= L.p[q].wg_base) pi = q;
    const LdsProblem& P = L.p[pi];
    const int local = blockIdx.x - P.wg_base;
    // the two halves of a tile are neighbours in the grid (they finish together); tiles of one weight slice (same n_tile, row
    // tiles m) are a multiple of 8 apart: same XCD, the slice goes through that L2 once
    const int t = P.ksplit == 2 ? local >> 1 : local;
    kh0 = P.ksplit == 2 ? (local & 1) : 0;
    kh1 = P.ksplit == 2 ? kh0 + 1 : 2;
    n_tile = t % P.n_tiles;
    m_tile = t / P.n_tiles;
    return m_tile < P.m_tiles;
}

}  // namespace

__global__ __launch_bounds__(kWaves * 64, 1) void rc_gemm_lds_kernel(const LdsLaunch L) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[kRing * kStage];
    __shared__ int s_rows[256];
    __shared__ int s_cnt[kWaves];
    __shared__ int s_last;
#ifdef RC_TRACE_TILES
    unsigned long long trace_t[4] = {0, 0, 0, 0};
#endif
    LDS_T(0);
    int pi, m_tile, n_tile, kh0, kh1;
    if (!locate(L, pi, m_tile, n_tile, kh0, kh1)) return;
    const LdsProblem& P = L.p[pi];
    const int B = L.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int grp = wave >> 2;                // waves 0-3 | 4-7: one wave of every SIMD each
    const unsigned ring0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)ring;

    // The first two k-blocks of weight planes depend on nothing the prologue computes: requested here, they travel (from HBM: every
    // slice has one reader per launch) while the rows are selected and their step numbers read -- three dependent memory round trips.
    const int Qs = P.Qs, Qh = Qs >> 1, Qq = Qh >> 1;               // k-blocks of the layer (K' = 2 H), of a half, of a quarter
    const unsigned my_lds = ring0 + (unsigned)wave * 3072u;
    {
        const u32x4* g_ = reinterpret_cast<const u32x4*>(P.Ws) + ((long long)(n_tile * 8 + wave) * Qs + (long long)kh0 * Qh) * 192 + lane;
        glds<0>(g_, my_lds); glds<1024>(g_, my_lds); glds<2048>(g_, my_lds);
        glds<0>(g_ + 192, my_lds + kStage); glds<1024>(g_ + 192, my_lds + kStage); glds<2048>(g_ + 192, my_lds + kStage);
    }
    // ---- active rows of this row tile (gemm_tile's stateless compaction, 256 rows per tile) ------------------------------------
    const int lo = m_tile * 256;
    int nrows;
    if (P.flag_bit == 0) {
        nrows = min(256, B - lo);
        if (nrows <= 0) return;
        if (tid < 256) s_rows[tid] = lo + min(tid, nrows - 1);
        __syncthreads();
    } else {
        int total = 0;
        for (int base = 0; base < B && total < lo + 256; base += kWaves * 64) {
            const int r = base + tid;
            const bool f = r < B && (P.flags[r] & P.flag_bit);
            const unsigned long long bal = __ballot(f);
            if (lane == 0) s_cnt[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0, sum = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const int cw = s_cnt[w];
                woff += (w < wave) ? cw : 0;
                sum += cw;
            }
            const int idx = total + woff + __popcll(bal & ((1ull << lane) - 1ull));
            if (f && idx >= lo && idx < lo + 256) s_rows[idx - lo] = r;
            total += sum;
            __syncthreads();
        }
        nrows = min(256, total - lo);
        if (nrows <= 0) return;                                     // (uniform: both halves of the tile leave, no ticket is drawn)
        if (tid < 256 && tid >= nrows) s_rows[tid] = s_rows[0];
        __syncthreads();
    }
    const int tile = m_tile * P.n_tiles + n_tile;
    const int row_r[2] = {s_rows[32 * wave + i], s_rows[32 * wave + 16 + i]};
    const bool need_st = (P.seg[0].par_mode | P.seg[1].par_mode) != 0;
    const int st_r[2] = {need_st ? P.steps[row_r[0]] + P.step_off : 0, need_st ? P.steps[row_r[1]] + P.step_off : 0};

    f32x4 acc0[2][8], acc1[2][8];             // the two quarter chains of the current half
    f32x4 raw[2][2][2];                       // [buffer][row block][chunk of 16 k]
    u32x4 pl[2][3];                           // [row block][plane] of the k-block being multiplied
    u32x4 bf[2][3];                           // [buffer][plane] of one column block's weights
    const unsigned rd0 = ring0 + lane * 16;
    float* const my_slab = P.slab + ((long long)tile * 2) * 32768 + ((wave * 2) * 8 * 64 + lane) * 4;   // + kh * 32768 + (r * 8 + j) * 256

    for (int kh = kh0; kh < kh1; ++kh) {
        const GemmSeg& sg = P.seg[kh];
        const float* pa[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int st = st_r[r];
            const int par = sg.par_mode == RC_PAR_SRC ? ((st - 1) % RC_HBUF) : (sg.par_mode == RC_PAR_DST ? (st % RC_HBUF) : 0);
            pa[r] = sg.base + (long long)par * sg.par_stride + rc_pk(row_r[r], 4 * kq, sg.ld);
        }
        const u32x4* pw = reinterpret_cast<const u32x4*>(P.Ws) + ((long long)(n_tile * 8 + wave) * Qs + (long long)kh * Qh) * 192 + lane;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc0[r][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[r][j] = acc0[r][j]; }

#define ISSUE_A(BUF, QI)                                                                                                    \
    do {                                                                                                                    \
        const long long ko_ = (long long)min((QI), Qh - 1) * 512;                                                           \
        gload(raw[BUF][0][0], pa[0] + ko_); gload1k(raw[BUF][0][1], pa[0] + ko_);                                           \
        gload(raw[BUF][1][0], pa[1] + ko_); gload1k(raw[BUF][1][1], pa[1] + ko_);                                           \
    } while (0)
#define ISSUE_B(QI)                                                                                                         \
    do {                                                                                                                    \
        const u32x4* g_ = pw + (long long)min((QI), Qh - 1) * 192;                                                          \
        const unsigned dst_ = my_lds + (unsigned)((QI) & (kRing - 1)) * kStage;                                                       \
        glds<0>(g_, dst_); glds<1024>(g_, dst_); glds<2048>(g_, dst_);                                                      \
    } while (0)
#define WAIT_RAW(BUF, N)                                                                                                    \
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(raw[BUF][0][0]), "+v"(raw[BUF][0][1]), "+v"(raw[BUF][1][0]), "+v"(raw[BUF][1][1]) : "n"(N) : "memory")
        // one column block: its weight planes were requested a slot earlier; the next block's go out first
#define SLOT(J, Q, ACC)                                                                                                     \
    do {                                                                                                                    \
        constexpr int bi_ = (J) & 1;                                                                                        \
        if ((RC_LDS_PHASE && (J) == 3) || (J) == 7) {                                                                       \
            /* the group whose NEXT k-block starts behind this barrier (group 0 at slot 7, group 1 -- half a k-block behind -- at slot 3, */ \
            /* the same barrier) has its pieces of B(Q + 1) in LDS first: A(Q + 2), B(Q + 2) may still be in flight */      \
            if (!(RC_LDS_ABL & 24) && (!RC_LDS_PHASE || grp == ((J) == 3))) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kNA + kPW) : "memory");   \
            if (RC_LDS_ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else                                      \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                 \
        }                                                                                                                   \
        if (RC_LDS_ABL & 4) { asm volatile("" : "+v"(bf[bi_ ^ 1][0]), "+v"(bf[bi_ ^ 1][1]), "+v"(bf[bi_ ^ 1][2])); }         \
        else if ((J) == 7) {                                                                                                \
            const unsigned a_ = rd0 + (unsigned)(((Q) + 1) & (kRing - 1)) * kStage;                                         \
            dsread<0>(bf[bi_ ^ 1][0], a_); dsread<1024>(bf[bi_ ^ 1][1], a_); dsread<2048>(bf[bi_ ^ 1][2], a_);              \
        } else {                                                                                                            \
            const unsigned a_ = rd0 + (unsigned)((Q) & (kRing - 1)) * kStage;                                               \
            dsread<((J) + 1) * 3072>(bf[bi_ ^ 1][0], a_); dsread<((J) + 1) * 3072 + 1024>(bf[bi_ ^ 1][1], a_);              \
            dsread<((J) + 1) * 3072 + 2048>(bf[bi_ ^ 1][2], a_);                                                            \
        }                                                                                                                   \
        if (!(RC_LDS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[bi_][0]), "+v"(bf[bi_][1]), "+v"(bf[bi_][2]) :: "memory");   \
        /* gemm_tile's products in its order (mma_kblock): small terms first */                                             \
        RC_MFMA(pl[0][2], bf[bi_][0], ACC[0][RC_DJ(J, 0)]); RC_MFMA(pl[1][2], bf[bi_][0], ACC[1][RC_DJ(J, 0)]);             \
        RC_MFMA(pl[0][0], bf[bi_][2], ACC[0][RC_DJ(J, 1)]); RC_MFMA(pl[1][0], bf[bi_][2], ACC[1][RC_DJ(J, 1)]);             \
        RC_MFMA(pl[0][1], bf[bi_][1], ACC[0][RC_DJ(J, 2)]); RC_MFMA(pl[1][1], bf[bi_][1], ACC[1][RC_DJ(J, 2)]);             \
        RC_MFMA(pl[0][1], bf[bi_][0], ACC[0][RC_DJ(J, 3)]); RC_MFMA(pl[1][1], bf[bi_][0], ACC[1][RC_DJ(J, 3)]);             \
        RC_MFMA(pl[0][0], bf[bi_][1], ACC[0][RC_DJ(J, 4)]); RC_MFMA(pl[1][0], bf[bi_][1], ACC[1][RC_DJ(J, 4)]);             \
        RC_MFMA(pl[0][0], bf[bi_][0], ACC[0][RC_DJ(J, 5)]); RC_MFMA(pl[1][0], bf[bi_][0], ACC[1][RC_DJ(J, 5)]);             \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)
        // k-block Q: its activations (requested two k-blocks ago) are split, the buffer goes to k-block Q + 2 together with the ring stage
#define ITER(CUR, Q, ACC)                                                                                                   \
    do {                                                                                                                    \
        if (!(RC_LDS_ABL & 24)) WAIT_RAW(CUR, kNA + kPW);                                                                   \
        if (RC_LDS_ABL & 1) { pl[0][0] = __builtin_bit_cast(u32x4, raw[CUR][0][0]); pl[0][1] = __builtin_bit_cast(u32x4, raw[CUR][0][1]); pl[0][2] = pl[0][0];   \
                              pl[1][0] = __builtin_bit_cast(u32x4, raw[CUR][1][0]); pl[1][1] = __builtin_bit_cast(u32x4, raw[CUR][1][1]); pl[1][2] = pl[1][0]; }  \
        else {                                                                                                              \
        split_block(raw[CUR][0][0], raw[CUR][0][1], pl[0][0], pl[0][1], pl[0][2]);                                          \
        split_block(raw[CUR][1][0], raw[CUR][1][1], pl[1][0], pl[1][1], pl[1][2]); }                                        \
        asm volatile("" : "+v"(pl[0][0]), "+v"(pl[0][1]), "+v"(pl[0][2]), "+v"(pl[1][0]), "+v"(pl[1][1]), "+v"(pl[1][2]));  \
        if (!(RC_LDS_ABL & 8)) ISSUE_A(CUR, (Q) + 2);                                                                       \
        if (!(RC_LDS_ABL & 16)) ISSUE_B((Q) + 2);                                                                           \
        SLOT(0, Q, ACC); SLOT(1, Q, ACC); SLOT(2, Q, ACC); SLOT(3, Q, ACC); SLOT(4, Q, ACC); SLOT(5, Q, ACC); SLOT(6, Q, ACC); SLOT(7, Q, ACC);   \
    } while (0)

        // ---- prologue of the half: two k-blocks in flight, the first column block's planes in registers
        if (kh == kh0) {                        // (the first half's weight planes are on their way since the top of the kernel)
            ISSUE_A(0, 0); ISSUE_A(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(kNA) : "memory");
        } else {
            __syncthreads();                   // second half of a ksplit = 1 tile: every wave is done with the ring
            ISSUE_A(0, 0); ISSUE_B(0); ISSUE_A(1, 1); ISSUE_B(1);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(kNA + kPW) : "memory");
        }
        // PHASE SHIFT. Waves w and w + 4 share a SIMD. With one barrier per k-block both reach their operand split (85 VALU
        // instructions in front of a k-block's 96 MFMAs) together and the matrix pipe idles through both: 4,314 cycles per k-block
        // against 3,072 of MFMA issue (tools/lds256_probe.cpp). Two barriers per k-block (slots 3 and 7) and ONE EXTRA in front of
        // group 1's loop put waves 4-7 half a k-block behind waves 0-3: a wave splits while its SIMD partner multiplies. The ring
        // is 4 deep for it (a stage is read for 1.5 k-blocks). Group 0 pays the extra barrier back behind the loop.
        if (RC_LDS_PHASE && grp) asm volatile("s_barrier" ::: "memory");
        dsread<0>(bf[0][0], rd0); dsread<1024>(bf[0][1], rd0); dsread<2048>(bf[0][2], rd0);
        if (kh == kh0) LDS_T(1);
        for (int q = 0; q < Qq; q += 2) { ITER(0, q, acc0); ITER(1, q + 1, acc0); }
        for (int q = Qq; q < Qh; q += 2) { ITER(0, q, acc1); ITER(1, q + 1, acc1); }
        if (RC_LDS_PHASE && !grp) asm volatile("s_barrier" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#undef ITER
#undef SLOT
#undef WAIT_RAW
#undef ISSUE_B
#undef ISSUE_A
        // the half sum p_a + p_b
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc0[r][j] += acc1[r][j];
        if (P.ksplit == 2 || kh == 0) {        // parked for whoever finishes the tile (ksplit = 1: this workgroup itself)
            float* s = my_slab + (long long)kh * 32768;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(s + (r * 8 + j) * 256) = acc0[r][j];
        }
    }

    LDS_T(2);
    // The epilogue's own operands (cell state, bias, which copy of h the step writes) are requested HERE: they travel during the
    // hand-over. (A workgroup that turns out to be first has read them for nothing; nobody writes them before the tile is finished.)
    const int q = i & 3, u = i >> 2;
    int rr_[2], r2_[2], dst_[2];
    float c_prev[2][8], bias[8];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        rr_[r] = 32 * wave + 16 * r + 4 * kq + q;
        r2_[r] = s_rows[rr_[r] < nrows ? rr_[r] : 0];
        dst_[r] = (P.steps[r2_[r]] + P.step_off) % RC_HBUF;
#pragma unroll
        for (int j = 0; j < 8; ++j) c_prev[r][j] = P.cstate[(long long)r2_[r] * P.H + n_tile * 32 + j * 4 + u];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = P.bias[n_tile * 128 + 16 * j + i];
    // ---- hand-over: the workgroup that finishes the tile LAST adds the other half sum and runs the epilogue ---------------------
    int other = 0;                             // which half sits in the slab
    if (P.ksplit == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's slab stores have left
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (behind the write-back: ROCm 7.2 may drop the fence's own wait)
            const int t = __hip_atomic_fetch_add(&P.tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == 1) {
                __hip_atomic_store(&P.tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the launch that uses this slot next
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            s_last = t;
        }
        __syncthreads();
        if (s_last == 0) { LDS_T(3); LDS_TRACE_OUT(0); return; }
        other = 1 - kh0;
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own stores of the first half: visible to this wave's loads behind this
    }
    LDS_T(3);
    {
        const float* s = my_slab + (long long)other * 32768;
        f32x4 o[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) o[r][j] = *reinterpret_cast<const f32x4*>(s + (r * 8 + j) * 256);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc0[r][j] += o[r][j];    // (p0 + p1) + (p2 + p3)
    }
    // ---- LSTM epilogue: lane (kq, u = i / 4, q = i % 4) owns (row 4 kq + q, unit u) of every 16 x 16 block ---------------------
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const bool ok = rr_[r] < nrows;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = acc0[r][j];
            v[0] += bias[j]; v[1] += bias[j]; v[2] += bias[j]; v[3] += bias[j];
            f32x4 g;
            quad_transpose(v, q, g);
            float cn, hn;
            rc_lstm_cell(g[0], g[1], g[2], g[3], c_prev[r][j], cn, hn);
            if (ok) {
                const int unit = n_tile * 32 + j * 4 + u;
                P.cstate[(long long)r2_[r] * P.H + unit] = cn;
                P.hstate[(long long)dst_[r] * P.h_par_stride + rc_pk(r2_[r], unit, P.H)] = hn;
            }
        }
    }
    LDS_TRACE_OUT(1);
}

void rc_launch_gemm_lds(const LdsLaunch& L, int total_wg, hipStream_t s, hipEvent_t stop) {
    const dim3 g(total_wg), b(kWaves * 64);
    if (stop) hipExtLaunchKernelGGL(rc_gemm_lds_kernel, g, b, 0, s, nullptr, stop, 0, L);
    else hipLaunchKernelGGL(rc_gemm_lds_kernel, g, b, 0, s, L);
}
