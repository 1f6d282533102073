// Gate GEMM of the LSTM layer steps with the WEIGHTS SHARED BY THE WORKGROUP (round 6) -- gfx950 only.
//
// Replaces, for contexts of >= RC_LDS_MIN_BATCH (65) rows in split-product mode and problems of >= RC_LDS_MIN_ROWS (half the batch) rows, the 64-row tiles of rc_gemm.hip on the twelve LSTM
// layer steps of a frame (net/sig_mp.py:126-129 -> articulate/utils/torch/rnn.py:129-133, aten::lstm; 96 % of the path's FLOPs).
//
// Why. rc_gemm.hip splits K over the four waves of a 64 x 128 tile; nothing is shared inside a workgroup, so every weight byte is
// pulled through FOUR CUs' vector-memory paths (the four row tiles of a batch of 256) and each wave issues its MFMAs, its operand
// split and 32 global loads per k-block in order: 39 cycles per MFMA against a 16-cycle issue rate. Here one
// workgroup owns ALL rows of a 128-column slice (up to 256: one row tile of the batch):
//   * 8 waves (two per SIMD) x all 128 columns, M split, no K split inside the workgroup: wave w owns the 16-row blocks w and w + 8
//     of the tile's compacted row list, so a tile with fewer than 256 active rows (the rnn4 / rnn6 problems of a mixed batch: the
//     rows that see the camera) still loads every SIMD evenly -- a wave multiplies 2, 1 or 0 row blocks (k_half<NR>);
//   * the weight planes of a k-block (24 KiB) are staged ONCE per workgroup in a 4-deep LDS ring by global_load_lds_dwordx4
//     (each wave moves one column block's three planes) and read back with ds_read_b128 one column block ahead of its MFMAs;
//   * every wave streams its own 32 rows of fp32 activations straight into VGPRs (two buffers, two k-blocks ahead) and splits them
//     into bf16 planes once per k-block; the second wave of the SIMD issues MFMAs meanwhile;
//   * one s_barrier per k-block, counted vmcnt waits (the ring stays two k-blocks ahead across the barrier).
// Measured on the K loop alone (tools/lds256_probe.cpp, profiles/r06_lds256_probe*.txt): 22.5 cycles per MFMA and SIMD with the
// weights streaming from HBM (71 % of the issue rate) against 39 in the product's 64 x 128 tile. What the remaining 29 % are
// (ablations on the product kernel, profiles/r06_lds_kernel_notes.txt): the issue of the wave's own 7 vector-memory instructions per
// k-block 11 %, the operand split 10 %, barrier + ds_reads 2 %; with all of them removed the loop runs at the MFMA issue rate.
// Tried and dropped there (profiles/r06_lds_kernel_with_experiments.hip.txt): the two waves of a SIMD half a k-block apart (two
// barriers per k-block: -12 %), v_sub_f32 instead of v_pk_add_f32 (+-0), MFMAs of a slot on different accumulators (+-0).
//
// K split ACROSS workgroups. A workgroup computes ONE HALF of K -- seg[0] (the layer's input x) or seg[1] (its own h(t - 1)) --
// as two chains (the quarters the four waves of a gemm_tile own), s = p_a + p_b. The two halves of a tile are neighbouring
// workgroups of the launch: the seg[0] half parks its half sum in a 128 KiB slab in device memory (one agent-scope release + a flag)
// and leaves, the seg[1] half waits for the flag (acquire), adds the slab to its own half sum and runs the epilogue:
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

#ifndef RC_LAND
#define RC_LAND 1         // (0: measurement builds only -- the resident kernel without its drains in front of the loop headers)
#endif

namespace {

constexpr int kWaves = 8;                     // waves per workgroup
constexpr int kStage = 8 * 3 * 1024;          // bytes of one k-block of weight planes in the ring: [column block][plane][lane] x 16 B
constexpr int kRing = 4;                      // stages (a power of two; three are in use at any time)
constexpr int kPW = 3;                        // LDS-DMA pieces per wave and k-block (one column block's three planes)

#define RC_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

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
// for by hand (counted vmcnt) and tied to the destination registers with "+v" operands.
__device__ __forceinline__ void gload(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload1k(f32x4& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(d) : "v"(p) : "memory"); }
template <int OFF>   // the instruction offset moves the global AND the LDS address: plane p of a column block sits 1 KiB further in both
__device__ __forceinline__ void glds(const u32x4* g, unsigned lds_) {
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_);   // m0 takes a scalar register: said here, where it is used (free when it already is one)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void dsread(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

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

template <int MAXP, typename PP>
__device__ __forceinline__ bool locate(const PP Lp, const int Ln, const int item, int& pi, int& m_tile, int& n_tile, int& kh0, int& kh1) {
    pi = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < Ln && item >= Lp[q].wg_base) pi = q;
    const auto& P = Lp[pi];
    const int local = item - P.wg_base;
    // the two halves of a tile are neighbours in the grid (they finish together); tiles of one weight slice (same n_tile, row
    // tiles m) are a multiple of 8 apart: same XCD, the slice goes through that L2 once
    const int t = P.ksplit == 2 ? local >> 1 : local;
    kh0 = P.ksplit == 2 ? (local & 1) : 0;
    kh1 = P.ksplit == 2 ? kh0 + 1 : 2;
    n_tile = t % P.n_tiles;
    m_tile = t / P.n_tiles;
    return m_tile < P.m_tiles;
}

struct HalfArgs {
    const float* pa[2];       // this lane's activation pointers of the wave's two row blocks (this half's segment)
    const u32x4* pw;          // this wave's column block of the weight planes, first k-block of the half
    unsigned rd0;             // LDS address this lane reads weight planes from (stage 0, column block 0, plane 0)
    unsigned my_lds;          // LDS address this wave's DMA pieces go to (stage 0)
    int Qh, Qq;               // k-blocks of the half, of a quarter
    bool primed;              // the first two k-blocks of weight planes are already on their way (first half of the workgroup)
};

// One half of K for a wave that multiplies NR of its two row blocks (0: the wave only moves its share of the weight planes).
// k-block Q: [activations of Q (requested two k-blocks ago) are split | the buffer and the ring stage go to k-block Q + 2 |
// 8 slots: one column block each, its planes requested a slot earlier | at slot 7: B(Q + 1) landed, barrier].
// Vector-memory operations complete in order, so with the issue order A(Q + 1), B(Q + 1), A(Q + 2), B(Q + 2) "all but the NA + 3
// youngest" is exactly "A(Q + 1) and B(Q + 1) have landed".
template <int NR, bool SHORT_OK>
__device__ __forceinline__ void k_half(const HalfArgs& c, f32x4 (&acc0)[2][8], f32x4 (&acc1)[2][8]) {
    constexpr int NA = 2 * NR;                // activation loads per k-block
    f32x4 raw[2][2][2];                       // [buffer][row block][chunk of 16 k]
    u32x4 pl[2][3];                           // [row block][plane] of the k-block being multiplied
    u32x4 bf[2][3];                           // [buffer][plane] of one column block's weights
    const int Qh = c.Qh;
#define ISSUE_A(BUF, QI)                                                                                                    \
    do {                                                                                                                    \
        const long long ko_ = (long long)min((QI), Qh - 1) * 512;                                                           \
        if constexpr (NR > 0) { gload(raw[BUF][0][0], c.pa[0] + ko_); gload1k(raw[BUF][0][1], c.pa[0] + ko_); }             \
        if constexpr (NR > 1) { gload(raw[BUF][1][0], c.pa[1] + ko_); gload1k(raw[BUF][1][1], c.pa[1] + ko_); }             \
    } while (0)
#define ISSUE_B(QI)                                                                                                         \
    do {                                                                                                                    \
        const u32x4* g_ = c.pw + (long long)min((QI), Qh - 1) * 192;                                                        \
        const unsigned dst_ = c.my_lds + (unsigned)((QI) & (kRing - 1)) * kStage;                                           \
        glds<0>(g_, dst_); glds<1024>(g_, dst_); glds<2048>(g_, dst_);                                                      \
    } while (0)
#define TIE_RAW(BUF)                                                                                                        \
    do {                                                                                                                    \
        if constexpr (NR > 0) asm volatile("" : "+v"(raw[BUF][0][0]), "+v"(raw[BUF][0][1]));                                \
        if constexpr (NR > 1) asm volatile("" : "+v"(raw[BUF][1][0]), "+v"(raw[BUF][1][1]));                                \
    } while (0)
#define SLOT(J, Q, ACC)                                                                                                     \
    do {                                                                                                                    \
        constexpr int bi_ = (J) & 1;                                                                                        \
        if ((J) == 7) {      /* B(Q + 1) has landed (A(Q + 2), B(Q + 2) may be in flight); every wave is done reading stage Q - 1 */  \
            wait_vm<NA + kPW>();                                                                                            \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                 \
        }                                                                                                                   \
        if constexpr (NR > 0) {                                                                                             \
            if ((J) == 7) {                                                                                                 \
                const unsigned a_ = c.rd0 + (unsigned)(((Q) + 1) & (kRing - 1)) * kStage;                                   \
                dsread<0>(bf[bi_ ^ 1][0], a_); dsread<1024>(bf[bi_ ^ 1][1], a_); dsread<2048>(bf[bi_ ^ 1][2], a_);          \
            } else {                                                                                                        \
                const unsigned a_ = c.rd0 + (unsigned)((Q) & (kRing - 1)) * kStage;                                         \
                dsread<((J) + 1) * 3072>(bf[bi_ ^ 1][0], a_); dsread<((J) + 1) * 3072 + 1024>(bf[bi_ ^ 1][1], a_);          \
                dsread<((J) + 1) * 3072 + 2048>(bf[bi_ ^ 1][2], a_);                                                        \
            }                                                                                                               \
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[bi_][0]), "+v"(bf[bi_][1]), "+v"(bf[bi_][2]) :: "memory");        \
            /* gemm_tile's products in its order (mma_kblock): small terms first */                                         \
            RC_MFMA(pl[0][2], bf[bi_][0], ACC[0][J]); if constexpr (NR > 1) RC_MFMA(pl[1][2], bf[bi_][0], ACC[1][J]);       \
            RC_MFMA(pl[0][0], bf[bi_][2], ACC[0][J]); if constexpr (NR > 1) RC_MFMA(pl[1][0], bf[bi_][2], ACC[1][J]);       \
            RC_MFMA(pl[0][1], bf[bi_][1], ACC[0][J]); if constexpr (NR > 1) RC_MFMA(pl[1][1], bf[bi_][1], ACC[1][J]);       \
            RC_MFMA(pl[0][1], bf[bi_][0], ACC[0][J]); if constexpr (NR > 1) RC_MFMA(pl[1][1], bf[bi_][0], ACC[1][J]);       \
            RC_MFMA(pl[0][0], bf[bi_][1], ACC[0][J]); if constexpr (NR > 1) RC_MFMA(pl[1][0], bf[bi_][1], ACC[1][J]);       \
            RC_MFMA(pl[0][0], bf[bi_][0], ACC[0][J]); if constexpr (NR > 1) RC_MFMA(pl[1][0], bf[bi_][0], ACC[1][J]);       \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
        }                                                                                                                   \
    } while (0)
#define ITER(CUR, Q, ACC)                                                                                                   \
    do {                                                                                                                    \
        if constexpr (NR > 0) {                                                                                             \
            wait_vm<NA + kPW>();                                                                                            \
            TIE_RAW(CUR);                                                                                                   \
            split_block(raw[CUR][0][0], raw[CUR][0][1], pl[0][0], pl[0][1], pl[0][2]);                                      \
            asm volatile("" : "+v"(pl[0][0]), "+v"(pl[0][1]), "+v"(pl[0][2]));                                              \
        }                                                                                                                   \
        if constexpr (NR > 1) {                                                                                             \
            split_block(raw[CUR][1][0], raw[CUR][1][1], pl[1][0], pl[1][1], pl[1][2]);                                      \
            asm volatile("" : "+v"(pl[1][0]), "+v"(pl[1][1]), "+v"(pl[1][2]));                                              \
        }                                                                                                                   \
        ISSUE_A(CUR, (Q) + 2);                                                                                              \
        ISSUE_B((Q) + 2);                                                                                                   \
        SLOT(0, Q, ACC); SLOT(1, Q, ACC); SLOT(2, Q, ACC); SLOT(3, Q, ACC); SLOT(4, Q, ACC); SLOT(5, Q, ACC); SLOT(6, Q, ACC); SLOT(7, Q, ACC);   \
    } while (0)

    // ---- prologue of the half: two k-blocks in flight, the first column block's planes in registers
    if (c.primed) {
        ISSUE_A(0, 0); ISSUE_A(1, 1);
        wait_vm<NA>();
    } else {
        __syncthreads();                       // second half of a ksplit = 1 tile: every wave is done with the ring
        ISSUE_A(0, 0); ISSUE_B(0); ISSUE_A(1, 1); ISSUE_B(1);
        wait_vm<NA + kPW>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (NR > 0) { dsread<0>(bf[0][0], c.rd0); dsread<1024>(bf[0][1], c.rd0); dsread<2048>(bf[0][2], c.rd0); }
    // Resident kernel (SHORT_OK). The loads of this loop are asm the compiler only sees REQUESTING a register (raw[]: two k-blocks ahead,
    // bf[]: one column block ahead); it is free to copy or spill that register the next instruction. The launch-per-tick kernel never gave
    // it a reason to; the resident kernel's longer-lived state does, and its register allocation spills in front of the loop headers -- a
    // fragment not yet landed was spilled and reloaded as the register's OLD content: garbage in column block 0 of the first k-block
    // (found with RC_DBG_DENSE_ITEMS, rc_api.cpp). So in front of a loop header everything lands and is tied to its registers. Cost: two
    // exposed memory latencies per half, ~5 % of an item (RC_LAND=0 builds, profiles/r06_resident_notes.txt).
#define LAND_ALL()                                                                                                          \
    do {                                                                                                                    \
        if constexpr (SHORT_OK && RC_LAND) {                                                                                \
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                        \
            TIE_RAW(0); TIE_RAW(1);                                                                                         \
            if constexpr (NR > 0) asm volatile("" : "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[0][2]), "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(bf[1][2])); \
        }                                                                                                                   \
    } while (0)
    LAND_ALL();
    if (SHORT_OK && c.Qq == 1) {                // a quarter = ONE k-block (linear1 with K' = 128): the two buffers are the two quarters
        ITER(0, 0, acc0); ITER(1, 1, acc1);
    } else {
        for (int q = 0; q < c.Qq; q += 2) { ITER(0, q, acc0); ITER(1, q + 1, acc0); }
        LAND_ALL();
        for (int q = c.Qq; q < Qh; q += 2) { ITER(0, q, acc1); ITER(1, q + 1, acc1); }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    // The last k-blocks' look-ahead loads (clamped, unused) are still landing up to here: their destination registers stay allocated until
    // the wait above. (Without this the straight-line short-K path reused them -- for addresses -- while the loads were in flight: a memory
    // fault two instructions later. The loop form kept them alive as loop-carried values by luck of structure, not by contract.)
    TIE_RAW(0); TIE_RAW(1);
    if constexpr (NR > 0) asm volatile("" : "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[0][2]), "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(bf[1][2]));
#undef LAND_ALL
#undef ITER
#undef SLOT
#undef TIE_RAW
#undef ISSUE_B
#undef ISSUE_A
}

}  // namespace

// One work item (a half tile, or a whole one with ksplit = 1; the padding of a problem's item range to a multiple of 8 included) of a launch:
// the body of rc_gemm_lds_kernel, and of every turn of the resident kernel's loop. Returns 1 when the item wrote the tile's h / c (a
// finisher), 0 otherwise; *pi_out = the item's problem.
template <int MAXP, bool WITH_DENSE, typename PP>   // PP: pointer to the launch's problems -- kernel arguments, or the resident kernel's table in the constant address space
__device__ __forceinline__ int lds_item(const PP Lp, const int Ln, const int LB, const int item, unsigned char* ring, int* s_rows, int* s_cnt, int* pi_out) {
#ifdef RC_TRACE_TILES
    unsigned long long trace_t[4] = {0, 0, 0, 0};
#endif
    LDS_T(0);
    int pi, m_tile, n_tile, kh0, kh1;
    const bool in_range = locate<MAXP>(Lp, Ln, item, pi, m_tile, n_tile, kh0, kh1);
    *pi_out = pi;
    if (!in_range) return 0;
#ifdef RC_SKIP_XHALF
    // Measurement builds only (tools/hoist_bound.sh; results are wrong): the half of K that multiplies the layer's INPUT (x . W_ih, seg[0]) is
    // left out -- what a tick costs when only the recurrent half h(t - 1) . W_hh runs per frame, i.e. the floor of a time-hoisted engine
    // whose tall input GEMMs came for free.
    if (Lp[pi].epi == RC_EPI_LSTM) {
        if (Lp[pi].ksplit == 2 && kh0 == 0) return 0;
        kh0 = 1;
    }
#endif
    const auto& P = Lp[pi];
    const int B = LB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const unsigned ring0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)ring;

    // The first two k-blocks of weight planes depend on nothing the prologue computes: requested here, they travel (from HBM: every
    // slice has one reader per launch) while the rows are selected and their step numbers read -- three dependent memory round trips.
    const int Qs = P.Qs, Qh = Qs >> 1;                             // k-blocks of the layer (K' = 2 H), of a half
    HalfArgs hc;
    hc.Qh = Qh; hc.Qq = Qh >> 1;
    hc.rd0 = ring0 + lane * 16;
    hc.my_lds = ring0 + (unsigned)wave * 3072u;
    {
        const u32x4* g_ = reinterpret_cast<const u32x4*>(P.Ws) + ((long long)(n_tile * 8 + wave) * Qs + (long long)kh0 * Qh) * 192 + lane;
        glds<0>(g_, hc.my_lds); glds<1024>(g_, hc.my_lds); glds<2048>(g_, hc.my_lds);
        glds<0>(g_ + 192, hc.my_lds + kStage); glds<1024>(g_ + 192, hc.my_lds + kStage); glds<2048>(g_ + 192, hc.my_lds + kStage);
    }
    // ---- active rows of this row tile (gemm_tile's stateless compaction, 256 rows per tile) ------------------------------------
    const int lo = m_tile * 256;
    int nrows;
    if (P.flag_bit == 0) {
        nrows = min(256, B - lo);
        if (nrows <= 0) return 0;
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
        if (nrows <= 0) return 0;                                   // (uniform: both halves of the tile leave, no flag is raised)
        if (tid < 256 && tid >= nrows) s_rows[tid] = s_rows[0];
        __syncthreads();
    }
    const int tile = m_tile * P.n_tiles + n_tile;
    // wave w owns the 16-row blocks w and w + 8 of the tile's row list: with nblk blocks in use, the waves of every SIMD (w, w + 4)
    // hold ceil or floor of nblk / 4 between them
    const int nblk = (nrows + 15) >> 4;
    const int nr = wave + 8 < nblk ? 2 : (wave < nblk ? 1 : 0);    // row blocks this wave multiplies (uniform in the wave)
    const int row_r[2] = {s_rows[16 * wave + i], s_rows[16 * (wave + 8) + i]};
    const bool need_st = (P.seg[0].par_mode | P.seg[1].par_mode) != 0;
    const int st_r[2] = {need_st ? P.steps[row_r[0]] + P.step_off : 0, need_st ? P.steps[row_r[1]] + P.step_off : 0};

    const bool dense = WITH_DENSE && P.epi != RC_EPI_LSTM;      // (relu(linear1) items exist in the resident kernel's tables only)
    f32x4 acc0[2][8], acc1[2][8];             // the two quarter chains of the current half
    float* const my_slab = P.slab + ((long long)tile * 2) * 32768 + ((wave * 2) * 8 * 64 + lane) * 4;   // + kh * 32768 + (r * 8 + j) * 256

    for (int kh = kh0; kh < kh1; ++kh) {
        const auto& sg = P.seg[kh];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int st = st_r[r];
            const int par = sg.par_mode == RC_PAR_SRC ? ((st - 1) % RC_HBUF) : (sg.par_mode == RC_PAR_DST ? (st % RC_HBUF) : 0);
            const float* a_base = sg.base;
            if (WITH_DENSE && P.sel_bit != 0 && !(P.sel_flags[row_r[r]] & P.sel_bit)) a_base = P.alt[kh];      // a rider's row: its deferred input
            hc.pa[r] = a_base + (long long)par * sg.par_stride + rc_pk(row_r[r], 4 * kq, sg.ld);
        }
        hc.pw = reinterpret_cast<const u32x4*>(P.Ws) + ((long long)(n_tile * 8 + wave) * Qs + (long long)kh * Qh) * 192 + lane;
        hc.primed = kh == kh0;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc0[r][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[r][j] = acc0[r][j]; }
        if (kh == kh0) LDS_T(1);
        if (nr == 2) k_half<2, WITH_DENSE>(hc, acc0, acc1);
        else if (nr == 1) k_half<1, WITH_DENSE>(hc, acc0, acc1);
        else k_half<0, WITH_DENSE>(hc, acc0, acc1);
        if (dense && kh == 1) {
            // a dense layer sums its quarters in sequence, ((p0 + p1) + p2) + p3 as gemm_tile's dense epilogue does: the parked p0 + p1 (this
            // workgroup's own stores: ksplit = 1) comes back here and the epilogue finds the finished sum in acc0
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < nr) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(my_slab + (r * 8 + j) * 256);
                        acc0[r][j] = (o + acc0[r][j]) + acc1[r][j];
                    }
                }
        } else {
            // the half sum p_a + p_b
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc0[r][j] += acc1[r][j];
        }
        if (kh == 0) {                         // parked for the workgroup that finishes the tile (ksplit = 1: this one itself)
            float* s = my_slab;
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < nr) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(s + (r * 8 + j) * 256) = acc0[r][j];
                }
        }
    }

    LDS_T(2);
    // The epilogue's own operands (cell state, bias, which copy of h the step writes) are requested HERE: they travel during the
    // hand-over. (The writer of a pair never gets here with work to do; nobody writes them before the tile is finished.)
    const int q = i & 3, u = i >> 2;
    int rr_[2], r2_[2], dst_[2];
    float c_prev[2][8], bias[8];
    const bool finisher = P.ksplit != 2 || kh0 == 1;
    if (dense) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            rr_[r] = 16 * (wave + 8 * r) + 4 * kq + q;
            r2_[r] = s_rows[rr_[r] < nrows ? rr_[r] : 0];
            dst_[r] = 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) bias[j] = P.bias[n_tile * 128 + 16 * j + i];
    } else if (finisher) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            rr_[r] = 16 * (wave + 8 * r) + 4 * kq + q;
            r2_[r] = s_rows[rr_[r] < nrows ? rr_[r] : 0];
            dst_[r] = (P.steps[r2_[r]] + P.step_off) % RC_HBUF;
#pragma unroll
            for (int j = 0; j < 8; ++j) c_prev[r][j] = P.cstate[(long long)r2_[r] * P.H + n_tile * 32 + j * 4 + u];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) bias[j] = P.bias[n_tile * 128 + 16 * j + i];
    }
    // ---- hand-over -------------------------------------------------------------------------------------------------------------
    // Fixed roles: the seg[0] half (even workgroup of the pair) WRITES its half sum and leaves; the seg[1] half (odd workgroup) keeps
    // its own in registers, waits for the writer's flag and runs the epilogue. The wait cannot deadlock: workgroup i of a dispatch
    // runs on XCD i % 8, so writers (even ids) and waiters (odd ids) never share an XCD -- a writer never queues behind a waiter, and
    // it waits for nothing itself. (A bound on the wait turns a can't-happen into a trap, not a hang.) Measured against the symmetric
    // form -- both halves write, a ticket decides who finishes, nobody waits: +4.8 % end to end (both paid the slab store, its
    // acknowledgement and an atomic round trip; profiles/r06_lds_kernel_notes.txt).
#ifdef RC_SKIP_XHALF
    if (false) {
#else
    if (P.ksplit == 2) {
#endif
        if (kh0 == 0) {                        // writer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's slab stores have left
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (behind the write-back: ROCm 7.2 may drop the fence's own wait)
                __hip_atomic_store(&P.tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            LDS_T(3); LDS_TRACE_OUT(0);
            return 0;
        }
        if (tid == 0) {
            const unsigned long long t_give_up = wall_clock64() + 200000000ull;   // 2 s at 100 MHz
            while (__hip_atomic_load(&P.tickets[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() > t_give_up) __builtin_trap();
            }
            __hip_atomic_store(&P.tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // for the launch that uses this slot next
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own stores of the first half: visible to this wave's loads behind this
    }
    LDS_T(3);
    // ---- (p0 + p1) + (p2 + p3), then the LSTM epilogue: lane (kq, u = i / 4, q = i % 4) owns (row 4 kq + q, unit u) of every block ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r >= nr) continue;
        const bool ok = rr_[r] < nrows;
        if (dense) {
            // relu(sum + bias) -> x1 in rc_pk order: after the quad transpose a lane holds four consecutive columns of one row, one 16-byte store
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 v = acc0[r][j];
                v[0] += bias[j]; v[1] += bias[j]; v[2] += bias[j]; v[3] += bias[j];
                f32x4 g;
                quad_transpose(v, q, g);
                g[0] = fmaxf(g[0], 0.0f); g[1] = fmaxf(g[1], 0.0f); g[2] = fmaxf(g[2], 0.0f); g[3] = fmaxf(g[3], 0.0f);
                if (ok) *reinterpret_cast<f32x4*>(&P.out[rc_pk(r2_[r], n_tile * 128 + 16 * j + 4 * u, P.ldo)]) = g;
            }
            continue;
        }
        f32x4 o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = *reinterpret_cast<const f32x4*>(my_slab + (r * 8 + j) * 256);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = o[j] + acc0[r][j];       // (fp32 add commutes: the order of the two halves does not show)
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
    return 1;
}

__global__ __launch_bounds__(kWaves * 64, 1) void rc_gemm_lds_kernel(const LdsLaunch L) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[kRing * kStage];
    __shared__ int s_rows[256];
    __shared__ int s_cnt[kWaves];
    int pi;
    (void)lds_item<RC_LDS_MAXP, false, const LdsProblem*>(L.p, L.n, L.B, (int)blockIdx.x, ring, s_rows, s_cnt, &pi);
}

// ================================================================================================ resident kernel
// The workgroups of ONE launch carry the layer steps of every tick of a segment (ResidentArgs, rc_internal.h). A workgroup takes the next
// position of the queue, finds its tick and item, requests the item's first weight planes and then waits -- thread 0 polls, the workgroup
// stands at a barrier -- until what the item reads is there:
//   * the second stream's chain (prep -> linear2 -> fuse -> tail, which raises flag_tail behind it) of the tick before the previous one --
//     of the PREVIOUS tick for linear1, which reads what that chain wrote, and for rnn2 behind an init_net state write;
//   * every item of the tick before the previous one (covers every write-after-read two ticks apart: the third copies of h and of
//     relu(linear1) are what makes one tick of slack enough, tests/test_wave_streams.py);
//   * the problems of the previous tick it reads: its own h(t - 1) and its input -- layer 0's h, or relu(linear1), itself an item.
// Everything an item waits for sits EARLIER in the queue or on the second stream, whose kernels never wait for a later item: whoever holds
// an item can finish it, no matter how many workgroups are resident. Waits are bounded (spin_bound): a wait that runs out raises *abort,
// every other wait ends at once and the host reports the segment as failed.
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (The wait stands in FRONT of the item, not behind its first weight requests: everything it holds -- the tick's table entry, the counters'
// addresses -- is dead before the item's registers are live; inside the item it cost 68 bytes of scratch per lane.)
__device__ __forceinline__ void resident_wait(const ResidentArgs& R, const __attribute__((address_space(4))) ResidentTick* T, const int k, const int pi) {
    if (threadIdx.x == 0) {
        const unsigned long long give_up = wall_clock64() + R.spin_bound;
        const int prev_items = k >= 2 ? R.item_base[k - 1] - R.item_base[k - 2] : 0;
        bool ok = false;
        for (;;) {
            ok = ld_agent(R.flag_tail) >= (T->need_tail[pi] ? k : k - 1);
            if (ok && k >= 2) ok = ld_agent(R.tick_done + (k - 2)) >= prev_items;
#pragma unroll
            for (int d = 0; d < 2; ++d)
                if (ok && T->dep[pi][d] >= 0) ok = ld_agent(R.done + (long long)(k - 1) * RC_RES_MAXP + T->dep[pi][d]) >= T->dep_items[pi][d];
            if (ok || ld_agent(R.abort) != 0) break;
            if (wall_clock64() > give_up) { __hip_atomic_store(R.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(kWaves * 64, 1) void rc_gemm_resident_kernel(const ResidentArgs R) {
    __shared__ __attribute__((aligned(1024))) unsigned char ring[kRing * kStage];
    __shared__ int s_rows[256];
    __shared__ int s_cnt[kWaves];
    __shared__ int s_item;
    const int total = R.item_base[R.n_ticks];
    int k = 0;
#ifdef RC_RES_PROF
    __shared__ unsigned long long s_prof[6];      // (in LDS: the kernel has no register to spare)
    enum { p_wait = 0, p_item = 1, p_n = 2, p_rel = 3, p_grab = 4, p_t = 5 };
    if (threadIdx.x == 0) { for (int i = 0; i < 5; ++i) s_prof[i] = 0; s_prof[p_t] = wall_clock64(); }
#define RES_STAMP(ACC) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); s_prof[ACC] += n_ - s_prof[p_t]; s_prof[p_t] = n_; } } while (0)
#else
#define RES_STAMP(ACC) do { } while (0)
#endif
    for (;;) {
        if (threadIdx.x == 0) s_item = __hip_atomic_fetch_add(R.head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int g = __builtin_amdgcn_readfirstlane(s_item);
#ifdef RC_RES_PROF
        if (g >= total) {
            if (threadIdx.x == 0) {
                // (-DRC_RES_PROF builds: five 64-bit sums behind the abort word -- [0] ticks waited for dependencies, [1] in items, [2] items, [3] in the release, [4] taking items)
                unsigned long long* prof = (unsigned long long*)(((unsigned long long)(R.abort + 1) + 7ull) & ~7ull);
                for (int i = 0; i < 5; ++i) atomicAdd(prof + i, s_prof[i]);
            }
            return;
        }
        RES_STAMP(p_grab);
#endif
        if (g >= total) return;
        while (g >= R.item_base[k + 1]) ++k;
        // The tick's table entry is read through the CONSTANT address space (scalar loads into SGPRs on demand, like the kernel arguments of
        // the launch-per-tick kernel): nothing on the device ever writes it. Read as ordinary global memory -- behind this loop's own
        // stores the compiler must assume it changed -- the fields of the problem sat in VGPRs for the length of the item.
        typedef const __attribute__((address_space(4))) ResidentTick* TickC;
        const TickC T = (TickC)(unsigned long long)(R.ticks + k);
        const int item = g - R.item_base[k];
        int pi = 0;
        for (int q = 1; q < T->n; ++q) if (item >= T->p[q].wg_base) pi = q;
        resident_wait(R, T, k, pi);
        RES_STAMP(p_wait);
        typedef const __attribute__((address_space(4))) LdsProblem* ProbC;
        (void)lds_item<RC_RES_MAXP, true, ProbC>(T->p, T->n, T->B, item, ring, s_rows, s_cnt, &pi);
        RES_STAMP(p_item);
#ifdef RC_RES_PROF
        if (threadIdx.x == 0) s_prof[p_n] += 1;
#endif
        // the item's stores (h, c; a writer's half sum went out behind its own flag) are visible before it counts as finished
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(R.done + (long long)k * RC_RES_MAXP + pi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(R.tick_done + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (nothing of this wave is in flight when the next item starts: its K loop counts outstanding vector-memory operations)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        RES_STAMP(p_rel);
    }
}

// second-stream side of the counters: a one-thread kernel raises a flag behind the kernels in front of it, another waits for a counter in
// front of the kernels behind it
__global__ void rc_flag_set_kernel(int* flag, int value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void rc_flag_wait_kernel(const int* counter, int target, int* abort, unsigned long long spin_bound) {
    const unsigned long long give_up = wall_clock64() + spin_bound;
    while (ld_agent(counter) < target && ld_agent(abort) == 0) {
        if (wall_clock64() > give_up) { __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

void rc_launch_gemm_resident(const ResidentArgs& R, int workgroups, hipStream_t s) {
    hipLaunchKernelGGL(rc_gemm_resident_kernel, dim3(workgroups), dim3(kWaves * 64), 0, s, R);
}
void rc_launch_flag_set(int* flag, int value, hipStream_t s) { hipLaunchKernelGGL(rc_flag_set_kernel, dim3(1), dim3(1), 0, s, flag, value); }
void rc_launch_flag_wait(const int* counter, int target, int* abort, unsigned long long spin_bound, hipStream_t s) {
    hipLaunchKernelGGL(rc_flag_wait_kernel, dim3(1), dim3(1), 0, s, counter, target, abort, spin_bound);
}

void rc_launch_gemm_lds(const LdsLaunch& L, int total_wg, hipStream_t s, hipEvent_t stop) {
    const dim3 g(total_wg), b(kWaves * 64);
    if (stop) hipExtLaunchKernelGGL(rc_gemm_lds_kernel, g, b, 0, s, nullptr, stop, 0, L);
    else hipLaunchKernelGGL(rc_gemm_lds_kernel, g, b, 0, s, L);
}
