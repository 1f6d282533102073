// Gate GEMM of the sig_mp sub-nets on CDNA4 (gfx950): y = epilogue(A[rows, K] * W^T + b) in exact fp32.
//
// Replaces, per step of one sub-net, the torch ops of f(i, x) (net/sig_mp.py:126-129):
//   linear1 + ReLU, aten::lstm (two layers; CPU path = oneDNN mkldnn_rnn_layer), linear2
// and rnn2.init_net (articulate/utils/torch/rnn.py:195-201).
//
// Shape regime: M = batch rows (256..1024, skinny), K = 2H (1024..2560), N = 4H. The grid is made of
// (16*MR)-row x (16*NC)-column tiles (shapes: gemm_tile below); a tile's K range is split over the 4 waves of its
// workgroup so that every LSTM layer yields 256 workgroups at batch 256 (one per CU; one workgroup per CU is also
// all that is resident -- see RC_LDS_FLOATS). Each wave streams ITS slice of the weights straight from
// L2/HBM into VGPRs as 1 KiB coalesced dwordx4 loads (weights are pre-packed in MFMA-B fragment order, nothing
// is shared between waves so LDS staging would be pure overhead) and feeds v_mfma_f32_16x16x4_f32 -- bitwise an
// fp32 fma chain, which is what keeps the 1e-4 parity budget. Partial sums meet in LDS; the epilogue applies the
// bias and, for LSTM layers, the gate non-linearities and the (c, h) update, so gates never touch HBM.
// The 8 row tiles that share a weight slice are mapped to the same XCD (block id % 8) to share it through L2.
//
// Row selection is stateless: every workgroup derives the compacted list of active rows from the per-row flag
// bytes itself (one ballot per wave), so masked sub-steps (vision branch on/off) cost only the rows they touch.
#include "rc_internal.h"
#include <hip/hip_ext.h>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DRC_TRACE_TILES (tools/tile_trace.py): every workgroup stores {block, tile shape | rows, CU, tile, 4 wall-clock
// stamps} into slot (launch base + block id) of a host-provided buffer -- per-CU timelines of a launch, no atomics.
// Not compiled into the product library.
#ifdef RC_TRACE_TILES
__device__ unsigned long long* g_trace_buf = nullptr;
__device__ unsigned long long g_trace_cap = 0;
extern "C" int rc_trace_tiles_set(unsigned long long* buf, unsigned long long cap_records) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cap), &cap_records, sizeof(cap_records)) == hipSuccess ? 0 : -1;
}
#define TRACE_T(i) do { if (threadIdx.x == 0) trace_t[i] = wall_clock64(); } while (0)
#else
#define TRACE_T(i) do { } while (0)
#endif

#ifndef LDS_PAD
#define LDS_PAD 16            // floats added to a partial-sum row: epilogue rows land on different banks
#endif
#define RC_LDS_HEAD 192       // ints in front of the partial sums: active rows of the tile [128], per-wave counts [4]

#include "rc_gates.h"
__device__ __forceinline__ float sigmoidf_(float x) { return rc_gate_sigmoid(x); }
__device__ __forceinline__ float tanhf_(float x) { return rc_gate_tanh(x); }

template <int MR, int NC>
struct Frag {                 // one chunk (16 k): a float4 per lane for each of the MR row blocks and NC column blocks
    f32x4 a[MR];
    f32x4 b[NC];
};

template <int MR, int NC, bool NTL = false>
__device__ __forceinline__ void load_chunk(Frag<MR, NC>& f, const float* const (&pa)[MR], long long aoff, const float* pb,
                                           long long bstride) {
#pragma unroll
    for (int r = 0; r < MR; ++r)
        f.a[r] = *reinterpret_cast<const f32x4*>(pa[r] + aoff);
#pragma unroll
    for (int j = 0; j < NC; ++j)
        {
            if constexpr (NTL) f.b[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + j * bstride));
            else f.b[j] = *reinterpret_cast<const f32x4*>(pb + j * bstride);
        }
}

template <int MR, int NC>
__device__ __forceinline__ void mma_chunk(const Frag<MR, NC>& f, f32x4 (&acc)[MR][NC]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r][s], f.b[j][s], acc[r][j], 0, 0, 0);
            }
        }
    }
}

// ---- split-bf16 products (GemmLaunch.split) -----------------------------------------------------------------------------
// An fp32 value is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significant bits, truncation split: hi = top 16 bits of
// a, mid = top 16 bits of a - hi, lo = a - hi - mid), and a bf16 x bf16 product is exact in fp32. The gate GEMM then runs on
// v_mfma_f32_16x16x32_bf16 (17 cycles per 16x16x32 block against 8 x 32 cycles of v_mfma_f32_16x16x4_f32) as RC_SPLIT_PRODUCTS
// partial products per block pair with fp32 accumulation: 6 keep every term down to 2^-16 of the product (hi.hi, hi.mid,
// mid.hi, mid.mid, hi.lo, lo.hi; what is dropped is <= 2^-23 of a product, below the fp32 rounding of the running sum that
// the fp32 instruction makes as well), 9 keep all of them (the products are then exact; only the order of the fp32
// additions differs from an fma chain). Weights are split once on the host (three bf16 planes, 6 B per weight), activations
// on the fly (they stay fp32 everywhere else). One k-block = 32 k = two 16-k chunks of the rc_pk layout: lane (kq, i) holds
// k = 32 kb + {4 kq .. 4 kq + 3} and 32 kb + 16 + {4 kq .. 4 kq + 3} -- the same 8 k for the A and the B operand.
#ifndef RC_SPLIT_PRODUCTS
#define RC_SPLIT_PRODUCTS 6
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef RC_SPLIT_W32
#define RC_SPLIT_W32 0        // 1: the weights stream as fp32 too (the fp32 packing, 4 B instead of 6 B per weight) and are split in the K
#endif                        // loop like the activations: fewer operand bytes per MFMA for 36 more VALU per column block and k-block
#define RC_WPL (RC_SPLIT_W32 ? 2 : 3)          // 1-KiB pieces per column block and k-block: two fp32 chunks, or three bf16 planes
// W32 (round 5: a template parameter as well as the build macro): the launches whose weight slices have ONE reader -- every problem a single
// 64-row tile per slice, i.e. contexts of 48-64 rows -- are pure weight streams, and 4 B per weight beat 6 B + no split (measured in round 4 with
// the macro: batch 48 +13 %, 64 +7 %; at 256 rows, four readers per slice through the L2, -7 %). Bitwise the same products either way.
template <int MR, int NC, bool W32 = (RC_SPLIT_W32 != 0)>
struct FragS {                // one k-block (32 k): fp32 activations (two float4 per row block), three weight planes per column block
    static constexpr int WPL = W32 ? 2 : 3;
    f32x4 a0[MR], a1[MR];
    u32x4 b[NC][WPL];
};

template <int MR, int NC, bool W32>
__device__ __forceinline__ void load_kblock(FragS<MR, NC, W32>& f, const float* const (&pa)[MR], long long aoff, const u32x4* pb,
                                            long long bstride) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        f.a0[r] = *reinterpret_cast<const f32x4*>(pa[r] + aoff);
        f.a1[r] = *reinterpret_cast<const f32x4*>(pa[r] + aoff + 256);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int p = 0; p < (W32 ? 2 : 3); ++p) f.b[j][p] = pb[j * bstride + p * 64];
}

// a = hi + mid + lo exactly; each output packs 8 bf16 (element e in the low / high half of dword e / 2).
// Two elements per step on float2 values: the two subtractions of a pair compile to one v_pk_add_f32 each (9 VALU ops per
// pair instead of 11: the K loop issues its VALU work beside the MFMAs at 2-3 cycles per instruction).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
    const f32x2 a[4] = {f32x2{x0[0], x0[1]}, f32x2{x0[2], x0[3]}, f32x2{x1[0], x1[1]}, f32x2{x1[2], x1[3]}};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const u32x2 ua = __builtin_bit_cast(u32x2, a[d]);
        const f32x2 r1 = a[d] - __builtin_bit_cast(f32x2, ua & 0xffff0000u);          // a - hi, exact
        const u32x2 um = __builtin_bit_cast(u32x2, r1);
        const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, um & 0xffff0000u);            // a - hi - mid, exact
        const u32x2 ul = __builtin_bit_cast(u32x2, r2);
        // bytes {hi[3], hi[2], lo[3], lo[2]} of the pair = the two truncated bf16
        h[d] = __builtin_amdgcn_perm(ua[1], ua[0], 0x07060302u);
        m[d] = __builtin_amdgcn_perm(um[1], um[0], 0x07060302u);
        l[d] = __builtin_amdgcn_perm(ul[1], ul[0], 0x07060302u);
    }
}

template <int MR, int NC, bool W32>
__device__ __forceinline__ void load_kblock_b(FragS<MR, NC, W32>& f, const u32x4* pb, long long bstride) {
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int p = 0; p < (W32 ? 2 : 3); ++p) f.b[j][p] = pb[j * bstride + p * 64];
}
template <int MR, int NC, bool W32>
__device__ __forceinline__ void load_kblock_a(FragS<MR, NC, W32>& f, const float* const (&pa)[MR], long long aoff) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        f.a0[r] = *reinterpret_cast<const f32x4*>(pa[r] + aoff);
        f.a1[r] = *reinterpret_cast<const f32x4*>(pa[r] + aoff + 256);
    }
}

template <int MR, int NC, bool W32>
__device__ __forceinline__ void mma_kblock(const FragS<MR, NC, W32>& f, f32x4 (&acc)[MR][NC]) {
    u32x4 wp[3][NC];              // the column blocks' planes: hi, mid, lo
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        if constexpr (W32) split3(__builtin_bit_cast(f32x4, f.b[j][0]), __builtin_bit_cast(f32x4, f.b[j][1]), wp[0][j], wp[1][j], wp[2][j]);
        else { wp[0][j] = f.b[j][0]; wp[1][j] = f.b[j][1]; wp[2][j] = f.b[j][2]; }
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        u32x4 uh, um, ul;
        split3(f.a0[r], f.a1[r], uh, um, ul);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, uh), am = __builtin_bit_cast(bf16x8, um), al = __builtin_bit_cast(bf16x8, ul);
        // small terms first; consecutive MFMAs go to different accumulators (NC of them between two uses of one)
#define RC_PROD(AV, PL)                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NC; ++j)                                                                        \
        acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AV, __builtin_bit_cast(bf16x8, wp[PL][j]), acc[r][j], 0, 0, 0);
#if RC_SPLIT_PRODUCTS == 9
        RC_PROD(al, 2) RC_PROD(am, 2) RC_PROD(al, 1)
#endif
        RC_PROD(al, 0) RC_PROD(ah, 2) RC_PROD(am, 1) RC_PROD(am, 0) RC_PROD(ah, 1) RC_PROD(ah, 0)
#undef RC_PROD
    }
}

// One workgroup = one (16*MR)-row x (16*NC)-column tile, K split over the RC_NW waves. Tile shapes are chosen so that
// every LSTM layer of every net is exactly 256 workgroups at batch 256 (one per CU), with as few operand bytes per
// MFMA as the register file allows -- measured on MI355X, a CU delivers only ~256 B of operands per 64 MFMA cycles
// while the matrix pipe is busy (profiles/r01_gemm_probe.txt):
//   MR x NC = 4 x 5 : 64 rows x 20 units (rnn4, H = 1280)    (4 + 5) KiB per 80 MFMAs  = 230 B per 64 cycles
//             4 x 4 : 64 rows x 16 units (rnn6, H = 1024)    (4 + 4) KiB per 64 MFMAs  = 256 B
//             2 x 4 : 32 rows x 16 units (H = 512, dense)    (2 + 4) KiB per 32 MFMAs  = 384 B
//             2 x 8 / 2 x 10: 32-row variants of the big nets (the ones in use: see rc_api.cpp)
//             1 x 1 : 16 rows x 4 units: few-row stages (regime transitions, batch 1): 320/256/128 tiles per layer so
//                     that every CU streams a slice of the weights (a 32 x 160 tile would leave 224 CUs idle)
//             1 x 2 : 16 rows x 32 columns for linear2 (N = 2..144): 3x the workgroups of a 32 x 64 tile -- these
//                     launches are parallelism-starved (16-48 workgroups), not bandwidth- or latency-bound
// v_mfma_f32_16x16x4_f32 instead of 32x32x2: same rate, half the accumulator traffic, measured -14 % time.
// PIPE pins a software pipeline (loads of chunk q+1 issued before the MFMAs of chunk q) with sched_barrier;
// without it hipcc issues both chunks' loads at the top of an iteration and drains them inside it.
template <int MR, int NC, int D, bool PIPE, bool SPLIT = false, bool DEEPOK = true, bool NTW = false, bool W32 = (RC_SPLIT_W32 != 0)>
__device__ __forceinline__ void gemm_tile(const GemmProblem& P, const int B, const int m_tile0, const int n_tile, float* s_mem) {
    constexpr int MT = 16 * MR, NT = 16 * NC, UT = 4 * NC, LD = NT + (MR >= 8 ? 4 : LDS_PAD);   // 128-row tiles: 140 KB with pad 4
#ifdef RC_TRACE_TILES
    unsigned long long trace_t[4] = {0, 0, 0, 0};
    bool traced = false;
#endif
  // Row tiles m_tile0, m_tile0 + m_tiles, ... : stages that expect only a few active rows launch a single row tile per
  // column tile and still cover any number of rows (a full grid of row tiles would mostly be workgroups that scan the
  // flags and exit: 9,216 of them per launch at batch 256 with 16-row tiles).
  for (int m_tile = m_tile0;; m_tile += P.m_tiles) {
    int* s_rows = reinterpret_cast<int*>(s_mem);                   // [MT <= 128]
    int* s_cnt = s_rows + 128;                                      // [RC_NW] (+ padding to RC_LDS_HEAD ints)
    float* s_part = s_mem + RC_LDS_HEAD;                            // [RC_NW][MT][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    TRACE_T(0);
    // ---- active rows of this tile -------------------------------------------------------------------------
    const int lo = m_tile * MT;
    int nrows;
    if (P.flag_bit == 0) {
        nrows = min(MT, B - lo);
        if (nrows <= 0) return;
        __syncthreads();                                            // previous row tile done with s_rows / s_part
        if (tid < MT) s_rows[tid] = lo + min(tid, nrows - 1);       // MT <= 128 < 256 threads
        __syncthreads();
    } else {
        int total = 0;
        __syncthreads();                                            // previous row tile done with s_rows / s_part
        for (int base = 0; base < B && total < lo + MT; base += RC_NW * 64) {
            const int r = base + tid;
            const bool f = r < B && (P.flags[r] & P.flag_bit);
            const unsigned long long bal = __ballot(f);
            if (lane == 0) s_cnt[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0, sum = 0;
#pragma unroll
            for (int w = 0; w < RC_NW; ++w) {
                const int cw = s_cnt[w];
                woff += (w < wave) ? cw : 0;
                sum += cw;
            }
            const int idx = total + woff + __popcll(bal & ((1ull << lane) - 1ull));
            if (f && idx >= lo && idx < lo + MT) s_rows[idx - lo] = r;
            total += sum;
            __syncthreads();
        }
        nrows = min(MT, total - lo);
        if (nrows <= 0) return;
        if (tid < MT && tid >= nrows) s_rows[tid] = s_rows[0];
        __syncthreads();
    }
    if (P.open_step && n_tile == 0 && tid < nrows) P.steps[s_rows[tid]] += 1;

    // ---- K loop: wave `wave` owns chunks [wave*Qw, (wave+1)*Qw) -------------------------------------------
    const int i = lane & 15, kq = lane >> 4;
    // per-lane A pointers of the MR 16-row blocks (rows may come from anywhere in the batch after compaction)
    const float* pa0[MR];
    const float* pa1[MR];
    int row_r[MR], st_r[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        row_r[r] = s_rows[16 * r + i];
        st_r[r] = (P.seg[0].par_mode | P.seg[1].par_mode) ? P.steps[row_r[r]] : 0;
    }
    // split products: the first k-blocks' weight planes depend on nothing the prologue computes -- requested here, behind the
    // step-counter reads (vmcnt completes in order) and in front of everything that waits for those, they travel while the
    // activation pointers are formed (a tile waits ~3 us for its first weights: profiles/r02_kloop_ablation.txt)
    FragS<MR, NC, W32> fa = {}, fb = {};
    const int Qs = P.Kp / 32, Qws = Qs / RC_NW;                     // k-blocks per wave (K' % 128 == 0 -> >= 1)
    constexpr int WPL = W32 ? 2 : 3;
    constexpr int KBU = WPL * 64;                                   // uint4 per column block and k-block (planes, or fp32 chunks, x 64 lanes)
    const long long bs = (long long)Qs * KBU;                       // uint4 between consecutive 16-column blocks
    const u32x4* pbs = reinterpret_cast<const u32x4*>(W32 ? (const void*)P.W : P.Ws) + ((long long)(n_tile * NC) * Qs + (long long)wave * Qws) * KBU + lane;
    constexpr bool DEEP = SPLIT && DEEPOK && MR >= 2 && (MR * 8 + NC * 4 * WPL) * 3 + (W32 ? NC * 12 : 0) + MR * NC * 4 <= 380;   // 2x4, 4x4, 4x5 (16-row tiles: 128-VGPR budget)
    if constexpr (SPLIT) {
        load_kblock_b(fa, pbs, bs);
        if constexpr (DEEP) load_kblock_b(fb, pbs + (long long)min(1, Qws - 1) * KBU, bs);
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int row = row_r[r];
        const int st = st_r[r] + ((P.seg[0].par_mode | P.seg[1].par_mode) ? P.step_off : 0);
        const float* pp[2];
#pragma unroll
        for (int sgi = 0; sgi < 2; ++sgi) {
            const GemmSeg& sg = P.seg[sgi];
            const int par = sg.par_mode == RC_PAR_SRC ? ((st - 1) % RC_HBUF) : (sg.par_mode == RC_PAR_DST ? (st % RC_HBUF) : 0);
            const float* base = sg.base;
            if (sgi == 0 && P.sel_bit && !(P.sel_flags[row] & P.sel_bit)) base = P.alt_base;
            pp[sgi] = base + (long long)par * sg.par_stride + rc_pk(row, 4 * kq, sg.ld);
        }
        pa0[r] = pp[0];
        pa1[r] = pp[1];
    }
    const int Q = P.Kp / RC_KC, Qw = Q / RC_NW;                     // chunks per wave: even (K' % 128 == 0)
    const int K0 = P.seg[0].K;
    const long long bstride = (long long)Q * 256;                   // floats between consecutive 16-column blocks
    const float* pb = P.W + ((long long)(n_tile * NC) * Q + (long long)wave * Qw) * 256 + lane * 4;
    const int kbase = wave * Qw * RC_KC;

    TRACE_T(1);
    f32x4 acc[MR][NC];
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define LOADC(F, QI)                                                                                      \
    do {                                                                                                  \
        const int k_ = kbase + (QI) * RC_KC;                                                              \
        if (k_ < K0) load_chunk<MR, NC, NTW>(F, pa0, (long long)k_ * 16, pb + (long long)(QI) * 256, bstride);  \
        else load_chunk<MR, NC, NTW>(F, pa1, (long long)(k_ - K0) * 16, pb + (long long)(QI) * 256, bstride);   \
    } while (0)
#define SB() do { if (PIPE) __builtin_amdgcn_sched_barrier(0); } while (0)
    // The steady-state loop has NO conditionals on the load path other than the wave-uniform segment select: with a
    // conditional prefetch hipcc (ROCm 7.2) waits vmcnt(0) in front of the MFMAs.
    // Software pipeline over D chunk buffers: D - 1 chunks are in flight behind every MFMA block. The steady-state
    // loop has NO conditionals other than the wave-uniform segment select (with a conditional prefetch hipcc waits
    // vmcnt(0) in front of the MFMAs); prefetch indices past the end are clamped (a redundant, valid load) and only
    // the remainder (< D chunks, already loaded) is predicated. D = 2 for the wide tiles (their 64-80 MFMAs per chunk
    // cover the latency), D = 8 for the 16-row tiles whose 4-8 MFMAs per chunk do not.
    if constexpr (SPLIT) {              // split-bf16 products: k-blocks of 32, two named buffers for every tile shape
        const int kb0 = wave * Qws * 32;
#define LOADS_A(F, QI)                                                                                              \
    do {                                                                                                            \
        const int k_ = kb0 + (QI) * 32;                                                                             \
        if (k_ < K0) load_kblock_a(F, pa0, (long long)k_ * 16);                                        \
        else load_kblock_a(F, pa1, (long long)(k_ - K0) * 16);                                              \
    } while (0)
#define LOADS(F, QI)                                                                                                \
    do {                                                                                                            \
        const int k_ = kb0 + (QI) * 32;                                                                             \
        if (k_ < K0) load_kblock(F, pa0, (long long)k_ * 16, pbs + (long long)(QI) * KBU, bs);              \
        else load_kblock(F, pa1, (long long)(k_ - K0) * 16, pbs + (long long)(QI) * KBU, bs);               \
    } while (0)
        // With the MFMA time cut 2.7x the K loop is bound by what a wave keeps in flight (one k-block = 23 KiB for a 64 x 80
        // tile; 4 waves x 23 KiB / ~2 us of L2 / fabric latency = the 47 GB/s per CU the two-buffer loop was measured at):
        // tile shapes whose registers allow it run THREE k-block buffers (two blocks in flight behind every MFMA block).
        int q = 0;
        // (Spreading the loads between the MFMAs -- sched_group_barrier patterns, or slices of the next block's loads in front of
        // every row block's MFMAs -- was measured at -2 % / +-1 %: profiles/r02_kloop_ablation.txt.)
#define STEP(FL, QL, FM) do { LOADS(FL, QL); SB(); mma_kblock(FM, acc); SB(); } while (0)
        if constexpr (DEEP) {
            FragS<MR, NC, W32> fc = {};
            LOADS_A(fa, 0);
            LOADS_A(fb, min(1, Qws - 1));
            for (; q + 3 <= Qws; q += 3) {  // prefetch indices past the end are clamped: a redundant, valid load, no branch
                STEP(fc, min(q + 2, Qws - 1), fa);
                STEP(fa, min(q + 3, Qws - 1), fb);
                STEP(fb, min(q + 4, Qws - 1), fc);
            }
            if (q < Qws) mma_kblock(fa, acc);
            if (q + 1 < Qws) mma_kblock(fb, acc);
        } else {
            LOADS_A(fa, 0);
            for (; q + 2 <= Qws; q += 2) {
                STEP(fb, min(q + 1, Qws - 1), fa);
                STEP(fa, min(q + 2, Qws - 1), fb);
            }
            if (q < Qws) mma_kblock(fa, acc);
        }
#undef STEP
#undef LOADS_A
#undef LOADS
    } else if constexpr (D == 2) {      // wide tiles: two named buffers (the array form below schedules worse here)
        Frag<MR, NC> fa = {}, fb = {};
        int q = 0;
        LOADC(fa, 0);
        for (; q + 2 < Qw; q += 2) {
            LOADC(fb, q + 1);
            SB();
            mma_chunk<MR, NC>(fa, acc);
            SB();
            LOADC(fa, q + 2);
            SB();
            mma_chunk<MR, NC>(fb, acc);
            SB();
        }
        LOADC(fb, q + 1);
        SB();
        mma_chunk<MR, NC>(fa, acc);
        SB();
        mma_chunk<MR, NC>(fb, acc);
    } else {
        Frag<MR, NC> f[D];
    #pragma unroll
        for (int d = 0; d < D; ++d) f[d] = Frag<MR, NC>{};
    #pragma unroll
        for (int d = 0; d < D - 1; ++d) LOADC(f[d], min(d, Qw - 1));
        int q = 0;
        for (; q + D <= Qw; q += D) {
    #pragma unroll
            for (int d = 0; d < D; ++d) {
                LOADC(f[(d + D - 1) % D], min(q + d + D - 1, Qw - 1));
                SB();
                mma_chunk<MR, NC>(f[d], acc);
                SB();
            }
        }
        const int rem = Qw - q;
    #pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (d < rem) mma_chunk<MR, NC>(f[d], acc);
    }
#undef LOADC
#undef SB

    TRACE_T(2);
    // The LSTM epilogue's global reads (previous cell state, step parity of the row) are requested here, in front of the
    // reduction: issued inside the item loop each waited a full L2 round trip behind the previous item's state stores
    // (measured: 0.85 us per item, 4-7 us per tile).
    constexpr int EPI_ITEMS = (MT * UT + RC_NW * 64 - 1) / (RC_NW * 64);
    float c_prev[EPI_ITEMS];
    int st_row[EPI_ITEMS];
    // The bias as well: a global load inside the item loop waits (vmcnt counts stores too) for the previous item's two state stores
    // to be acknowledged -- 0.5 us per item, 8 items per thread in a 64 x 128 tile. A thread's items share their unit (and so their
    // bias) whenever the workgroup size is a multiple of the units per tile: one float4 then, not one per item (registers).
    constexpr bool BIAS_ONE = (RC_NW * 64) % UT == 0;             // (64 x 80 and 32 x 160 tiles: the load stays in the loop)
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (P.epi == RC_EPI_LSTM) {
#pragma unroll
        for (int k = 0; k < EPI_ITEMS; ++k) {
            const int item = tid + k * RC_NW * 64;
            const int rr = item / UT, u = item - rr * UT;
            const bool ok = item < MT * UT && rr < nrows;
            const int r2 = s_rows[ok ? rr : 0];
            c_prev[k] = ok ? P.cstate[(long long)r2 * P.H + n_tile * UT + u] : 0.f;
            st_row[k] = ok ? P.steps[r2] : 0;
        }
    }
    // ---- split-K reduction through LDS (C layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg) --------
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s_part[(wave * MT + 16 * r + 4 * kq + e) * LD + 16 * j + i] = acc[r][j][e];
    if (P.epi == RC_EPI_LSTM) {       // (behind the accumulators' last use: their registers are free; the barrier covers the latency)
        if constexpr (BIAS_ONE) bias4 = *reinterpret_cast<const f32x4*>(&P.bias[n_tile * NT + 4 * (tid % UT)]);
    }
    __syncthreads();

    if (P.epi == RC_EPI_LSTM) {
        // item -> (row rr, unit u); every 16-column block holds 4 hidden units x (i, f, g, o) -- the four gates of a unit
        // in four consecutive columns -- so the weight packing does not depend on the tile width (the host picks NC per
        // launch) and an item reads each wave's partial sums, and the bias, as ONE 16-byte access
#pragma unroll
        for (int k = 0; k < EPI_ITEMS; ++k) {
            const int item = tid + k * RC_NW * 64;
            const int rr = item / UT, u = item - rr * UT;
            if (item >= MT * UT || rr >= nrows) continue;
            const int unit = n_tile * UT + u;
            // (p0 + p1) + (p2 + p3): the two halves of K -- seg[0] | seg[1] of a layer step -- are summed on their own first, which is
            // what lets rc_gemm_lds.hip compute them in two workgroups and still produce these bits (round 6)
            static_assert(RC_NW == 4, "the reduction order below names the four K quarters");
            f32x4 g4 = (*reinterpret_cast<const f32x4*>(&s_part[rr * LD + 4 * u]) + *reinterpret_cast<const f32x4*>(&s_part[(MT + rr) * LD + 4 * u])) +
                       (*reinterpret_cast<const f32x4*>(&s_part[(2 * MT + rr) * LD + 4 * u]) + *reinterpret_cast<const f32x4*>(&s_part[(3 * MT + rr) * LD + 4 * u]));
            if constexpr (BIAS_ONE) g4 += bias4; else g4 += *reinterpret_cast<const f32x4*>(&P.bias[n_tile * NT + 4 * u]);
            const int r2 = s_rows[rr];
            const int dst = (st_row[k] + P.step_off) % RC_HBUF;
            const long long ci = (long long)r2 * P.H + unit;
            float cn, hn;
            rc_lstm_cell(g4[0], g4[1], g4[2], g4[3], c_prev[k], cn, hn);
            P.cstate[ci] = cn;
            P.hstate[(long long)dst * P.h_par_stride + rc_pk(r2, unit, P.H)] = hn;
        }
    } else {
        // packed outputs whose columns come in whole groups of four (linear1 -> x1: the A operand of the LSTM's first layer):
        // four columns per item, 16-byte LDS reads and one 16-byte store -- the same sums in the same order
        const bool vec4 = P.out_packed && P.out_bit == 0 && ((P.out_col0 | P.N) & 3) == 0;
        if (vec4) {
            // the items' bias first (every global load behind a store waits for that store: see the LSTM epilogue above); one float4
            // where a thread's items share their four columns
            constexpr int V4_ITEMS = (MT * (NT / 4) + RC_NW * 64 - 1) / (RC_NW * 64);
            constexpr bool B4_ONE = (RC_NW * 64) % (NT / 4) == 0;
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (B4_ONE) b4 = *reinterpret_cast<const f32x4*>(&P.bias[n_tile * NT + (tid % (NT / 4)) * 4]);
#pragma unroll
            for (int k = 0; k < V4_ITEMS; ++k) {
                const int item = tid + k * RC_NW * 64;
                const int rr = item / (NT / 4), c4 = (item - rr * (NT / 4)) * 4;
                const int n = n_tile * NT + c4;
                if (item >= MT * (NT / 4) || rr >= nrows || n >= P.N) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(&s_part[rr * LD + c4]);
#pragma unroll
                for (int w = 1; w < RC_NW; ++w) v += *reinterpret_cast<const f32x4*>(&s_part[(w * MT + rr) * LD + c4]);
                if constexpr (B4_ONE) v += b4; else v += *reinterpret_cast<const f32x4*>(&P.bias[n]);
                if (P.epi == RC_EPI_RELU) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
                *reinterpret_cast<f32x4*>(&P.out[rc_pk(s_rows[rr], P.out_col0 + n, P.ldo)]) = v;
            }
        } else
        for (int item = tid; item < MT * NT; item += RC_NW * 64) {
            const int rr = item / NT, col = item - rr * NT;
            if (rr >= nrows) continue;
            const int n = n_tile * NT + col;
            float v = s_part[rr * LD + col];
#pragma unroll
            for (int w = 1; w < RC_NW; ++w) v += s_part[(w * MT + rr) * LD + col];
            v += P.bias[n];
            if (P.epi == RC_EPI_RELU) v = fmaxf(v, 0.0f);
            const int r2 = s_rows[rr];
            if (n < P.N && (P.out_bit == 0 || (P.out_flags[r2] & P.out_bit))) {
                P.out[P.out_packed ? rc_pk(r2, P.out_col0 + n, P.ldo) : (long long)r2 * P.ldo + P.out_col0 + n] = v;
            }
        }
    }
#ifdef RC_TRACE_TILES
    if (threadIdx.x == 0 && !traced && g_trace_buf) {
        traced = true;
        const unsigned long long idx = (unsigned long long)P.trace_base + blockIdx.x;
        if (idx < g_trace_cap) {
            unsigned long long* r = g_trace_buf + idx * 8;
            r[0] = (unsigned long long)blockIdx.x + 1; r[1] = (unsigned long long)(MR * 16 + NC) | ((unsigned long long)nrows << 16);
            r[2] = __smid(); r[3] = (unsigned long long)n_tile | ((unsigned long long)m_tile << 32);
            r[4] = trace_t[0]; r[5] = trace_t[1]; r[6] = trace_t[2]; r[7] = wall_clock64();
        }
    }
#endif
  }   // row-tile loop
}

#ifndef RC_LDS_FLOATS
#define RC_LDS_FLOATS (RC_LDS_HEAD + RC_NW * 64 * (16 * 8 + LDS_PAD))   // 148 KB = the 4 x 8 tile (8 x 4: 140 KB, 4 x 5: 97 KB, 2 x 10: 91 KB)
#endif

#ifndef RC_WPS
#define RC_WPS 1   // waves per SIMD the register allocation must allow
#endif
// which problem and which tile of it this workgroup owns; false when the (XCD-padded) grid slot is empty
__device__ __forceinline__ bool locate_tile(const GemmLaunch& L, int& pi, int& m_tile, int& n_tile) {
    pi = 0;
#pragma unroll
    for (int q = 1; q < RC_MAX_PROB; ++q)
        if (q < L.n && (int)blockIdx.x >= L.p[q].wg_base) pi = q;
    const GemmProblem& P = L.p[pi];
    const int local = blockIdx.x - P.wg_base;
    if ((P.n_tiles & 7) == 0) {   // XCD-aware: the row tiles of one weight slice share block-id % 8
        const int xcd = local & 7, s = local >> 3;
        m_tile = s % P.m_tiles;
        n_tile = (s / P.m_tiles) * 8 + xcd;
    } else {
        m_tile = local % P.m_tiles;
        n_tile = local / P.m_tiles;
    }
    return n_tile < P.n_tiles;
}

template <bool SPLIT>
__device__ __forceinline__ void wide_tiles(const GemmLaunch& L, float* s_mem) {
    int pi, m_tile, n_tile;
    if (!locate_tile(L, pi, m_tile, n_tile)) return;
    const GemmProblem& P = L.p[pi];
    switch (P.mr * 16 + P.nc) {
        case 8 * 16 + 4: gemm_tile<8, 4, 2, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 4 * 16 + 8: gemm_tile<4, 8, 2, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 4 * 16 + 5: gemm_tile<4, 5, 2, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 4 * 16 + 4: gemm_tile<4, 4, 2, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 2 * 16 + 10: gemm_tile<2, 10, 2, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 2 * 16 + 8: gemm_tile<2, 8, 2, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 1 * 16 + 2: gemm_tile<1, 2, 8, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 1 * 16 + 1: gemm_tile<1, 1, 8, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        default: gemm_tile<2, 4, 2, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
    }
}

__global__ __launch_bounds__(RC_NW * 64, RC_WPS) void rc_gemm_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_LDS_FLOATS];
    wide_tiles<false>(L, s_mem);
}
// the same tiles with split-bf16 products (GemmLaunch.split; see mma_kblock)
__global__ __launch_bounds__(RC_NW * 64, RC_WPS) void rc_gemm_split_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_LDS_FLOATS];
    wide_tiles<true>(L, s_mem);
}

// Split-product launches of 64 x 128 tiles in which every weight slice has ONE reader (a single row tile per problem: contexts of 33-64
// rows on the wavefront engine): the weights stream as fp32 and are split in the K loop (W32, above).
__global__ __launch_bounds__(RC_NW * 64, RC_WPS) void rc_gemm_split48_w32_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_LDS_FLOATS];
    int pi, m_tile, n_tile;
    if (!locate_tile(L, pi, m_tile, n_tile)) return;
    gemm_tile<4, 8, 2, true, true, true, false, true>(L.p[pi], L.B, m_tile, n_tile, s_mem);
}

// Launches whose tiles are all at most 32 x 64 (the linear1 launches; LSTM stages of batches below 128) do not need the
// 148 KB of the wide kernel: with 41 KB of LDS and a register budget for two waves per SIMD three of these workgroups
// share a CU, so the 328 short tiles of a linear1 launch no longer take two rounds of 256.
#define RC_MID_LDS_FLOATS (RC_LDS_HEAD + RC_NW * 32 * (16 * 4 + LDS_PAD))
template <bool SPLIT>
__device__ __forceinline__ void mid_tiles(const GemmLaunch& L, float* s_mem) {
    int pi, m_tile, n_tile;
    if (!locate_tile(L, pi, m_tile, n_tile)) return;
    const GemmProblem& P = L.p[pi];
    switch (P.mr * 16 + P.nc) {
        case 1 * 16 + 2: gemm_tile<1, 2, 8, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        case 1 * 16 + 1: gemm_tile<1, 1, 8, true, SPLIT>(P, L.B, m_tile, n_tile, s_mem); break;
        default: gemm_tile<2, 4, 2, true, SPLIT, false>(P, L.B, m_tile, n_tile, s_mem); break;   // two buffers: 256-register budget
    }
}
__global__ __launch_bounds__(RC_NW * 64, 2) void rc_gemm_mid_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_MID_LDS_FLOATS];
    mid_tiles<false>(L, s_mem);
}
__global__ __launch_bounds__(RC_NW * 64, 2) void rc_gemm_mid_split_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_MID_LDS_FLOATS];
    mid_tiles<true>(L, s_mem);
}

// Launches whose problems all use 16-row tiles (batch <= 16: live mode, transition rows) are weight-streaming, not
// MFMA-bound: their own kernel with a 12 KB LDS footprint and a register budget for 4 waves per SIMD keeps four
// workgroups -- 4 x 32 KB of weight loads in flight -- on every CU instead of one.
// (Tried and dropped for sequence mode: running such launches on a second stream BESIDE a wide-tile launch. Four of these
// workgroups fill a CU's register file and lock the wide tiles' 300-register waves out; a variant with its LDS padded to
// 60 KB, so that only one fits beside a wide-tile workgroup, still stretched a 116 us rnn4 launch to 150-220 us.)
#define RC_SMALL_LDS_FLOATS (RC_LDS_HEAD + RC_NW * 16 * (16 * 2 + LDS_PAD))
template <bool SPLIT, bool NTW = false>
__device__ __forceinline__ void small_tiles(const GemmLaunch& L, float* s_mem) {
    int pi, m_tile, n_tile;
    if (!locate_tile(L, pi, m_tile, n_tile)) return;
    const GemmProblem& P = L.p[pi];
    if constexpr (NTW) {       // per problem: weights that should stay in the Infinity Cache for the next frame use ordinary loads
        if (P.nc == 2) gemm_tile<1, 2, 4, true, SPLIT, true, true>(P, L.B, m_tile, n_tile, s_mem);
        else if (P.nt) gemm_tile<1, 1, 8, true, SPLIT, true, true>(P, L.B, m_tile, n_tile, s_mem);
        else gemm_tile<1, 1, 8, true, SPLIT, true, false>(P, L.B, m_tile, n_tile, s_mem);
    } else {
        if (P.nc == 2) gemm_tile<1, 2, 4, true, SPLIT, true, false>(P, L.B, m_tile, n_tile, s_mem);
        else gemm_tile<1, 1, 8, true, SPLIT, true, false>(P, L.B, m_tile, n_tile, s_mem);
    }
}
__global__ __launch_bounds__(RC_NW * 64, 4) void rc_gemm_small_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_SMALL_LDS_FLOATS];
    small_tiles<false>(L, s_mem);
}
// Live frames (one frame per host round trip), every weight slice read by ONE workgroup: non-temporal weight loads -- measured
// 127 -> 122 us on the live frame's p50. Not for throughput runs: back-to-back frames at batch 1-16 lose 6-11 % with them (what
// the ordinary loads leave in the Infinity Cache serves the next frame), two row tiles per slice (batch 32) 8 %.
__global__ __launch_bounds__(RC_NW * 64, 4) void rc_gemm_small_nt_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_SMALL_LDS_FLOATS];
    small_tiles<false, true>(L, s_mem);
}
__global__ __launch_bounds__(RC_NW * 64, 4) void rc_gemm_small_split_kernel(const GemmLaunch L) {
    __shared__ __attribute__((aligned(16))) float s_mem[RC_SMALL_LDS_FLOATS];
    small_tiles<true>(L, s_mem);
}

bool rc_gemm_is_small(const GemmLaunch& L) {
    bool small = true;
    for (int q = 0; q < L.n; ++q) small = small && L.p[q].mr == 1 && L.p[q].nc <= 2;
    return small;
}

bool rc_gemm_is_mid(const GemmLaunch& L) {
    bool mid = true;
    for (int q = 0; q < L.n; ++q) mid = mid && L.p[q].mr <= 2 && L.p[q].nc <= 4 && !(L.p[q].mr == 2 && L.p[q].nc < 4);
    return mid;
}

// stop: optional event signalled by THIS dispatch's completion (hipExtLaunchKernelGGL: no separate marker packet in the queue)
// true: a wide-tile split-product launch that rc_launch_gemm puts on rc_gemm_split48_w32_kernel (every problem 64 x 128 tiles, one row tile)
bool rc_gemm_is_w32(const GemmLaunch& L) {
    if (rc_gemm_is_small(L) || rc_gemm_is_mid(L)) return false;
    bool one_reader = L.split != 0;
    for (int q = 0; q < L.n; ++q) one_reader = one_reader && L.p[q].mr == 4 && L.p[q].nc == 8 && L.p[q].m_tiles == 1;
    static const bool w32 = !std::getenv("RC_GEMM_W32") || std::atoi(std::getenv("RC_GEMM_W32")) != 0;
    return one_reader && w32;
}

void rc_launch_gemm(const GemmLaunch& L, int total_wg, hipStream_t s, hipEvent_t stop) {
    const dim3 g(total_wg), b(RC_NW * 64);
#define RC_GO(K) do { if (stop) hipExtLaunchKernelGGL(K, g, b, 0, s, nullptr, stop, 0, L); else hipLaunchKernelGGL(K, g, b, 0, s, L); } while (0)
    if (rc_gemm_is_small(L)) {
        bool single_reader = L.live != 0;      // a live frame's launch (rc_api.cpp: launch_problems)
        for (int q = 0; q < L.n; ++q) single_reader = single_reader && L.p[q].m_tiles == 1;
        if (L.split) RC_GO(rc_gemm_small_split_kernel);
        else if (single_reader) RC_GO(rc_gemm_small_nt_kernel);
        else RC_GO(rc_gemm_small_kernel);
    } else if (rc_gemm_is_mid(L)) {
        if (L.split) RC_GO(rc_gemm_mid_split_kernel);
        else RC_GO(rc_gemm_mid_kernel);
    } else {
        if (rc_gemm_is_w32(L)) RC_GO(rc_gemm_split48_w32_kernel);
        else if (L.split) RC_GO(rc_gemm_split_kernel);
        else RC_GO(rc_gemm_kernel);
    }
#undef RC_GO
}
