// Gate GEMM of the sig_mp sub-nets on CDNA4 (gfx950): y = epilogue(A[rows, K] * W^T + b) in exact fp32.
//
// Replaces, per step of one sub-net, the torch ops of f(i, x) (net/sig_mp.py:126-129):
//   linear1 + ReLU, aten::lstm (two layers; CPU path = oneDNN mkldnn_rnn_layer), linear2
// and rnn2.init_net (articulate/utils/torch/rnn.py:195-201).
//
// Shape regime: M = batch rows (256..1024, skinny), K = 2H (1024..2560), N = 4H. The grid is made of
// 32-row x 64-column tiles; a tile's K range is split over the 4 waves of its workgroup so that a 512-unit
// layer already yields 256 workgroups (one per CU). Each wave streams ITS slice of the weights straight from
// L2/HBM into VGPRs as 1 KiB coalesced dwordx4 loads (weights are pre-packed in MFMA-B fragment order, nothing
// is shared between waves so LDS staging would be pure overhead) and feeds v_mfma_f32_32x32x2_f32 -- bitwise an
// fp32 fma chain, which is what keeps the 1e-4 parity budget. Partial sums meet in LDS; the epilogue applies the
// bias and, for LSTM layers, the gate non-linearities and the (c, h) update, so gates never touch HBM.
// The 8 row tiles that share a weight slice are mapped to the same XCD (block id % 8) to share it through L2.
//
// Row selection is stateless: every workgroup derives the compacted list of active rows from the per-row flag
// bytes itself (one ballot per wave), so masked sub-steps (vision branch on/off) cost only the rows they touch.
#include "rc_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDS_LD (RC_NT + 16)   // +16 floats: the epilogue's two rows per half-wave land on disjoint banks

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Frag {                 // one prefetch group: RC_G chunks of 8 k for A and for both column blocks
    f32x4 a[RC_G];
    f32x4 b0[RC_G];
    f32x4 b1[RC_G];
};

#ifndef RC_ABLATE
#define RC_ABLATE 0      // tools/gemm_probe.cpp: 1 = no A loads, 2 = no B loads, 3 = no loads, 4 = no MFMA
#endif

__device__ __forceinline__ void load_group(Frag& f, const float* pa, const float* pb0, const float* pb1) {
#pragma unroll
    for (int c = 0; c < RC_G; ++c) {
        if (!(RC_ABLATE & 1) || RC_ABLATE == 4) f.a[c] = *reinterpret_cast<const f32x4*>(pa + 256 * c);
        if (!(RC_ABLATE & 2) || RC_ABLATE == 4) {
            f.b0[c] = *reinterpret_cast<const f32x4*>(pb0 + 256 * c);
            f.b1[c] = *reinterpret_cast<const f32x4*>(pb1 + 256 * c);
        }
    }
}

__device__ __forceinline__ void mma_group(const Frag& f, f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int c = 0; c < RC_G; ++c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#if RC_ABLATE == 4
            acc0[s] += f.a[c][s] + f.b0[c][s];
            acc1[s] += f.b1[c][s];
#else
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[c][s], f.b0[c][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[c][s], f.b1[c][s], acc1, 0, 0, 0);
#endif
        }
    }
}

__global__ __launch_bounds__(RC_NW * 64) void rc_gemm_kernel(const GemmLaunch L) {
    __shared__ float s_part[RC_NW][RC_MT][LDS_LD];
    __shared__ int s_rows[RC_MT];
    __shared__ int s_cnt[RC_NW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < RC_MAX_PROB; ++q)
        if (q < L.n && (int)blockIdx.x >= L.p[q].wg_base) pi = q;
    const GemmProblem& P = L.p[pi];
    const int local = blockIdx.x - P.wg_base;
    int m_tile, n_tile;
    if ((P.n_tiles & 7) == 0) {   // XCD-aware: the row tiles of one weight slice share block-id % 8
        const int xcd = local & 7, s = local >> 3;
        m_tile = s % P.m_tiles;
        n_tile = (s / P.m_tiles) * 8 + xcd;
    } else {
        m_tile = local % P.m_tiles;
        n_tile = local / P.m_tiles;
    }
    if (n_tile >= P.n_tiles) return;

    // ---- active rows of this tile -------------------------------------------------------------------------
    const int B = L.B, lo = m_tile * RC_MT;
    int nrows;
    if (P.flag_bit == 0) {
        nrows = min(RC_MT, B - lo);
        if (nrows <= 0) return;
        if (tid < RC_MT) s_rows[tid] = lo + min(tid, nrows - 1);
        __syncthreads();
    } else {
        int total = 0;
        for (int base = 0; base < B && total < lo + RC_MT; base += RC_NW * 64) {
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
            if (f && idx >= lo && idx < lo + RC_MT) s_rows[idx - lo] = r;
            total += sum;
            __syncthreads();
        }
        nrows = min(RC_MT, total - lo);
        if (nrows <= 0) return;
        if (tid < RC_MT && tid >= nrows) s_rows[tid] = s_rows[0];
        __syncthreads();
    }
    if (P.open_step && n_tile == 0 && tid < nrows) P.steps[s_rows[tid]] += 1;

    // ---- K loop: wave `wave` owns chunks [wave*Qw, (wave+1)*Qw) -------------------------------------------
    const int i = lane & 31, kh = lane >> 5;
    const int row = s_rows[i];
    const int st = (P.seg[0].par_mode | P.seg[1].par_mode) ? P.steps[row] : 0;
    const float* pa_seg[2];
#pragma unroll
    for (int sgi = 0; sgi < 2; ++sgi) {
        const GemmSeg& sg = P.seg[sgi];
        const int par = sg.par_mode == RC_PAR_SRC ? ((st - 1) & 1) : (sg.par_mode == RC_PAR_DST ? (st & 1) : 0);
        pa_seg[sgi] = sg.base + (long long)par * sg.par_stride + rc_pk(row, 4 * kh, sg.ld);
    }
    const int Q = P.Kp >> 3, Qw = Q / RC_NW, ng = Qw / RC_G;
    const int K0 = P.seg[0].K;
    const float* pb0 = P.W + ((long long)(n_tile * 2) * Q + (long long)wave * Qw) * 256 + lane * 4;
    const float* pb1 = pb0 + (long long)Q * 256;
    const int kbase = wave * Qw * 8;

    f32x16 acc0 = {0}, acc1 = {0};
    auto a_ptr = [&](int g) -> const float* {
        const int k = kbase + g * (8 * RC_G);
        return k < K0 ? pa_seg[0] + k * 32 : pa_seg[1] + (k - K0) * 32;      // chunk k/8 -> 256 floats
    };
    // Software pipeline: the loads of group g+1 are issued BEFORE the 32 MFMAs of group g and stay in flight
    // behind them (12 dwordx4 = 12 KiB per wave; the wait in front of an MFMA block is a counted vmcnt(12)).
    // Two things are needed for hipcc (ROCm 7.2) to keep it that way: no conditional inside the steady-state
    // loop (else it waits vmcnt(0) and round-trips the accumulators through VGPRs), and sched_barrier(0)
    // between the phases (else it hoists both load groups to the loop top and drains them inside the iteration).
#define LOADG(F, G) load_group(F, a_ptr(G), pb0 + (G) * (256 * RC_G), pb1 + (G) * (256 * RC_G))
#ifndef RC_PIPE
#define RC_PIPE 0
#endif
#if RC_PIPE
#define SB() __builtin_amdgcn_sched_barrier(0)
#else
#define SB() ((void)0)
#endif
    Frag fa = {}, fb = {};
    int g = 0;
    LOADG(fa, 0);
    if (ng & 1) {                       // odd group count (only K' = 128): peel one group
        SB();
        mma_group(fa, acc0, acc1);
        g = 1;
        if (ng > 1) LOADG(fa, 1);
    }
    if (g < ng) {
        for (; g + 2 < ng; g += 2) {
            LOADG(fb, g + 1);
            SB();
            mma_group(fa, acc0, acc1);
            SB();
            LOADG(fa, g + 2);
            SB();
            mma_group(fb, acc0, acc1);
            SB();
        }
        LOADG(fb, g + 1);
        SB();
        mma_group(fa, acc0, acc1);
        SB();
        mma_group(fb, acc0, acc1);
    }
#undef LOADG
#undef SB

    // ---- split-K reduction through LDS (C layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * kh;
        s_part[wave][rr][i] = acc0[r];
        s_part[wave][rr][32 + i] = acc1[r];
    }
    __syncthreads();

    if (P.epi == RC_EPI_LSTM) {
        // thread -> (row rr, unit u); columns of a tile are [i(16) | f(16) | g(16) | o(16)]
        const int u = tid & 15;
        const int unit = n_tile * RC_UNITS + u;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rr = (tid >> 4) + 16 * pass;
            float gsum[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int col = gq * 16 + u;
                float v = s_part[0][rr][col];
#pragma unroll
                for (int w = 1; w < RC_NW; ++w) v += s_part[w][rr][col];
                gsum[gq] = v + P.bias[n_tile * RC_NT + col];
            }
            if (rr < nrows) {
                const int r2 = s_rows[rr];
                const int dst = P.steps[r2] & 1;
                const long long ci = (long long)r2 * P.H + unit;
                const float ig = sigmoidf_(gsum[0]), fg = sigmoidf_(gsum[1]);
                const float gg = tanhf(gsum[2]), og = sigmoidf_(gsum[3]);
                const float cn = fg * P.cstate[ci] + ig * gg;
                P.cstate[ci] = cn;
                P.hstate[(long long)dst * P.h_par_stride + rc_pk(r2, unit, P.H)] = og * tanhf(cn);
            }
        }
    } else {
        const int col = tid & 63;
        const int n = n_tile * RC_NT + col;
        const float bv = P.bias[n];
        for (int rr = tid >> 6; rr < nrows; rr += RC_NW) {
            float v = s_part[0][rr][col];
#pragma unroll
            for (int w = 1; w < RC_NW; ++w) v += s_part[w][rr][col];
            v += bv;
            if (P.epi == RC_EPI_RELU) v = fmaxf(v, 0.0f);
            if (n < P.N) {
                const int r2 = s_rows[rr];
                P.out[P.out_packed ? rc_pk(r2, P.out_col0 + n, P.ldo) : (long long)r2 * P.ldo + P.out_col0 + n] = v;
            }
        }
    }
}

void rc_launch_gemm(const GemmLaunch& L, int total_wg, hipStream_t s) {
    hipLaunchKernelGGL(rc_gemm_kernel, dim3(total_wg), dim3(RC_NW * 64), 0, s, L);
}
