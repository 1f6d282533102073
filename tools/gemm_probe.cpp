// Ablation probe of rc_gemm_kernel (not part of the product): times one LSTM-layer launch of rnn4-like shape (B rows, H hidden).
// The ablation switches (-DRC_ABLATE=n: 1 = no A loads, 2 = no B loads, 3 = no loads, 4 = no MFMA; -DRC_ABL_SPLIT=mask for the
// split-bf16 K loop) are NOT in the product source -- timing-only builds that compute wrong results must not be one stray -D away
// from a library that passes the ABI test. They live in profiles/r04_gemm_ablation_macros.diff: apply it to a COPY of rc_gemm.hip
//   cp robustcap_amd/csrc/rc_gemm.hip /tmp/rc_gemm_probe.hip && patch /tmp/rc_gemm_probe.hip profiles/r04_gemm_ablation_macros.diff
// and include that copy here instead. Unpatched, this file times the product kernel (RC_ABLATE reads 0):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I robustcap_amd/csrc tools/gemm_probe.cpp -o tools/probe0
#ifndef RC_ABLATE
#define RC_ABLATE 0
#endif
#include "../robustcap_amd/csrc/rc_gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 1280, iters = 20;
    const int NBv = argc > 3 ? atoi(argv[3]) : (H == 1280 ? 5 : 4);
    const int MRv = argc > 4 ? atoi(argv[4]) : (H >= 1024 ? 4 : 2);
    const long long BH = (long long)B * H;
    float *x1, *h, *c, *W, *bias; int* steps;
    hipMalloc(&x1, BH * 4); hipMalloc(&h, 2 * BH * 4); hipMalloc(&c, BH * 4);
    hipMalloc(&W, (size_t)4 * H * 2 * H * 4); hipMalloc(&bias, 4 * H * 4); hipMalloc(&steps, B * 4);
    std::vector<float> hw((size_t)4 * H * 2 * H);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((i * 2654435761u) % 2001) / 1000.0f * 0.02f - 0.02f;
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hx(BH);
    for (long long i = 0; i < BH; ++i) hx[i] = ((i * 40503u) % 1999) / 1000.0f - 1.0f;
    hipMemcpy(x1, hx.data(), BH * 4, hipMemcpyHostToDevice);
    hipMemcpy(h, hx.data(), BH * 4, hipMemcpyHostToDevice);
    hipMemset(h + BH, 0, BH * 4); hipMemset(c, 0, BH * 4); hipMemset(bias, 0, 4 * H * 4);
    std::vector<int> one(B, 1);
    hipMemcpy(steps, one.data(), B * 4, hipMemcpyHostToDevice);
    GemmLaunch L{};
    L.n = 1; L.B = B;
    GemmProblem& p = L.p[0];
    p.seg[0] = {x1, 0, H, H, RC_PAR_NONE, 0};
    p.seg[1] = {h, BH, H, H, RC_PAR_SRC, 0};
    p.W = W; p.bias = bias; p.hstate = h; p.cstate = c; p.h_par_stride = BH; p.H = H; p.steps = steps;
    p.flag_bit = 0; p.epi = RC_EPI_LSTM; p.n_tiles = H / (4 * NBv); p.nc = NBv; p.mr = MRv; p.m_tiles = (B + 16 * MRv - 1) / (16 * MRv); p.Kp = 2 * H; p.wg_base = 0;
    const int wgs = p.n_tiles * p.m_tiles;
    int occ = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rc_gemm_kernel, RC_NW * 64, 0);
    hipFuncAttributes fa_; hipFuncGetAttributes(&fa_, (const void*)rc_gemm_kernel);
    printf("occupancy API: %d blocks/CU, regs %d, lds %zu\n", occ, fa_.numRegs, fa_.sharedSizeBytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) rc_launch_gemm(L, wgs, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) rc_launch_gemm(L, wgs, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, flop = 2.0 * B * (2.0 * H) * (4.0 * H);
    printf("ablate=%d mr=%d nc=%d B=%d H=%d wgs=%d  %.1f us/launch  %.1f TFLOP/s  weights %.1f MB -> %.2f TB/s\n", RC_ABLATE, MRv, NBv, B, H, wgs, us,
           flop / us * 1e-6, 4.0 * H * 2 * H * 4e-6, 4.0 * H * 2 * H * 4 / us * 1e-6);
    return 0;
}
