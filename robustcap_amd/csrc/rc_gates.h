// Gate non-linearities of the LSTM epilogues (rc_gemm.hip, rc_live.hip): one definition, so that every path computes the same bits.
#pragma once
#ifndef RC_FAST_GATES
#define RC_FAST_GATES 1       // on v_exp_f32 / v_rcp_f32 (0: libm expf / tanhf, A/B builds): measured +3.6 % frame rate with parity
#endif                        // margins unchanged (profiles/r02_parity_margins.json)
#if RC_FAST_GATES
__device__ __forceinline__ float rc_gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x)); }
// tanh(x) = sign(x) (1 - t) / (1 + t), t = exp(-2 |x|) in (0, 1]: no cancellation near 0 (1 - 2 / (1 + e^2x) loses every
// significant bit of a small x there)
__device__ __forceinline__ float rc_gate_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(-2.88539008177793f * __builtin_fabsf(x));
    return __builtin_copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
}
#else
__device__ __forceinline__ float rc_gate_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float rc_gate_tanh(float x) { return tanhf(x); }
#endif

// One LSTM cell update from the four gate pre-activations (torch order i, f, g, o; articulate/utils/torch/rnn.py:129-133 -> aten::lstm):
// c' = sigma(f) c + sigma(i) tanh(g), h' = sigma(o) tanh(c'). ONE definition with the contraction written out (fma(f, c, i * g)), so
// that every tile shape of rc_gemm.hip and the shared-weight kernel of rc_gemm_lds.hip produce the same bits.
__device__ __forceinline__ void rc_lstm_cell(float gi, float gf, float gg, float go, float c_prev, float& c_new, float& h_new) {
    const float ig = rc_gate_sigmoid(gi), fg = rc_gate_sigmoid(gf);
    const float cg = rc_gate_tanh(gg), og = rc_gate_sigmoid(go);
    c_new = __builtin_fmaf(fg, c_prev, ig * cg);
    h_new = og * rc_gate_tanh(c_new);
}
