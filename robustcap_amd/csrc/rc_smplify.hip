// smplify optimiser on gfx950: loss and analytic gradient of the temporal body-fitting objective.
//
// Reference (autograd on the host or device through the full 6890-vertex mesh):
//   net/smplify/temporal_smplify.py:150-166  closure: batch_rodrigues -> forward_kinematics(calc_mesh) -> sync_mp3d ->
//                                            temporal_body_fitting_loss(...).backward()
//   net/smplify/losses.py:15-87              the objective (reprojection GMoF, GMM pose prior, angle prior, 3D term,
//                                            gradient-dead IMU term, temporal smoothness in 2D and 3D)
//   net/smplify/prior.py:164-179             MaxMixturePrior.merged_log_likelihood
// Here: parameters x = [T, 72 axis-angle + 3 translation]. Three kernels per evaluation:
//   rc_smplify_prior_kernel GMM pose prior of 64 frames per workgroup (lane = frame, wave = mixture): value, argmin, gradient rows
//   rc_smplify_fwd_kernel   one workgroup per frame; primal: rotations, FK, 33 landmarks, projection, per-frame loss terms
//   rc_smplify_grad_kernel  one workgroup per frame, on the forward kernel's primal (rotations, landmarks): the adjoint lambda = dL/d(landmark)
//                           of the frame (incl. the smoothness terms that couple it to frames t-1 and t+1), contracted with the constant
//                           skinning data into per-joint co-factors, pulled from the leaves of the kinematic tree to the root (one pass by
//                           tree level over child lists), and dotted with dRl/d(axis-angle component) per joint -- reverse mode throughout
// and, for a batch of rows (rc_smplify_run_batch), one launch each for set-up (residual of the initial pose, R -> axis-angle, joints and
// preserved landmarks) and wrap-up (axis-angle -> R, residual after). HBM traffic per frame and evaluation: 75 parameters in, 75 gradients
// out, ~4 KB of primal (rotations, landmarks, projections) written by the forward kernel and read by the gradient kernel.
#include "rc_device.h"
#include <algorithm>

#define SM_SIGMA 100.0f
#define SM_NG 8
#define SM_DIM 69


// net/smplify/temporal_smplify.py:25-59: R = I + sin(th) K + (1 - cos(th)) K^2, th = |v + 1e-8|, K = [v / th]x
__device__ __forceinline__ void batch_rodrigues(const float* v, float* R) {
    const float e[3] = {v[0] + 1e-8f, v[1] + 1e-8f, v[2] + 1e-8f};
    const float th = norm3(e);
    const float k[3] = {v[0] / th, v[1] / th, v[2] / th};
    const float s = sinf(th), c1 = 1.0f - cosf(th);
    const float Km[9] = {0.f, -k[2], k[1], k[2], 0.f, -k[0], -k[1], k[0], 0.f};
    float K2[9];
    mat3_mul(Km, Km, K2);
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = ((q % 4 == 0 ? 1.0f : 0.0f) + s * Km[q]) + c1 * K2[q];
}

// derivative of batch_rodrigues with respect to component `a` of v
__device__ __forceinline__ void batch_rodrigues_tangent(const float* v, int a, float* dR) {
    const float e[3] = {v[0] + 1e-8f, v[1] + 1e-8f, v[2] + 1e-8f};
    const float th = norm3(e);
    const float dth = e[a] / th;
    float k[3], dk[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        k[q] = v[q] / th;
        dk[q] = ((q == a) ? 1.0f / th : 0.0f) - v[q] * dth / (th * th);
    }
    const float s = sinf(th), c = cosf(th);
    const float Km[9] = {0.f, -k[2], k[1], k[2], 0.f, -k[0], -k[1], k[0], 0.f};
    const float dK[9] = {0.f, -dk[2], dk[1], dk[2], 0.f, -dk[0], -dk[1], dk[0], 0.f};
    float K2[9], A[9], Bm[9];
    mat3_mul(Km, Km, K2);
    mat3_mul(dK, Km, A);
    mat3_mul(Km, dK, Bm);
#pragma unroll
    for (int q = 0; q < 9; ++q) dR[q] = (c * dth) * Km[q] + s * dK[q] + (s * dth) * K2[q] + (1.0f - c) * (A[q] + Bm[q]);
}

__device__ __forceinline__ float gmof(float e) { return (SM_SIGMA * SM_SIGMA * (e * e)) / (SM_SIGMA * SM_SIGMA + e * e); }
__device__ __forceinline__ float gmof_d(float e) {
    const float s2 = SM_SIGMA * SM_SIGMA, q = s2 + e * e;
    return 2.0f * s2 * s2 * e / (q * q);
}
__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// primal of one frame into LDS: rotations -> FK -> landmarks (s.J33 includes the translation)
__device__ __forceinline__ void frame_primal(const BodyConst* body, WaveScratch& s, const float* aa, const float* tran, int lane) {
    if (lane < 24) {
        float R[9];
        batch_rodrigues(aa + 3 * lane, R);
#pragma unroll
        for (int k = 0; k < 9; ++k) s.Rl[lane][k] = R[k];
    }
    __syncthreads();
    wave_body_fk(body, s, tran, lane);
}

// ================================================================================================ GMM pose prior
// prior.py:164-179 (MaxMixturePrior.merged_log_likelihood) and its gradient for every frame, ahead of the forward kernel.
// Rounds 1-4 formed the eight 69 x 69 quadratic forms inside the per-frame forward workgroup: 152 KB of precision rows through the
// L1 per frame and evaluation (lanes = rows of P, 276 B apart) -- 0.8 of the forward kernel's 830 us on config 3's 43,200 frames.
// Here a LANE is a FRAME and a WAVE a MIXTURE: the frame's 69 differences d = x - mean_m stay in VGPRs, a row of P_m is wave-uniform
// and arrives through the scalar cache (s_load into SGPRs: one v_fma per element, no LDS, no vector memory in the loop). The
// arithmetic is the per-frame kernel's operation for operation -- r_i = sum_j P[i][j] d_j as one fma chain in j, lane sums
// q_l = r_l d_l (+ r_{l+64} d_{l+64}), and wave_sum's xor butterfly as the pairwise tree it is: the leaves taken in bit-reversed
// order, partial sums on a six-deep stack -- so losses, argmin and gradients are bitwise what they were.
// Rows of P and of S = P + P^T are padded to SM_PLD floats (rc_smplify_set_prior) so that every row starts 32-byte aligned.
#define SM_PLD 72
#define RC_CONST4 __attribute__((address_space(4)))

template <int N>
__device__ __forceinline__ float prior_row_dot(const RC_CONST4 float* __restrict__ row, const float (&d)[N]) {
    float r = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) r = __builtin_fmaf(row[j], d[j], r);       // (explicit: the SLP vectoriser would pair the tail into v_pk_mul + adds)
    return r;
}

__device__ __forceinline__ void smplify_prior_block(const SmplifyArgs& A, const int t0) {
    __shared__ float s_x[SM_DIM][65];         // pose components [j][frame]; after the argmin: the selected gradient rows
    __shared__ float s_ll[SM_NG][64];
    __shared__ int s_arg[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);               // mixture of this wave
    const int nf = min(64, A.T - t0);
    for (int e = tid; e < 64 * 72; e += 512) {
        const int f = e / 72, c = e % 72;
        if (c >= 3) s_x[c - 3][f] = (f < nf) ? A.aa[(long long)t0 * 72 + e] : 0.0f;
    }
    __syncthreads();
    const RC_CONST4 float* mean = (const RC_CONST4 float*)(A.means + w * SM_DIM);
    const RC_CONST4 float* Pw = (const RC_CONST4 float*)(A.prec + (long long)w * SM_DIM * SM_PLD);
    const RC_CONST4 float* Sw = (const RC_CONST4 float*)(A.prec_sym + (long long)w * SM_DIM * SM_PLD);
    float d[SM_DIM];
#pragma unroll
    for (int j = 0; j < SM_DIM; ++j) d[j] = s_x[j][lane] - mean[j];
    // ---- 0.5 d^T P d: leaves q_l in bit-reversed order, pairwise sums on a stack = wave_sum's butterfly over l = 0..63
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, total = 0.f;
#pragma unroll 1
    for (int k = 0; k < 64; ++k) {
        const int i = (int)(__builtin_bitreverse32((unsigned)k) >> 26);
        const float r = prior_row_dot(Pw + i * SM_PLD, d);
        float q = __builtin_fmaf(r, s_x[i][lane] - mean[i], 0.0f);
        if (i + 64 < SM_DIM) {
            const float r2 = prior_row_dot(Pw + (i + 64) * SM_PLD, d);
            q = __builtin_fmaf(r2, s_x[i + 64][lane] - mean[i + 64], q);
        }
        float v = q;
        if (k & 1) {
            v = s0 + v;
            if (k & 2) {
                v = s1 + v;
                if (k & 4) {
                    v = s2 + v;
                    if (k & 8) {
                        v = s3 + v;
                        if (k & 16) {
                            v = s4 + v;
                            if (k & 32) total = s5 + v; else s5 = v;
                        } else s4 = v;
                    } else s3 = v;
                } else s2 = v;
            } else s1 = v;
        } else s0 = v;
    }
    s_ll[w][lane] = 0.5f * total - ((const RC_CONST4 float*)A.lognll)[w];
    __syncthreads();
    if (w == 0) {
        float best = s_ll[0][lane];
        int bi = 0;
        for (int m = 1; m < SM_NG; ++m) {
            const float ll = s_ll[m][lane];
            if (ll < best) { best = ll; bi = m; }
        }
        s_arg[lane] = bi;
        if (lane < nf) { A.prior_ll[t0 + lane] = best; A.argmin[t0 + lane] = bi; }
    }
    __syncthreads();
    // ---- gradient rows (P + P^T) d of the frames whose mixture this wave is (a wave none of whose frames chose it has nothing to do)
    const bool mine = (s_arg[lane] == w) && (lane < nf);
    __syncthreads();                                                      // every d is in registers: s_x becomes the output tile
    if (__ballot(mine) != 0ull) {
#pragma unroll 1
        for (int i = 0; i < SM_DIM; ++i) {
            const float r = prior_row_dot(Sw + i * SM_PLD, d);
            if (mine) s_x[i][lane] = r;
        }
    }
    __syncthreads();
    for (int e = tid; e < nf * SM_DIM; e += 512) A.prior_g[(long long)t0 * SM_DIM + e] = s_x[e % SM_DIM][e / SM_DIM];
}

__global__ __launch_bounds__(512) void rc_smplify_prior_kernel(SmplifyArgs A) { smplify_prior_block(A, (int)blockIdx.x * 64); }
__global__ __launch_bounds__(512) void rc_smplify_prior_rows_kernel(const SmplifyArgs* __restrict__ rows) {
    const SmplifyArgs A = rows[blockIdx.y];
    if ((int)blockIdx.x * 64 >= A.T) return;
    smplify_prior_block(A, (int)blockIdx.x * 64);
}

// ============================================================================================== forward pass
__device__ __forceinline__ void smplify_fwd_frame(const SmplifyArgs& A, const BodyConst* __restrict__ body_g, const int t) {
    __shared__ WaveScratch s;
    const int lane = threadIdx.x;
    const BodyConst* body = body_g;     // (read through the L1: staging the 4.9 KB in LDS per one-wave workgroup cost more than it saved, 207 vs 200 us)
    const float* aa = A.aa + (long long)t * 72;
    const float tr[3] = {A.tran[t * 3], A.tran[t * 3 + 1], A.tran[t * 3 + 2]};
    frame_primal(body, s, aa, tr, lane);
    {   // the rotations of the primal for the gradient kernel (it used to redo Rodrigues and the FK sweep)
        float* fk = A.fk + (long long)t * 432;
        const float* rl = &s.Rl[0][0];
        const float* gg = &s.G[0][0];
        for (int e = lane; e < 216; e += 64) { fk[e] = rl[e]; fk[216 + e] = gg[e]; }
    }

    // landmarks, projection, reprojection + 3D terms
    float part = 0.0f;
    if (lane < 33) {
        const float x = s.J33[lane][0], y = s.J33[lane][1], z = s.J33[lane][2];
        const float xn = x / z, yn = y / z, on = z / z;
        const float u = (A.K[0] * xn + A.K[1] * yn) + A.K[2] * on, v = (A.K[3] * xn + A.K[4] * yn) + A.K[5] * on;
        float* mj = A.mj + ((long long)t * 33 + lane) * 3;
        mj[0] = x; mj[1] = y; mj[2] = z;
        A.proj[((long long)t * 33 + lane) * 2] = u;
        A.proj[((long long)t * 33 + lane) * 2 + 1] = v;
        const float* k3 = A.kp + ((long long)t * 33 + lane) * 3;
        const float cf = ((A.ign_mask >> lane) & 1ull) ? 0.0f : k3[2];
        part = (cf * cf) * (gmof(u - k3[0]) + gmof(v - k3[1]));                          // losses.py:43-46
        if (lane >= 1) {                                                               // losses.py:31-33
            const float* r = A.ref3d + (long long)t * 99;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = (s.J33[lane][c] - s.J33[0][c]) - (r[3 * lane + c] - r[c]);
                part += d * d;
            }
        }
    }
    float loss = wave_sum(part);
    // angle prior (losses.py:15-21, 55): exp(+-x)^2 on pose_axis[52, 55, 9, 12]
    if (lane < 4) {
        const int idx[4] = {55, 58, 12, 15};
        const float sg[4] = {1.f, -1.f, -1.f, -1.f};
        const float e = expf(aa[idx[lane]] * sg[lane]);
        part = (15.2f * 15.2f) * (e * e);
    } else part = 0.0f;
    loss += wave_sum(part);
    // GMM prior (prior.py:164-179): min_m 0.5 d^T P_m d - log(nll_w_m), formed for all frames by rc_smplify_prior_kernel
    loss += 0.01f * A.prior_ll[t];                                                      // pose_prior_weight ** 2
    // gradient-dead IMU term (losses.py:39-40): value only
    part = 0.0f;
    if (lane < 6) {
        const int ji[6] = {18, 19, 4, 5, 15, 0};
        float a3[3];
        rotmat_to_aa(s.G[ji[lane]], a3);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = A.imu_aa[t * 18 + 3 * lane + c] - a3[c]; part += d * d; }
    }
    const float imu = 0.25f * wave_sum(part);
    if (lane == 0) { A.frame_loss[t] = loss; A.imu_loss[t] = imu; }
}

__global__ __launch_bounds__(64) void rc_smplify_fwd_kernel(SmplifyArgs A, const BodyConst* __restrict__ body_g) {
    smplify_fwd_frame(A, body_g, (int)blockIdx.x);
}
// all rows of a lock-step round of the batched optimiser (rc_smplify_api.cpp: RowBatch) in one launch: grid (T_max, rows)
__global__ __launch_bounds__(64) void rc_smplify_fwd_rows_kernel(const SmplifyArgs* __restrict__ rows, const BodyConst* __restrict__ body_g) {
    const SmplifyArgs A = rows[blockIdx.y];
    if ((int)blockIdx.x >= A.T) return;
    smplify_fwd_frame(A, body_g, (int)blockIdx.x);
}

// ============================================================================================= gradient pass
// Round 4: the chain rule through the kinematic tree runs in REVERSE (adjoints pulled from the leaves to the root, one pass by tree
// level) instead of pushing 72 tangents forward through the descendants of their joints. Round 3 kept those tangents in 92 KB of
// LDS -- one 128-thread workgroup per CU, 2 waves on a 256-CU chip per 600-frame evaluation: 72 rows x 600 frames x 26 evaluations
// took ~200 ms however they were batched. Now 2 KB of adjoints; the kernel is no longer limited to one workgroup per CU.
__device__ __forceinline__ void smplify_grad_frame(const SmplifyArgs& A, const BodyConst* __restrict__ body_g, const int t) {
    __shared__ __attribute__((aligned(16))) BodyConst s_body;
    __shared__ float s_Rl[24][9], s_G[24][9], s_J33[33][3];   // the forward kernel's primal of this frame
    __shared__ float s_lam[33][3], s_e3[33][3];
    __shared__ float s_Gb[24][9], s_pb[24][3];        // adjoints of the loss w.r.t. a joint's global rotation / position
    const int tid = threadIdx.x, T = A.T;
    stage_body(&s_body, body_g, tid, 128);
    {
        const float* fk = A.fk + (long long)t * 432;
        float* rl = &s_Rl[0][0];
        float* gg = &s_G[0][0];
        for (int e = tid; e < 216; e += 128) { rl[e] = fk[e]; gg[e] = fk[216 + e]; }
        if (tid < 99) (&s_J33[0][0])[tid] = A.mj[(long long)t * 99 + tid];
    }
    __syncthreads();
    const BodyConst* body = &s_body;
    const float* aa = A.aa + (long long)t * 72;

    // ---- adjoint of the loss with respect to the 33 landmarks of THIS frame ---------------------------------
    float sm = 0.0f;
    if (tid < 33) {
        const int v = tid;
        const float x = s_J33[v][0], y = s_J33[v][1], z = s_J33[v][2];
        const float* k3 = A.kp + ((long long)t * 33 + v) * 3;
        const float cf = ((A.ign_mask >> v) & 1ull) ? 0.0f : k3[2], c2 = cf * cf;
        const float u = A.proj[((long long)t * 33 + v) * 2], w = A.proj[((long long)t * 33 + v) * 2 + 1];
        float du = c2 * gmof_d(u - k3[0]), dv = c2 * gmof_d(w - k3[1]);                 // reprojection
        float lam[3] = {0.f, 0.f, 0.f};
        if (t > 0) {                                                                   // smoothness pair (t-1, t)
            const float* pp = A.proj + ((long long)(t - 1) * 33 + v) * 2;
            const float* pm = A.mj + ((long long)(t - 1) * 33 + v) * 3;
            du += 1e-4f * c2 * sgn(u - pp[0]);
            dv += 1e-4f * c2 * sgn(w - pp[1]);
            sm += 1e-4f * c2 * (fabsf(u - pp[0]) + fabsf(w - pp[1]));
            const float d3[3] = {x - pm[0], y - pm[1], z - pm[2]};
#pragma unroll
            for (int c = 0; c < 3; ++c) { lam[c] += c2 * sgn(d3[c]); sm += c2 * fabsf(d3[c]); }
        }
        if (t + 1 < T) {                                                               // pair (t, t+1): conf of t+1
            const float* kn = A.kp + ((long long)(t + 1) * 33 + v) * 3;
            const float cn = ((A.ign_mask >> v) & 1ull) ? 0.0f : kn[2], cn2 = cn * cn;
            const float* pn = A.proj + ((long long)(t + 1) * 33 + v) * 2;
            const float* mn = A.mj + ((long long)(t + 1) * 33 + v) * 3;
            du -= 1e-4f * cn2 * sgn(pn[0] - u);
            dv -= 1e-4f * cn2 * sgn(pn[1] - w);
            lam[0] -= cn2 * sgn(mn[0] - x); lam[1] -= cn2 * sgn(mn[1] - y); lam[2] -= cn2 * sgn(mn[2] - z);
        }
        const float dxn = du * A.K[0] + dv * A.K[3], dyn = du * A.K[1] + dv * A.K[4];
        lam[0] += dxn / z;
        lam[1] += dyn / z;
        lam[2] += -(dxn * x + dyn * y) / (z * z);
        float e3[3] = {0.f, 0.f, 0.f};
        if (v >= 1) {                                                                  // 3D term, own part
            const float* r = A.ref3d + (long long)t * 99;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                e3[c] = (s_J33[v][c] - s_J33[0][c]) - (r[3 * v + c] - r[c]);
                lam[c] += 2.0f * e3[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_lam[v][c] = lam[c]; s_e3[v][c] = e3[c]; }
    }
    if (tid < 64) {
        const float tot = wave_sum(sm);
        if (tid == 0) A.smooth_loss[t] = tot;
    }
    __syncthreads();
    if (tid < 3) {                                                                     // 3D term, landmark 0 part
        float acc = 0.0f;
        for (int v = 1; v < 33; ++v) acc += 2.0f * s_e3[v][tid];
        s_lam[0][tid] -= acc;
    }
    __syncthreads();
    // ---- contract lambda with the constant skinning data into per-joint co-factors: thread (joint i, row a) --
    if (tid < 72) {
        const int i = tid / 3, a = tid % 3;
        float Ma[3] = {0.f, 0.f, 0.f}, m = 0.0f, o = 0.0f;
        for (int v = 0; v < 33; ++v) {
            const int oj = body->override_joint[v];
            const float la = s_lam[v][a];
            if (oj >= 0) {
                if (oj == i) o += la;
                continue;
            }
            const float w = body->w33[v][i];
            m += w * la;
#pragma unroll
            for (int b = 0; b < 3; ++b) Ma[b] += w * la * body->v33[v][b];
        }
        // L depends on joint i through <M, G_i> + m . (P_i - G_i jrest_i) + o . P_i  (model.py:235: T_i = P_i - G_i jrest_i):
        // own adjoints  Gb_i = M - m jrest_i^T,  pb_i = m + o
        const float* jr = body->jrest[i];
#pragma unroll
        for (int b = 0; b < 3; ++b) s_Gb[i][3 * a + b] = Ma[b] - m * jr[b];
        s_pb[i][a] = m + o;
    }
    __syncthreads();
    // ---- reverse sweep: G_c = G_i Rl_c, P_c = P_i + G_i bone_c  =>  Gb_i += Gb_c Rl_c^T + pb_c bone_c^T,  pb_i += pb_c for every
    // child c of i. A joint PULLS from its children (final once their level is done): no two lanes write the same adjoint.
    {
        const int lvl = tid < 24 ? body->level[tid] : -1;
        const int nc = tid < 24 ? body->nchild[tid] : 0;
        for (int l = 8; l >= 0; --l) {
            if (lvl == l && nc > 0) {
                const int i = tid;
                float Gb[9], pb[3];
#pragma unroll
                for (int q = 0; q < 9; ++q) Gb[q] = s_Gb[i][q];
#pragma unroll
                for (int q = 0; q < 3; ++q) pb[q] = s_pb[i][q];
                for (int ci = 0; ci < nc; ++ci) {
                    const int c = body->child[i][ci];
                    const float* Gc = s_Gb[c];
                    const float* Rc = s_Rl[c];
                    const float* pc = s_pb[c];
                    const float* bc = body->bone[c];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b)     // (Gb_c Rl_c^T)[a][b] = sum_k Gb_c[a][k] Rl_c[b][k]
                            Gb[3 * a + b] += ((Gc[3 * a] * Rc[3 * b] + Gc[3 * a + 1] * Rc[3 * b + 1]) + Gc[3 * a + 2] * Rc[3 * b + 2]) + pc[a] * bc[b];
#pragma unroll
                    for (int q = 0; q < 3; ++q) pb[q] += pc[q];
                }
#pragma unroll
                for (int q = 0; q < 9; ++q) s_Gb[i][q] = Gb[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) s_pb[i][q] = pb[q];
            }
            __syncthreads();
        }
    }

    float* gout = A.grad_aa + (long long)t * 72;
    if (tid < 72) {
        // ---- d/d aa[joint j][axis a] = <G_parent^T Gb_j, d Rl_j / d aa_a>  (G_j = G_parent Rl_j; the root's parent is the identity)
        const int j = tid / 3, a = tid % 3, k = tid;
        float dRl[9], Rb[9];
        batch_rodrigues_tangent(aa + 3 * j, a, dRl);
        if (j == 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) Rb[q] = s_Gb[0][q];
        } else mat3T_mul(s_G[body->parent[j]], s_Gb[j], Rb);
        float g = 0.0f;
#pragma unroll
        for (int q = 0; q < 9; ++q) g += Rb[q] * dRl[q];
        // priors act on the 69 non-root components
        if (k >= 3) {
            // d/dx of x^T P x = (P + P^T) x for the frame's mixture: row k - 3 of it from rc_smplify_prior_kernel
            const float r = A.prior_g[(long long)t * SM_DIM + (k - 3)];
            g += 0.01f * 0.5f * r;
            const float sg = (k == 55) ? 1.0f : ((k == 58 || k == 12 || k == 15) ? -1.0f : 0.0f);
            if (sg != 0.0f) { const float e = expf(aa[k] * sg); g += (15.2f * 15.2f) * 2.0f * (e * e) * sg; }
        }
        gout[k] = g;
    } else if (tid < 75) {
        const int c = tid - 72;                                       // d landmark / d tran = I
        float g = 0.0f;
        for (int v = 0; v < 33; ++v) g += s_lam[v][c];
        A.grad_tran[(long long)t * 3 + c] = g;
    }
}

__global__ __launch_bounds__(128) void rc_smplify_grad_kernel(SmplifyArgs A, const BodyConst* __restrict__ body_g) {
    smplify_grad_frame(A, body_g, (int)blockIdx.x);
}
__global__ __launch_bounds__(128) void rc_smplify_grad_rows_kernel(const SmplifyArgs* __restrict__ rows, const BodyConst* __restrict__ body_g) {
    const SmplifyArgs A = rows[blockIdx.y];
    if ((int)blockIdx.x >= A.T) return;
    smplify_grad_frame(A, body_g, (int)blockIdx.x);
}

void rc_launch_smplify(const SmplifyArgs& A, const BodyConst* body, hipStream_t st) {
    if (A.T <= 0) return;
    hipLaunchKernelGGL(rc_smplify_prior_kernel, dim3((A.T + 63) / 64), dim3(512), 0, st, A);
    hipLaunchKernelGGL(rc_smplify_fwd_kernel, dim3(A.T), dim3(64), 0, st, A, body);
    hipLaunchKernelGGL(rc_smplify_grad_kernel, dim3(A.T), dim3(128), 0, st, A, body);
}
// ---- total loss of every row of a round, on the device (round 5). The host used to read the per-frame terms of ALL rows back every round
// (518 KB for 72 rows x 600 frames) and add them in double: 0.2 of a round's 0.77 ms. One wave per row reads its 3 T terms 64 at a time
// and adds them in the HOST's order -- frame by frame, three running sums, the lane's value broadcast to the whole wave so that every lane
// carries the same chain; no fused multiply-add in the last line -- so the sum is bit for bit total_loss() of rc_smplify_api.cpp
// (losses.py:57-87: the IMU term counts T times), and a row's line search takes the decisions it takes alone.
__global__ __launch_bounds__(64) void rc_smplify_total_rows_kernel(const SmplifyArgs* __restrict__ rows, double* __restrict__ out) {
    const SmplifyArgs* A = rows + blockIdx.x;
    const int T = A->T, lane = threadIdx.x;
    const float* fl = A->frame_loss; const float* il = A->imu_loss; const float* sl = A->smooth_loss;
    double f = 0.0, imu = 0.0, sm = 0.0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const float vf = t < T ? fl[t] : 0.f, vi = t < T ? il[t] : 0.f, vs = t < T ? sl[t] : 0.f;
        const int n = min(64, T - t0);
        for (int k = 0; k < n; ++k) {
            f = __dadd_rn(f, (double)__shfl(vf, k));
            imu = __dadd_rn(imu, (double)__shfl(vi, k));
            sm = __dadd_rn(sm, (double)__shfl(vs, k));
        }
    }
    if (lane == 0) out[blockIdx.x] = __dadd_rn(__dadd_rn(f, __dmul_rn((double)T, imu)), sm);
}
void rc_launch_smplify_totals(const SmplifyArgs* rows_dev, int n_rows, double* out_dev, hipStream_t st) {
    if (n_rows <= 0) return;
    hipLaunchKernelGGL(rc_smplify_total_rows_kernel, dim3((unsigned)n_rows), dim3(64), 0, st, rows_dev, out_dev);
}

// The row (or job) of these launches comes from blockIdx.y, and a grid's y extent ends at 65,535: longer tables go out in slices of the
// table (round-4 advice: a lock-step round of the batched optimiser can ask for 6 (max_iter + 1) inner products per row -- 606 jobs a
// row at max_iter = 100 -- and failed as a whole from ~108 rows on).
#define RC_GRID_Y_MAX 65535
void rc_launch_smplify_rows(const SmplifyArgs* rows_dev, int n_rows, int T_max, const BodyConst* body, hipStream_t st) {
    if (n_rows <= 0 || T_max <= 0) return;
    for (int y0 = 0; y0 < n_rows; y0 += RC_GRID_Y_MAX) {
        const unsigned ny = (unsigned)std::min(RC_GRID_Y_MAX, n_rows - y0);
        hipLaunchKernelGGL(rc_smplify_prior_rows_kernel, dim3((unsigned)(T_max + 63) / 64, ny), dim3(512), 0, st, rows_dev + y0);
        hipLaunchKernelGGL(rc_smplify_fwd_rows_kernel, dim3((unsigned)T_max, ny), dim3(64), 0, st, rows_dev + y0, body);
        hipLaunchKernelGGL(rc_smplify_grad_rows_kernel, dim3((unsigned)T_max, ny), dim3(128), 0, st, rows_dev + y0, body);
    }
}

// ======================================================= set-up / wrap-up of a batch of rows, one launch each (grid: frame x row)
// The arithmetic of rc_residual_kernel, rc_R2aa_kernel, rc_body_fk_kernel and rc_aa2R_kernel (rc_frame.hip), which rc_smplify_run
// launches one after another per row: 72 rows x 8 launches and copies were 6.7 of config 3's 32 ms.
__device__ __forceinline__ void residual_of_frame(const WaveScratch& s, const float* K, const float* kp, long long b, float sigma,
                                                  unsigned long long ign_mask, float* loss, int lane) {
    if (lane < 33) {
        const float z = s.J33[lane][2];
        const float q[3] = {s.J33[lane][0] / z, s.J33[lane][1] / z, z / z};
        const float u = (K[0] * q[0] + K[1] * q[1]) + K[2] * q[2];
        const float v = (K[3] * q[0] + K[4] * q[1]) + K[5] * q[2];
        const float* k3 = kp + (b * 33 + lane) * 3;
        const bool ign = (ign_mask >> lane) & 1ull;
        const float cf = ign ? 0.0f : k3[2];
        const float s2 = sigma * sigma;
        const float dx = u - k3[0], dy = v - k3[1];
        const float ex = (s2 * (dx * dx)) / (s2 + dx * dx), ey = (s2 * (dy * dy)) / (s2 + dy * dy);
        loss[b * 33 + lane] = (cf * cf) * (ex + ey);
    }
}

// residual of the initial pose (the pre-check reads its first frame), optimiser parameters (R -> axis-angle, translation), IMU
// orientations as axis-angle, joints and the preserved 3D landmarks of the initial pose (temporal_smplify.py:111-139)
__global__ __launch_bounds__(64) void rc_smplify_begin_rows_kernel(const SmplifyRowIO* __restrict__ rows, const BodyConst* __restrict__ body,
                                                                  float sigma, unsigned long long ign_mask) {
    __shared__ WaveScratch s;
    const SmplifyRowIO& io = rows[blockIdx.y];
    const long long b = blockIdx.x;
    const int T = io.T;
    if (b >= T) return;
    const int lane = threadIdx.x;
    for (int e = lane; e < 216; e += 64) s.Rl[e / 9][e % 9] = io.pose[b * 216 + e];
    const float t[3] = {io.tran[b * 3], io.tran[b * 3 + 1], io.tran[b * 3 + 2]};
    __syncthreads();
    if (lane < 24) rotmat_to_aa(s.Rl[lane], io.x + b * 72 + 3 * lane);
    else if (lane < 30) rotmat_to_aa(io.imu_ori + b * 54 + 9 * (lane - 24), io.imu_aa + b * 18 + 3 * (lane - 24));
    else if (lane < 33) io.x[(long long)T * 72 + b * 3 + (lane - 30)] = t[lane - 30];
    wave_body_fk(body, s, t, lane);
    residual_of_frame(s, io.K, io.kp, b, sigma, ign_mask, io.res0, lane);
    if (lane < 24) {
#pragma unroll
        for (int c = 0; c < 3; ++c) io.joint[(b * 24 + lane) * 3 + c] = s.P[lane][c] + t[c];
    }
    if (lane < 33) {
#pragma unroll
        for (int c = 0; c < 3; ++c) io.ref3d[(b * 33 + lane) * 3 + c] = s.J33[lane][c];
    }
}

// rotations of the refined parameters (art.math.axis_angle_to_rotation_matrix), translation, residual after (run.py:31-34)
__global__ __launch_bounds__(64) void rc_smplify_end_rows_kernel(const SmplifyRowIO* __restrict__ rows, const BodyConst* __restrict__ body,
                                                                float sigma, unsigned long long ign_mask) {
    __shared__ WaveScratch s;
    const SmplifyRowIO& io = rows[blockIdx.y];
    const long long b = blockIdx.x;
    const int T = io.T;
    if (b >= T || !io.live) return;
    const int lane = threadIdx.x;
    if (lane < 24) {
        const float* aa = io.x + b * 72 + 3 * lane;
        const float a[3] = {aa[0], aa[1], aa[2]};
        const float th = norm3(a);
        float k[3] = {a[0] / th, a[1] / th, a[2] / th};
#pragma unroll
        for (int q = 0; q < 3; ++q) if (!(fabsf(k[q]) <= 3.0e38f)) k[q] = 0.0f;          // NaN / inf -> 0
        const float c = cosf(th), sn = sinf(th), tt = 1.0f - c;
        const float Km[9] = {0.f, -k[2], k[1], k[2], 0.f, -k[0], -k[1], k[0], 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float v = ((r == q ? c : 0.0f) + tt * (k[r] * k[q])) + sn * Km[3 * r + q];
                s.Rl[lane][3 * r + q] = v;
                io.pose_out[b * 216 + lane * 9 + 3 * r + q] = v;
            }
    }
    const float* xt = io.x + (long long)T * 72 + b * 3;
    const float t[3] = {xt[0], xt[1], xt[2]};
    if (lane < 3) io.tran_out[b * 3 + lane] = t[lane];
    __syncthreads();
    wave_body_fk(body, s, t, lane);
    residual_of_frame(s, io.K, io.kp, b, sigma, ign_mask, io.res1, lane);
}

void rc_launch_smplify_begin_rows(const SmplifyRowIO* rows_dev, int n_rows, int T_max, const BodyConst* body, float sigma,
                                  unsigned long long ign_mask, hipStream_t st) {
    if (n_rows <= 0 || T_max <= 0) return;
    for (int y0 = 0; y0 < n_rows; y0 += RC_GRID_Y_MAX)
        hipLaunchKernelGGL(rc_smplify_begin_rows_kernel, dim3((unsigned)T_max, (unsigned)std::min(RC_GRID_Y_MAX, n_rows - y0)), dim3(64), 0, st,
                           rows_dev + y0, body, sigma, ign_mask);
}
void rc_launch_smplify_end_rows(const SmplifyRowIO* rows_dev, int n_rows, int T_max, const BodyConst* body, float sigma,
                                unsigned long long ign_mask, hipStream_t st) {
    if (n_rows <= 0 || T_max <= 0) return;
    for (int y0 = 0; y0 < n_rows; y0 += RC_GRID_Y_MAX)
        hipLaunchKernelGGL(rc_smplify_end_rows_kernel, dim3((unsigned)T_max, (unsigned)std::min(RC_GRID_Y_MAX, n_rows - y0)), dim3(64), 0, st,
                           rows_dev + y0, body, sigma, ign_mask);
}

// ================================================================= vector kernels of the device-resident L-BFGS
// (rc_smplify_api.cpp: minimize_on_device). The optimiser's vectors -- parameters, gradients of the bracket points, the
// curvature pairs -- never leave the device; what the host reads back per step is a handful of inner products.
__global__ void rc_vec_axpy_kernel(const float* x, const float* __restrict__ d, float t, float* out, long long n) {   // out may be x
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] + t * d[i];
}
// curvature pair of an iteration: y = g_new - g_old, s = t d  (torch.optim.LBFGS: y = flat_grad.sub(prev_flat_grad), s = d.mul(t))
__global__ void rc_vec_pair_kernel(const float* __restrict__ g_new, const float* __restrict__ g_old, const float* __restrict__ d, float t,
                                   float* __restrict__ y, float* __restrict__ s, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y[i] = g_new[i] - g_old[i]; s[i] = d[i] * t; }
}
// out = sum_k c[k] v[k]  (the search direction as a combination of the curvature pairs and the gradient)
__global__ void rc_vec_comb_kernel(VecComb c, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
    for (int k = 0; k < c.n_vec; ++k) acc += c.c[k] * c.v[k][i];
    out[i] = acc;
}
// partial[job][block] of job (a, b, op): op 0 sum a_i b_i, 1 max |a_i|, 2 sum |a_i| -- float64 accumulation, one partial per
// 4,096-element block; the host adds the partials of a job in block order (deterministic).
__global__ __launch_bounds__(256) void rc_vec_dots_kernel(const VecJob* __restrict__ jobs, long long n, int n_blocks, double* __restrict__ partial) {
    __shared__ double s_red[4];
    const VecJob jb = jobs[blockIdx.y];
    const long long lo = (long long)blockIdx.x * 4096;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int q = 0; q < 16; ++q) {
        const long long i = lo + q * 256 + threadIdx.x;
        if (i < n) {
            const double a = jb.a[i];
            if (jb.op == 0) acc += a * (double)jb.b[i];
            else if (jb.op == 1) acc = fmax(acc, fabs(a));
            else acc += fabs(a);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(acc, off);
        acc = jb.op == 1 ? fmax(acc, o) : acc + o;
    }
    if (lane == 0) s_red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = s_red[0];
        for (int w = 1; w < 4; ++w) r = jb.op == 1 ? fmax(r, s_red[w]) : r + s_red[w];
        partial[(long long)blockIdx.y * n_blocks + blockIdx.x] = r;
    }
}

// ---- the same operations for ALL rows of a lock-step round: one launch each, the row from blockIdx.y (descriptor tables in device memory)
__global__ void rc_vec_ops_kernel(const VecOp* __restrict__ ops) {
    const VecOp o = ops[blockIdx.y];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= o.n) return;
    if (o.kind == 0) o.out[i] = o.a[i] + o.t * o.b[i];                   // axpy (out may be a)
    else { o.out[i] = o.a[i] - o.b[i]; o.out2[i] = o.c[i] * o.t; }       // curvature pair: y = g_new - g_old, s = t d
}
__global__ void rc_vec_comb_rows_kernel(const VecCombRow* __restrict__ rows) {
    const VecCombRow* r = rows + blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r->n) return;
    const int nv = r->c.n_vec;
    float acc = 0.0f;
    for (int k = 0; k < nv; ++k) acc += r->c.c[k] * r->c.v[k][i];
    r->out[i] = acc;
}
__global__ __launch_bounds__(256) void rc_vec_dots_rows_kernel(const VecJobN* __restrict__ jobs, int nb_max, double* __restrict__ partial) {
    __shared__ double s_red[4];
    const VecJobN jb = jobs[blockIdx.y];
    const long long lo = (long long)blockIdx.x * 4096;
    if (lo >= jb.n) return;                                              // (the host adds the first ceil(n / 4096) partials of a job)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int q = 0; q < 16; ++q) {
        const long long i = lo + q * 256 + threadIdx.x;
        if (i < jb.n) {
            const double a = jb.a[i];
            if (jb.op == 0) acc += a * (double)jb.b[i];
            else if (jb.op == 1) acc = fmax(acc, fabs(a));
            else acc += fabs(a);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(acc, off);
        acc = jb.op == 1 ? fmax(acc, o) : acc + o;
    }
    if (lane == 0) s_red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = s_red[0];
        for (int w = 1; w < 4; ++w) r = jb.op == 1 ? fmax(r, s_red[w]) : r + s_red[w];
        partial[(long long)blockIdx.y * nb_max + blockIdx.x] = r;
    }
}
void rc_launch_vec_ops(const VecOp* ops_dev, int n_ops, long long n_max, hipStream_t st) {
    if (n_ops <= 0 || n_max <= 0) return;
    for (int y0 = 0; y0 < n_ops; y0 += RC_GRID_Y_MAX)
        hipLaunchKernelGGL(rc_vec_ops_kernel, dim3((unsigned)((n_max + 255) / 256), (unsigned)std::min(RC_GRID_Y_MAX, n_ops - y0)), dim3(256), 0, st, ops_dev + y0);
}
void rc_launch_vec_comb_rows(const VecCombRow* rows_dev, int n_rows, long long n_max, hipStream_t st) {
    if (n_rows <= 0 || n_max <= 0) return;
    for (int y0 = 0; y0 < n_rows; y0 += RC_GRID_Y_MAX)
        hipLaunchKernelGGL(rc_vec_comb_rows_kernel, dim3((unsigned)((n_max + 255) / 256), (unsigned)std::min(RC_GRID_Y_MAX, n_rows - y0)), dim3(256), 0, st,
                           rows_dev + y0);
}
void rc_launch_vec_dots_rows(const VecJobN* jobs_dev, int n_jobs, int nb_max, double* partial, hipStream_t st) {
    if (n_jobs <= 0 || nb_max <= 0) return;
    for (int y0 = 0; y0 < n_jobs; y0 += RC_GRID_Y_MAX)
        hipLaunchKernelGGL(rc_vec_dots_rows_kernel, dim3((unsigned)nb_max, (unsigned)std::min(RC_GRID_Y_MAX, n_jobs - y0)), dim3(256), 0, st, jobs_dev + y0,
                           nb_max, partial + (long long)y0 * nb_max);
}

void rc_launch_vec_axpy(const float* x, const float* d, float t, float* out, long long n, hipStream_t st) {
    hipLaunchKernelGGL(rc_vec_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, d, t, out, n);
}
void rc_launch_vec_pair(const float* g_new, const float* g_old, const float* d, float t, float* y, float* s, long long n, hipStream_t st) {
    hipLaunchKernelGGL(rc_vec_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g_new, g_old, d, t, y, s, n);
}
void rc_launch_vec_comb(const VecComb& c, float* out, long long n, hipStream_t st) {
    hipLaunchKernelGGL(rc_vec_comb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, out, n);
}
void rc_launch_vec_dots(const VecJob* jobs_dev, int n_jobs, long long n, double* partial, hipStream_t st) {
    if (n_jobs <= 0) return;
    const int nb = (int)((n + 4095) / 4096);
    for (int y0 = 0; y0 < n_jobs; y0 += RC_GRID_Y_MAX)
        hipLaunchKernelGGL(rc_vec_dots_kernel, dim3((unsigned)nb, (unsigned)std::min(RC_GRID_Y_MAX, n_jobs - y0)), dim3(256), 0, st, jobs_dev + y0, n, nb,
                           partial + (long long)y0 * nb);
}
