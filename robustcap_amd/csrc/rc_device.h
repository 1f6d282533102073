// Device-side helpers shared by rc_frame.hip and rc_smplify.hip (gfx950): wave reductions, 3x3 algebra, 6D -> R,
// and the wave-per-body FK + 33-landmark skinning of articulate/model.py:229-241.
#pragma once
#include "rc_internal.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// value of lane `l` (a compile-time constant) on every lane: v_readlane_b32
__device__ __forceinline__ float lane_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float norm3(const float* v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

// C = A * B (3x3 row-major)
__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = (A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c]) + A[3 * r + 2] * B[6 + c];
}
// C = A^T * B
__device__ __forceinline__ void mat3T_mul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = (A[r] * B[c] + A[3 + r] * B[3 + c]) + A[6 + r] * B[6 + c];
}
// y = A * x
__device__ __forceinline__ void mat3_vec(const float* A, const float* x, float* y) {
#pragma unroll
    for (int r = 0; r < 3; ++r) y[r] = (A[3 * r] * x[0] + A[3 * r + 1] * x[1]) + A[3 * r + 2] * x[2];
}

// articulate/math/angular.py:249-264: columns (c0, c1, c0 x c1), NaN -> 0, no epsilon.
__device__ __forceinline__ void r6d_to_R(const float* v, float* R) {
    const float a[3] = {v[0], v[1], v[2]}, b[3] = {v[3], v[4], v[5]};
    const float na = norm3(a);
    const float c0[3] = {a[0] / na, a[1] / na, a[2] / na};
    const float d = (c0[0] * b[0] + c0[1] * b[1]) + c0[2] * b[2];
    const float t[3] = {b[0] - d * c0[0], b[1] - d * c0[1], b[2] - d * c0[2]};
    const float nt = norm3(t);
    const float c1[3] = {t[0] / nt, t[1] / nt, t[2] / nt};
    const float c2[3] = {c0[1] * c1[2] - c0[2] * c1[1], c0[2] * c1[0] - c0[0] * c1[2], c0[0] * c1[1] - c0[1] * c1[0]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        R[3 * r + 0] = (c0[r] != c0[r]) ? 0.0f : c0[r];
        R[3 * r + 1] = (c1[r] != c1[r]) ? 0.0f : c1[r];
        R[3 * r + 2] = (c2[r] != c2[r]) ? 0.0f : c2[r];
    }
}

struct WaveScratch {
    float Rg[24][9];    // global rotations predicted by rnn7
    float Rl[24][9];    // local rotations (pose)
    float G[24][9];     // global rotations re-chained from the local pose
    float P[24][3];     // joint positions, root at the origin
    float T[24][3];     // P_j - G_j * jrest_j   (articulate/model.py:235)
    float J33[33][3];   // landmarks
};

// Body constants are read with data-dependent indices (parent chains): staging them in LDS once per workgroup turns
// ~10 dependent global-load round trips of the tail kernel into LDS reads.
// Two phases, so that a latency-bound kernel can request the constants together with its other reads and store them later:
// 16-byte pieces (both sides are 16-byte aligned: hipMalloc / an aligned __shared__ object), then the few words left over.
template <int NT>
struct BodyStage {
    static constexpr int n = (int)(sizeof(BodyConst) / sizeof(int)), n4 = n / 4, iters = (n4 + NT - 1) / NT;
    int4 v[iters];
    int tail;
    __device__ __forceinline__ void load(const BodyConst* __restrict__ src, int tid) {
        const int4* s4 = reinterpret_cast<const int4*>(src);
#pragma unroll
        for (int q = 0; q < iters; ++q) {
            const int i = tid + q * NT;
            v[q] = i < n4 ? s4[i] : int4{0, 0, 0, 0};
        }
        tail = tid < n - 4 * n4 ? reinterpret_cast<const int*>(src)[4 * n4 + tid] : 0;
    }
    __device__ __forceinline__ void store(BodyConst* dst, int tid) const {
        int4* d4 = reinterpret_cast<int4*>(dst);
#pragma unroll
        for (int q = 0; q < iters; ++q) {
            const int i = tid + q * NT;
            if (i < n4) d4[i] = v[q];
        }
        if (tid < n - 4 * n4) reinterpret_cast<int*>(dst)[4 * n4 + tid] = tail;
    }
};
__device__ __forceinline__ void stage_body(BodyConst* dst, const BodyConst* __restrict__ src, int tid, int nthreads) {
    constexpr int n = (int)(sizeof(BodyConst) / sizeof(int)), n4 = n / 4;
    const int4* s4 = reinterpret_cast<const int4*>(src);
    int4* d4 = reinterpret_cast<int4*>(dst);
    for (int i = tid; i < n4; i += nthreads) d4[i] = s4[i];
    const int* s = reinterpret_cast<const int*>(src);
    int* d = reinterpret_cast<int*>(dst);
    if (tid < n - 4 * n4) d[4 * n4 + tid] = s[4 * n4 + tid];
}

// joint `j` of fk(glb_pose) (net/sig_mp.py:131-135): sum of parent-rotated rest bone vectors, root -> leaf order.
__device__ __forceinline__ void bone_chain(const BodyConst* body, const float (*Rg)[9], int j, float* out) {
    int path[12], n = 0;
    for (int q = j; q > 0; q = body->parent[q]) path[n++] = q;
    out[0] = out[1] = out[2] = 0.0f;
    for (int t = n - 1; t >= 0; --t) {
        const int q = path[t];
        float pb[3];
        mat3_vec(Rg[body->parent[q]], body->bone[q], pb);
        if (t == n - 1) { out[0] = pb[0]; out[1] = pb[1]; out[2] = pb[2]; }
        else { out[0] += pb[0]; out[1] += pb[1]; out[2] += pb[2]; }
    }
}

// Block-wide barrier, or -- in kernels that give every row its own wave AND its own LDS scratch (several rows per workgroup) --
// a wave-local one: the LDS operations of one wave execute in issue order, so all that is needed is that the compiler keeps
// them in program order across this point.
template <bool WAVE_LOCAL>
__device__ __forceinline__ void rc_sync() {
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// ParametricModel.forward_kinematics(calc_mesh=True) restricted to 33 landmarks + sync_mp3d
// (articulate/model.py:229-241, net/sig_mp.py:287-299). Expects s.Rl filled and synced; one wave.
template <bool WAVE_LOCAL = false>
__device__ __forceinline__ void wave_body_fk(const BodyConst* body, WaveScratch& s, const float* tran, int lane) {
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s.G[0][k] = s.Rl[0][k];
        s.P[0][0] = s.P[0][1] = s.P[0][2] = 0.0f;
    }
    rc_sync<WAVE_LOCAL>();
    const int lvl = lane < 24 ? body->level[lane] : -1;
    for (int l = 1; l < 10; ++l) {
        if (lvl == l) {
            const int p = body->parent[lane];
            float g[9], pb[3];
            mat3_mul(s.G[p], s.Rl[lane], g);
            mat3_vec(s.G[p], body->bone[lane], pb);
#pragma unroll
            for (int k = 0; k < 9; ++k) s.G[lane][k] = g[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) s.P[lane][k] = pb[k] + s.P[p][k];
        }
        rc_sync<WAVE_LOCAL>();
    }
    if (lane < 24) {
        float gj[3];
        mat3_vec(s.G[lane], body->jrest[lane], gj);
#pragma unroll
        for (int k = 0; k < 3; ++k) s.T[lane][k] = s.P[lane][k] - gj[k];
    }
    rc_sync<WAVE_LOCAL>();
    if (lane < 33) {
        float out[3];
        const int oj = body->override_joint[lane];
        if (oj >= 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] = s.P[oj][k] + tran[k];
        } else {
            float A[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) A[k] = 0.0f;
            for (int j = 0; j < 24; ++j) {
                const float w = body->w33[lane][j];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    A[4 * r + 0] += w * s.G[j][3 * r + 0];
                    A[4 * r + 1] += w * s.G[j][3 * r + 1];
                    A[4 * r + 2] += w * s.G[j][3 * r + 2];
                    A[4 * r + 3] += w * s.T[j][r];
                }
            }
            const float* v = body->v33[lane];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                out[r] = (((A[4 * r] * v[0] + A[4 * r + 1] * v[1]) + A[4 * r + 2] * v[2]) + A[4 * r + 3]) + tran[r];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) s.J33[lane][k] = out[k];
    }
    rc_sync<WAVE_LOCAL>();
}

// bbox-normalised keypoints of lane `lane` (< 33): xy / max(width, height), rows != 23 relative to row 23.
__device__ __forceinline__ void bbox_normalise(float x, float y, int lane, float& xn, float& yn) {
    const bool on = lane < 33;
    const float inf = __builtin_inff();
    const float w = wave_max(on ? x : -inf) - wave_min(on ? x : inf);
    const float h = wave_max(on ? y : -inf) - wave_min(on ? y : inf);
    const float sc = fmaxf(w, h);
    xn = x / sc;
    yn = y / sc;
    const float hx = __shfl(xn, 23), hy = __shfl(yn, 23);
    if (lane != 23) { xn -= hx; yn -= hy; }
}

// art.math.rotation_matrix_to_axis_angle (angular.py:236-246 loops cv2.Rodrigues on the host): the matrix -> vector
// branch of OpenCV 4.2's cvRodrigues2 restated step by step in float64 (see oracle/sig_mp_oracle.py for the list):
// range check -> nearest orthonormal matrix (OpenCV: U V^T of the SVD; here the same polar factor by scaled Newton
// iteration Q <- (g Q + Q^-T / g) / 2, which converges to U V^T for any non-singular input) -> theta = acos(clamp c),
// s = |vee(Q - Q^T)| / 2 -> s < 1e-5 ? (c > 0 ? 0 : sqrt-diagonal branch) : r theta / (2 s) -> float32.
// PARITY UNPINNED against OpenCV itself (absent); cross-checked against scipy and the SVD-based oracle.
__device__ __forceinline__ void rotmat_to_aa(const float* Rm, float* aa) {
    double Q[9];
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 9; ++q) { Q[q] = (double)Rm[q]; ok = ok && (fabs(Q[q]) < 100.0); }   // NaN fails the compare
    aa[0] = 0.0f; aa[1] = 0.0f; aa[2] = 0.0f;
    if (!ok) return;
    for (int it = 0; it < 24; ++it) {
        double C[9];                                     // cofactors: Q^-T = C / det
        C[0] = Q[4] * Q[8] - Q[5] * Q[7]; C[1] = Q[5] * Q[6] - Q[3] * Q[8]; C[2] = Q[3] * Q[7] - Q[4] * Q[6];
        C[3] = Q[2] * Q[7] - Q[1] * Q[8]; C[4] = Q[0] * Q[8] - Q[2] * Q[6]; C[5] = Q[1] * Q[6] - Q[0] * Q[7];
        C[6] = Q[1] * Q[5] - Q[2] * Q[4]; C[7] = Q[2] * Q[3] - Q[0] * Q[5]; C[8] = Q[0] * Q[4] - Q[1] * Q[3];
        const double det = Q[0] * C[0] + Q[1] * C[1] + Q[2] * C[2];
        if (!(fabs(det) > 1e-300)) return;               // singular input: OpenCV's U V^T is not unique either
        double nq = 0.0, nc = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) { nq += Q[q] * Q[q]; nc += C[q] * C[q]; }
        const double g = sqrt(sqrt(nc) / (fabs(det) * sqrt(nq)));   // sqrt(|Q^-1|_F / |Q|_F)
        double delta = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const double n = 0.5 * (g * Q[q] + C[q] / (det * g));
            delta = fmax(delta, fabs(n - Q[q]));
            Q[q] = n;
        }
        if (delta < 1e-15) break;
    }
    double r[3] = {Q[7] - Q[5], Q[2] - Q[6], Q[3] - Q[1]};
    const double s = sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
    double c = (Q[0] + Q[4] + Q[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0.0) return;
        r[0] = sqrt(fmax((Q[0] + 1.0) * 0.5, 0.0));
        r[1] = sqrt(fmax((Q[4] + 1.0) * 0.5, 0.0)) * (Q[1] < 0.0 ? -1.0 : 1.0);
        r[2] = sqrt(fmax((Q[8] + 1.0) * 0.5, 0.0)) * (Q[2] < 0.0 ? -1.0 : 1.0);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && ((Q[5] > 0.0) != (r[1] * r[2] > 0.0))) r[2] = -r[2];
        const double n = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (!(n > 0.0)) return;
        const double k = theta / n;
        aa[0] = (float)(r[0] * k); aa[1] = (float)(r[1] * k); aa[2] = (float)(r[2] * k);
    } else {
        const double k = theta / (2.0 * s);
        aa[0] = (float)(r[0] * k); aa[1] = (float)(r[1] * k); aa[2] = (float)(r[2] * k);
    }
}
