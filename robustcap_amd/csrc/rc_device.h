// Device-side helpers shared by rc_frame.hip and rc_smplify.hip (gfx950): wave reductions, 3x3 algebra, 6D -> R,
// and the wave-per-body FK + 33-landmark skinning of articulate/model.py:229-241.
#pragma once
#include "rc_internal.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
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
__device__ __forceinline__ void stage_body(BodyConst* dst, const BodyConst* __restrict__ src, int tid, int nthreads) {
    const int n = (int)(sizeof(BodyConst) / sizeof(int));
    const int* s = reinterpret_cast<const int*>(src);
    int* d = reinterpret_cast<int*>(dst);
    for (int i = tid; i < n; i += nthreads) d[i] = s[i];
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

// ParametricModel.forward_kinematics(calc_mesh=True) restricted to 33 landmarks + sync_mp3d
// (articulate/model.py:229-241, net/sig_mp.py:287-299). Expects s.Rl filled and synced; one wave.
__device__ __forceinline__ void wave_body_fk(const BodyConst* body, WaveScratch& s, const float* tran, int lane) {
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s.G[0][k] = s.Rl[0][k];
        s.P[0][0] = s.P[0][1] = s.P[0][2] = 0.0f;
    }
    __syncthreads();
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
        __syncthreads();
    }
    if (lane < 24) {
        float gj[3];
        mat3_vec(s.G[lane], body->jrest[lane], gj);
#pragma unroll
        for (int k = 0; k < 3; ++k) s.T[lane][k] = s.P[lane][k] - gj[k];
    }
    __syncthreads();
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
    __syncthreads();
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

// art.math.rotation_matrix_to_axis_angle (angular.py:236-246 loops cv2.Rodrigues on the host). Restated from the
// Rodrigues formula in float64: theta = atan2(|v|, (tr - 1) / 2), v = vee(R - R^T) / 2; near pi the axis comes
// from the symmetric part. PARITY UNPINNED against OpenCV (absent); validated by round trip.
__device__ __forceinline__ void rotmat_to_aa(const float* Rm, float* aa) {
    double R[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = (double)Rm[q];
    const double v[3] = {(R[7] - R[5]) * 0.5, (R[2] - R[6]) * 0.5, (R[3] - R[1]) * 0.5};
    const double sn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    const double th = atan2(sn, c);
    double o[3];
    if (sn > 1e-9) {
        const double kf = th / sn;
        o[0] = v[0] * kf; o[1] = v[1] * kf; o[2] = v[2] * kf;
    } else if (c > 0.0) {
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    } else {                                     // theta = pi: R + I = 2 a a^T
        const double d[3] = {sqrt(fmax((R[0] + 1.0) * 0.5, 0.0)), sqrt(fmax((R[4] + 1.0) * 0.5, 0.0)), sqrt(fmax((R[8] + 1.0) * 0.5, 0.0))};
        int m = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
        double ax[3] = {d[0], d[1], d[2]};
        for (int q = 0; q < 3; ++q)
            if (q != m && (R[3 * m + q] + R[3 * q + m]) < 0.0) ax[q] = -ax[q];
        const double pi = 3.14159265358979323846;
        o[0] = ax[0] * pi; o[1] = ax[1] * pi; o[2] = ax[2] * pi;
    }
    aa[0] = (float)o[0]; aa[1] = (float)o[1]; aa[2] = (float)o[2];
}
