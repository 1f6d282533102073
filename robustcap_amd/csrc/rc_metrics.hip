// Evaluation metrics on the device, gfx950.
//
// Reference: evaluate.py:120-133 cal_mpjpe -- skin the full mesh for prediction and ground truth (translation zero),
// regress keypoints with J_regressor_h36m, keep the first 14, pelvis-align, then MPJPE, PVE and PA-MPJPE
// (utils.py:138-203: per-frame Procrustes by numpy SVD in a Python loop); articulate/evaluator.py:100-129
// PositionErrorEvaluator. Here one workgroup per frame does all of it and the 2 x 6890 x 3 vertex arrays never exist
// in HBM: both meshes are skinned in registers, the per-vertex distance and the regressor dot products accumulate on the
// fly, and one lane finishes the frame (pelvis alignment, 3x3 Procrustes by a Jacobi eigen-solve in float64).
// HBM/L2 traffic per frame: weights 661 KB + template 83 KB + regressor 386 KB read (L2-resident across frames),
// 12 B written -- against 2 x 83 KB written and read back for the torch formulation.
#include "rc_device.h"

#define MET_MAXK 24        // keypoints per frame: 14 regressor joints (or the 24 SMPL joints when no regressor is set)
#define MET_MAXR 17        // regressor rows accumulated in registers

namespace {

// eigen-decomposition of a symmetric 3x3 matrix by cyclic Jacobi; A is destroyed (its diagonal becomes the spectrum),
// V receives the eigenvectors as columns (a product of rotations: det V = +1).
__device__ void jacobi3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {                       // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {                       // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// mean over the nk points of |s R x1 + t - x2| for the optimal similarity transform (utils.py:138-203).
// K = X1^T X2 = U S V^T; R = V Z U^T with Z = diag(1, 1, sign det(U V^T)): built from the eigenvectors of K^T K,
// u_i = K v_i / s_i for the two largest singular values and u_3 = u_1 x u_2, which absorbs Z.
__device__ double procrustes_error(const float (*p1)[3], const float (*p2)[3], int nk) {
    double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
    for (int j = 0; j < nk; ++j)
        for (int c = 0; c < 3; ++c) { mu1[c] += p1[j][c]; mu2[c] += p2[j][c]; }
    for (int c = 0; c < 3; ++c) { mu1[c] /= nk; mu2[c] /= nk; }
    double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
    for (int j = 0; j < nk; ++j) {
        double a[3], b[3];
        for (int c = 0; c < 3; ++c) { a[c] = p1[j][c] - mu1[c]; b[c] = p2[j][c] - mu2[c]; var1 += a[c] * a[c]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) K[r][c] += a[r] * b[c];
    }
    double A[3][3], V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = K[0][r] * K[0][c] + K[1][r] * K[1][c] + K[2][r] * K[2][c];
    jacobi3(A, V);
    int o[3] = {0, 1, 2};                                            // eigenvalues in descending order
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (A[o[j]][o[j]] < A[o[j + 1]][o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    double v[3][3];                                                  // v[i] = i-th right singular vector
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 3; ++c) v[i][c] = V[c][o[i]];
    {                                                                // proper: v3 = v1 x v2
        v[2][0] = v[0][1] * v[1][2] - v[0][2] * v[1][1];
        v[2][1] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
        v[2][2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
    }
    double u[3][3];
    for (int i = 0; i < 2; ++i) {
        for (int r = 0; r < 3; ++r) u[i][r] = K[r][0] * v[i][0] + K[r][1] * v[i][1] + K[r][2] * v[i][2];
        if (i == 1) {
            const double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
            for (int r = 0; r < 3; ++r) u[1][r] -= d * u[0][r];
        }
        const double n = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + u[i][2] * u[i][2]);
        for (int r = 0; r < 3; ++r) u[i][r] /= fmax(n, 1e-300);
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    double R[3][3], tr = 0.0;                                        // R = sum_i v_i u_i^T ; scale = trace(R K) / var1
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r][c] = v[0][r] * u[0][c] + v[1][r] * u[1][c] + v[2][r] * u[2][c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) tr += R[r][c] * K[c][r];
    const double scale = tr / fmax(var1, 1e-300);
    double err = 0.0;
    for (int j = 0; j < nk; ++j) {
        double a[3], e2 = 0.0;
        for (int c = 0; c < 3; ++c) a[c] = p1[j][c] - mu1[c];
        for (int r = 0; r < 3; ++r) {
            const double h = scale * (R[r][0] * a[0] + R[r][1] * a[1] + R[r][2] * a[2]) - (p2[j][r] - mu2[r]);
            e2 += h * h;
        }
        err += sqrt(e2);
    }
    return err / nk;
}

// one skinned vertex (articulate/model.py:235-241, zero translation). Explicit fmaf and no implicit contraction: the two
// meshes of a frame must go through bit-identical arithmetic so that identical poses give a PVE of exactly zero, as
// they do in the reference.
__device__ __forceinline__ void skin_vertex(const WaveScratch& s, const float* wv, float x, float y, float z, float* out) {
#pragma clang fp contract(off)
    float A[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) A[k] = 0.0f;
    for (int j = 0; j < 24; ++j) {
        const float wj = wv[j];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            A[4 * r + 0] = fmaf(wj, s.G[j][3 * r + 0], A[4 * r + 0]);
            A[4 * r + 1] = fmaf(wj, s.G[j][3 * r + 1], A[4 * r + 1]);
            A[4 * r + 2] = fmaf(wj, s.G[j][3 * r + 2], A[4 * r + 2]);
            A[4 * r + 3] = fmaf(wj, s.T[j][r], A[4 * r + 3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r] = fmaf(A[4 * r + 2], z, fmaf(A[4 * r + 1], y, A[4 * r] * x)) + A[4 * r + 3];
}

}  // namespace

// out[frame] = {MPJPE over the keypoints (pelvis-aligned), PVE, PA-MPJPE}
__global__ __launch_bounds__(256) void rc_mesh_metrics_kernel(const BodyConst* __restrict__ body, const float* __restrict__ vt,
                                                              const float* __restrict__ w, int V, const float* __restrict__ Jr,
                                                              int nk, const float* pose_p, const float* pose_t, float* out) {
    __shared__ WaveScratch sp, st;
    __shared__ float red[4][2 * 3 * MET_MAXR + 1];
    __shared__ float kp[MET_MAXK][3], kt[MET_MAXK][3];
    const long long b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 216; e += 256) { sp.Rl[e / 9][e % 9] = pose_p[b * 216 + e]; st.Rl[e / 9][e % 9] = pose_t[b * 216 + e]; }
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    __syncthreads();
    wave_body_fk(body, sp, zero, tid < 64 ? tid : 64);
    wave_body_fk(body, st, zero, tid < 64 ? tid : 64);

    float pve = 0.0f, ap[MET_MAXR][3], at[MET_MAXR][3];
#pragma unroll
    for (int k = 0; k < MET_MAXR; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) { ap[k][c] = 0.0f; at[k][c] = 0.0f; }
    for (int v = tid; v < V; v += 256) {
        const float* wv = w + (long long)v * 24;
        float xp[3], xt[3];
        const float x = vt[3 * v] - body->jroot[0], y = vt[3 * v + 1] - body->jroot[1], z = vt[3 * v + 2] - body->jroot[2];
        skin_vertex(sp, wv, x, y, z, xp);
        skin_vertex(st, wv, x, y, z, xt);
        const float d[3] = {xt[0] - xp[0], xt[1] - xp[1], xt[2] - xp[2]};
        pve += norm3(d);
        if (Jr) {
#pragma unroll
            for (int k = 0; k < MET_MAXR; ++k) {
                if (k < nk) {
                    const float jw = Jr[(long long)k * V + v];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ap[k][c] += jw * xp[c]; at[k][c] += jw * xt[c]; }
                }
            }
        }
    }
    // workgroup reduction: shuffles inside a wave, LDS across the four waves
    pve = wave_sum(pve);
    if (lane == 0) red[wave][2 * 3 * MET_MAXR] = pve;
    if (Jr) {
#pragma unroll
        for (int k = 0; k < MET_MAXR; ++k) {
            if (k < nk) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a = wave_sum(ap[k][c]), t2 = wave_sum(at[k][c]);
                    if (lane == 0) { red[wave][(k * 3 + c) * 2] = a; red[wave][(k * 3 + c) * 2 + 1] = t2; }
                }
            }
        }
    }
    __syncthreads();
    if (tid < nk * 3) {
        const int k = tid / 3, c = tid % 3;
        if (Jr) {
            kp[k][c] = (red[0][tid * 2] + red[1][tid * 2]) + (red[2][tid * 2] + red[3][tid * 2]);
            kt[k][c] = (red[0][tid * 2 + 1] + red[1][tid * 2 + 1]) + (red[2][tid * 2 + 1] + red[3][tid * 2 + 1]);
        } else {                                                     // no regressor: the SMPL joints stand in
            kp[k][c] = sp.P[k][c];
            kt[k][c] = st.P[k][c];
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float p0[3] = {kp[0][0], kp[0][1], kp[0][2]}, t0[3] = {kt[0][0], kt[0][1], kt[0][2]};
        float mp = 0.0f;
        for (int k = 0; k < nk; ++k) {                               // pelvis alignment, evaluate.py:126-129
            for (int c = 0; c < 3; ++c) { kp[k][c] -= p0[c]; kt[k][c] -= t0[c]; }
            const float d[3] = {kt[k][0] - kp[k][0], kt[k][1] - kp[k][1], kt[k][2] - kp[k][2]};
            mp += norm3(d);
        }
        out[b * 3 + 0] = mp / (float)nk;
        out[b * 3 + 1] = ((red[0][2 * 3 * MET_MAXR] + red[1][2 * 3 * MET_MAXR]) + (red[2][2 * 3 * MET_MAXR] + red[3][2 * 3 * MET_MAXR])) / (float)V;
        out[b * 3 + 2] = (float)procrustes_error(kp, kt, nk);
    }
}

// per-point Euclidean distance (articulate/evaluator.py:129 before the mean): a[n,3], b[n,3] -> d[n]
__global__ void rc_point_distance_kernel(const float* a, const float* b, float* d, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e[3] = {a[3 * i] - b[3 * i], a[3 * i + 1] - b[3 * i + 1], a[3 * i + 2] - b[3 * i + 2]};
    d[i] = norm3(e);
}

// reconstruction_error(S1, S2, reduction=None) (utils.py:189-203) on raw point sets: one lane per frame
__global__ void rc_procrustes_kernel(const float* S1, const float* S2, int nk, float* err, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    err[i] = (float)procrustes_error(reinterpret_cast<const float(*)[3]>(S1 + i * nk * 3), reinterpret_cast<const float(*)[3]>(S2 + i * nk * 3), nk);
}

void rc_launch_procrustes(const float* S1, const float* S2, int nk, float* err, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_procrustes_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, S1, S2, nk, err, n);
}
void rc_launch_mesh_metrics(const BodyConst* body, const float* vt, const float* w, int V, const float* Jr, int nk, const float* pose_p,
                            const float* pose_t, float* out, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_mesh_metrics_kernel, dim3((unsigned)n), dim3(256), 0, st, body, vt, w, V, Jr, nk, pose_p, pose_t, out);
}
void rc_launch_point_distance(const float* a, const float* b, float* d, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_point_distance_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, d, n);
}
