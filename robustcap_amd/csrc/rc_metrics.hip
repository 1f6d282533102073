// Evaluation metrics and full-mesh skinning on the device, gfx950.
//
// Reference: evaluate.py:120-133 cal_mpjpe -- skin the full mesh for prediction and ground truth (translation zero),
// regress keypoints with J_regressor_h36m, keep the first 14, pelvis-align, then MPJPE, PVE and PA-MPJPE
// (utils.py:138-203: per-frame Procrustes by numpy SVD in a Python loop); articulate/evaluator.py:100-129
// PositionErrorEvaluator; articulate/model.py:235-241 for the skinning itself.
//
// Round 3 layout (round 2 ran one workgroup per frame that re-read the whole 1.13 MB constant set -- skinning weights,
// template, regressor rows -- for ONE frame: 32 us per frame, L2-bandwidth bound, 1.1 KB of scratch). Skinning is linear
// in the joint transforms, so everything that does not depend on the pose is folded away first:
//   * regressed keypoint k = sum_v Jr[k,v] sum_j w[v,j] (G_j x_v + T_j) = sum_j (G_j M[k,j] + T_j m[k,j]) with the constants
//     M[k,j] = sum_v Jr[k,v] w[v,j] x_v and m[k,j] = sum_v Jr[k,v] w[v,j] (host, float64, once per mesh / regressor): the
//     regressor never meets a vertex again -- 14 x 24 x 12 FMAs per frame instead of 6,890 x 14 x 6;
//   * the vertex error x_t - x_p = sum_j w[v,j] ((Gt_j - Gp_j) x_v + (Tt_j - Tp_j)): ONE blend per vertex instead of two
//     (identical poses still give a PVE of exactly zero);
//   * rc_metrics_frame_kernel (a wave per frame): both joint chains, the 24 difference transforms to a scratch row
//     (1,152 B), keypoints, pelvis alignment, MPJPE and the float64 Procrustes;
//   * rc_metrics_pve_kernel: a THREAD owns four vertices -- their 96 skinning weights and rest positions stay in registers
//     -- and sweeps a group of frames over them; a frame's difference transforms are wave-uniform reads (scalar loads), so
//     the loop is FMAs only: the constant set is read once per (vertex slab, frame group), not once per frame;
//   * rc_metrics_finish_kernel adds the slab partial sums of a frame in a fixed order (deterministic).
// rc_body_mesh (vertices out, 12 B per vertex and frame: HBM-write bound) has the same thread-per-vertex shape.
#include "rc_device.h"

#define MET_MAXK 24        // keypoints per frame: 14 regressor joints (or the 24 SMPL joints when no regressor is set)
#define MET_VPT 4          // vertices per thread of the sweep kernels
#define MET_SLAB (256 * MET_VPT)
#define MET_FG 16          // frames per workgroup of the sweep kernels

namespace {

// eigen-decomposition of a symmetric 3x3 matrix by cyclic Jacobi; A is destroyed (its diagonal becomes the spectrum),
// V receives the eigenvectors as columns (a product of rotations: det V = +1). Every index is a compile-time constant
// (the p, q, k loops are unrolled): the matrices live in registers, no scratch.
__device__ __forceinline__ void jacobi3(double (&A)[3][3], double (&V)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {                       // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {                       // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
#pragma unroll
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
// p1 / p2 are memory (LDS or global); everything else is named registers.
__device__ double procrustes_error(const float (*p1)[3], const float (*p2)[3], int nk) {
    double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
    for (int j = 0; j < nk; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { mu1[c] += p1[j][c]; mu2[c] += p2[j][c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { mu1[c] /= nk; mu2[c] /= nk; }
    double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
    for (int j = 0; j < nk; ++j) {
        double a[3], b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { a[c] = p1[j][c] - mu1[c]; b[c] = p2[j][c] - mu2[c]; var1 += a[c] * a[c]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) K[r][c] += a[r] * b[c];
    }
    double A[3][3], V[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[r][c] = K[0][r] * K[0][c] + K[1][r] * K[1][c] + K[2][r] * K[2][c];
    jacobi3(A, V);
    // the two leading right singular vectors: eigen-columns of the two largest eigenvalues (selects, no index arrays)
    const double l0 = A[0][0], l1 = A[1][1], l2 = A[2][2];
    const int first = (l0 >= l1 && l0 >= l2) ? 0 : ((l1 >= l2) ? 1 : 2);
    const int second = first == 0 ? (l1 >= l2 ? 1 : 2) : (first == 1 ? (l0 >= l2 ? 0 : 2) : (l0 >= l1 ? 0 : 1));
    double v[3][3];                                                  // v[i] = i-th right singular vector
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v[0][c] = first == 0 ? V[c][0] : (first == 1 ? V[c][1] : V[c][2]);
        v[1][c] = second == 0 ? V[c][0] : (second == 1 ? V[c][1] : V[c][2]);
    }
    v[2][0] = v[0][1] * v[1][2] - v[0][2] * v[1][1];                 // proper: v3 = v1 x v2
    v[2][1] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
    v[2][2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
    double u[3][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u[i][r] = K[r][0] * v[i][0] + K[r][1] * v[i][1] + K[r][2] * v[i][2];
        if (i == 1) {
            const double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
#pragma unroll
            for (int r = 0; r < 3; ++r) u[1][r] -= d * u[0][r];
        }
        const double n = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + u[i][2] * u[i][2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) u[i][r] /= fmax(n, 1e-300);
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    double R[3][3], tr = 0.0;                                        // R = sum_i v_i u_i^T ; scale = trace(R K) / var1
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r][c] = v[0][r] * u[0][c] + v[1][r] * u[1][c] + v[2][r] * u[2][c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) tr += R[r][c] * K[c][r];
    const double scale = tr / fmax(var1, 1e-300);
    double err = 0.0;
    for (int j = 0; j < nk; ++j) {
        double a[3], e2 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = p1[j][c] - mu1[c];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double h = scale * (R[r][0] * a[0] + R[r][1] * a[1] + R[r][2] * a[2]) - (p2[j][r] - mu2[r]);
            e2 += h * h;
        }
        err += sqrt(e2);
    }
    return err / nk;
}

}  // namespace

// One wave per frame: joint chains of prediction and ground truth (zero translation), the 24 difference transforms
// xf[frame][j] = (Gt_j - Gp_j | Tt_j - Tp_j) as 12 floats (row r: 3 rotation entries then the offset), keypoints,
// out[frame] = {MPJPE over the keypoints (pelvis-aligned), -, PA-MPJPE}. kM [nk][24][4]: the folded regressor
// (M[k,j] xyz, m[k,j]); nullptr: the 24 SMPL joints stand in for the regressor joints.
__global__ __launch_bounds__(64) void rc_metrics_frame_kernel(const BodyConst* __restrict__ body_g, const float* __restrict__ kM, int nk,
                                                              const float* __restrict__ pose_p, const float* __restrict__ pose_t,
                                                              float* __restrict__ xf, float* __restrict__ out) {
    __shared__ WaveScratch sp, st;
    __shared__ __attribute__((aligned(16))) BodyConst s_body;
    __shared__ float kp[MET_MAXK][3], kt[MET_MAXK][3];
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    stage_body(&s_body, body_g, lane, 64);
    for (int e = lane; e < 216; e += 64) { sp.Rl[e / 9][e % 9] = pose_p[b * 216 + e]; st.Rl[e / 9][e % 9] = pose_t[b * 216 + e]; }
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    __syncthreads();
    wave_body_fk(&s_body, sp, zero, lane);
    wave_body_fk(&s_body, st, zero, lane);
    for (int e = lane; e < 288; e += 64) {
        const int j = e / 12, q = e % 12, r = q >> 2, c = q & 3;
        xf[b * 288 + e] = c < 3 ? st.G[j][3 * r + c] - sp.G[j][3 * r + c] : st.T[j][r] - sp.T[j][r];
    }
    for (int e = lane; e < nk * 3; e += 64) {
        const int k = e / 3, c = e % 3;
        if (kM) {
            float ap = 0.0f, at = 0.0f;
            for (int j = 0; j < 24; ++j) {
                const float* m = kM + ((long long)k * 24 + j) * 4;
                ap += ((sp.G[j][3 * c] * m[0] + sp.G[j][3 * c + 1] * m[1]) + sp.G[j][3 * c + 2] * m[2]) + sp.T[j][c] * m[3];
                at += ((st.G[j][3 * c] * m[0] + st.G[j][3 * c + 1] * m[1]) + st.G[j][3 * c + 2] * m[2]) + st.T[j][c] * m[3];
            }
            kp[k][c] = ap; kt[k][c] = at;
        } else {
            kp[k][c] = sp.P[k][c]; kt[k][c] = st.P[k][c];
        }
    }
    __syncthreads();
    if (lane == 0) {
        const float p0[3] = {kp[0][0], kp[0][1], kp[0][2]}, t0[3] = {kt[0][0], kt[0][1], kt[0][2]};
        float mp = 0.0f;
        for (int k = 0; k < nk; ++k) {                               // pelvis alignment, evaluate.py:126-129
#pragma unroll
            for (int c = 0; c < 3; ++c) { kp[k][c] -= p0[c]; kt[k][c] -= t0[c]; }
            const float d[3] = {kt[k][0] - kp[k][0], kt[k][1] - kp[k][1], kt[k][2] - kp[k][2]};
            mp += norm3(d);
        }
        out[b * 3 + 0] = mp / (float)nk;
        out[b * 3 + 2] = (float)procrustes_error(kp, kt, nk);
    }
}

// vertex data of the four vertices a thread owns (slab * MET_SLAB + q * 256 + tid): rest position and skinning weights
struct VertexRegs {
    float x[MET_VPT][3];
    float w[MET_VPT][24];
};
__device__ __forceinline__ void load_vertices(VertexRegs& r, const float* __restrict__ vt, const float* __restrict__ w,
                                              const float* jroot, int V, int v0, int tid) {
#pragma unroll
    for (int q = 0; q < MET_VPT; ++q) {
        const int v = v0 + q * 256 + tid;
        const bool ok = v < V;
        const int vv = ok ? v : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) r.x[q][c] = ok ? vt[3 * vv + c] - jroot[c] : 0.0f;
#pragma unroll
        for (int j = 0; j < 24; ++j) r.w[q][j] = ok ? w[(long long)vv * 24 + j] : 0.0f;     // w = 0: a padding vertex adds nothing
    }
}

// PVE partial sums: part[slab][frame] = sum over the slab's vertices of |x_t - x_p|. The 288 floats of a frame's difference
// transforms are read with wave-uniform addresses (scalar loads), the weights sit in registers.
__global__ __launch_bounds__(256) void rc_metrics_pve_kernel(const BodyConst* __restrict__ body, const float* __restrict__ vt,
                                                             const float* __restrict__ w, int V, const float* __restrict__ xf,
                                                             long long n, float* __restrict__ part) {
    __shared__ float s_sum[4][MET_FG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slab = blockIdx.x;
    const long long f0 = (long long)blockIdx.y * MET_FG;
    VertexRegs r;
    const float jr[3] = {body->jroot[0], body->jroot[1], body->jroot[2]};
    load_vertices(r, vt, w, jr, V, slab * MET_SLAB, tid);
    const int nf = (int)min((long long)MET_FG, n - f0);
    for (int f = 0; f < nf; ++f) {
        const float* __restrict__ X = xf + (f0 + f) * 288;
        float acc[MET_VPT][12];
#pragma unroll
        for (int q = 0; q < MET_VPT; ++q)
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[q][k] = 0.0f;
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            float xj[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) xj[k] = X[j * 12 + k];
#pragma unroll
            for (int q = 0; q < MET_VPT; ++q)
#pragma unroll
                for (int k = 0; k < 12; ++k) acc[q][k] = fmaf(r.w[q][j], xj[k], acc[q][k]);
        }
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < MET_VPT; ++q) {
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                d[c] = fmaf(acc[q][4 * c + 2], r.x[q][2], fmaf(acc[q][4 * c + 1], r.x[q][1], acc[q][4 * c] * r.x[q][0])) + acc[q][4 * c + 3];
            s += norm3(d);
        }
        s = wave_sum(s);
        if (lane == 0) s_sum[wave][f] = s;
    }
    __syncthreads();
    if (tid < nf) part[(long long)slab * n + f0 + tid] = (s_sum[0][tid] + s_sum[1][tid]) + (s_sum[2][tid] + s_sum[3][tid]);
}

__global__ void rc_metrics_finish_kernel(const float* __restrict__ part, int n_slab, int V, long long n, float* __restrict__ out) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    float s = 0.0f;
    for (int q = 0; q < n_slab; ++q) s += part[(long long)q * n + f];
    out[f * 3 + 1] = s / (float)V;
}

// Full-mesh linear-blend skinning (articulate/model.py:235-241). rc_mesh_frame_kernel chains the joints of a frame and
// stores its 24 transforms (G_j | T_j, 12 floats each); rc_body_mesh_sweep_kernel: thread = four vertices (weights in
// registers) x a group of frames. HBM-write bound: 12 B out per vertex and frame.
__global__ __launch_bounds__(64) void rc_mesh_frame_kernel(const BodyConst* __restrict__ body_g, const float* __restrict__ pose,
                                                           float* __restrict__ xf) {
    __shared__ WaveScratch s;
    __shared__ __attribute__((aligned(16))) BodyConst s_body;
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    stage_body(&s_body, body_g, lane, 64);
    for (int e = lane; e < 216; e += 64) s.Rl[e / 9][e % 9] = pose[b * 216 + e];
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    __syncthreads();
    wave_body_fk(&s_body, s, zero, lane);
    for (int e = lane; e < 288; e += 64) {
        const int j = e / 12, q = e % 12, r = q >> 2, c = q & 3;
        xf[b * 288 + e] = c < 3 ? s.G[j][3 * r + c] : s.T[j][r];
    }
}

__global__ __launch_bounds__(256) void rc_body_mesh_sweep_kernel(const BodyConst* __restrict__ body, const float* __restrict__ vt,
                                                                 const float* __restrict__ w, int V, const float* __restrict__ xf,
                                                                 const float* __restrict__ tran, long long n, float* __restrict__ vert) {
    const int tid = threadIdx.x;
    const int slab = blockIdx.x;
    const long long f0 = (long long)blockIdx.y * MET_FG;
    VertexRegs r;
    const float jr[3] = {body->jroot[0], body->jroot[1], body->jroot[2]};
    load_vertices(r, vt, w, jr, V, slab * MET_SLAB, tid);
    const int nf = (int)min((long long)MET_FG, n - f0);
    for (int f = 0; f < nf; ++f) {
        const float* __restrict__ X = xf + (f0 + f) * 288;
        const float t[3] = {tran[(f0 + f) * 3], tran[(f0 + f) * 3 + 1], tran[(f0 + f) * 3 + 2]};
        float acc[MET_VPT][12];
#pragma unroll
        for (int q = 0; q < MET_VPT; ++q)
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[q][k] = 0.0f;
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            float xj[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) xj[k] = X[j * 12 + k];
#pragma unroll
            for (int q = 0; q < MET_VPT; ++q)
#pragma unroll
                for (int k = 0; k < 12; ++k) acc[q][k] += r.w[q][j] * xj[k];
        }
#pragma unroll
        for (int q = 0; q < MET_VPT; ++q) {
            const int v = slab * MET_SLAB + q * 256 + tid;
            if (v < V) {
                float* o = vert + ((f0 + f) * V + v) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    o[c] = (((acc[q][4 * c] * r.x[q][0] + acc[q][4 * c + 1] * r.x[q][1]) + acc[q][4 * c + 2] * r.x[q][2]) + acc[q][4 * c + 3]) + t[c];
            }
        }
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
// scratch: n * 288 floats (difference transforms) + ceil(V / 1024) * n floats (slab partial sums); kM: folded regressor or nullptr
long long rc_mesh_metrics_scratch_floats(int V, long long n) { return n * 288 + (long long)((V + MET_SLAB - 1) / MET_SLAB) * n; }
void rc_launch_mesh_metrics(const BodyConst* body, const float* vt, const float* w, int V, const float* kM, int nk, const float* pose_p,
                            const float* pose_t, float* out, long long n, float* scratch, hipStream_t st) {
    if (n <= 0) return;
    const int n_slab = (V + MET_SLAB - 1) / MET_SLAB;
    float* xf = scratch;
    float* part = scratch + n * 288;
    hipLaunchKernelGGL(rc_metrics_frame_kernel, dim3((unsigned)n), dim3(64), 0, st, body, kM, nk, pose_p, pose_t, xf, out);
    hipLaunchKernelGGL(rc_metrics_pve_kernel, dim3((unsigned)n_slab, (unsigned)((n + MET_FG - 1) / MET_FG)), dim3(256), 0, st, body, vt, w, V, xf, n, part);
    hipLaunchKernelGGL(rc_metrics_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, n_slab, V, n, out);
}
long long rc_body_mesh_scratch_floats(long long n) { return n * 288; }
void rc_launch_body_mesh(const BodyConst* body, const float* vt, const float* w, int V, const float* pose, const float* tran,
                         float* vert, long long n, float* scratch, hipStream_t st) {
    if (n <= 0) return;
    const int n_slab = (V + MET_SLAB - 1) / MET_SLAB;
    hipLaunchKernelGGL(rc_mesh_frame_kernel, dim3((unsigned)n), dim3(64), 0, st, body, pose, scratch);
    hipLaunchKernelGGL(rc_body_mesh_sweep_kernel, dim3((unsigned)n_slab, (unsigned)((n + MET_FG - 1) / MET_FG)), dim3(256), 0, st, body, vt, w, V,
                       scratch, tran, n, vert);
}
void rc_launch_point_distance(const float* a, const float* b, float* d, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_point_distance_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, d, n);
}
