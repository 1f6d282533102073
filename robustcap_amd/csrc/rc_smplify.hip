// smplify optimiser on gfx950: loss and analytic gradient of the temporal body-fitting objective.
//
// Reference (autograd on the host or device through the full 6890-vertex mesh):
//   net/smplify/temporal_smplify.py:150-166  closure: batch_rodrigues -> forward_kinematics(calc_mesh) -> sync_mp3d ->
//                                            temporal_body_fitting_loss(...).backward()
//   net/smplify/losses.py:15-87              the objective (reprojection GMoF, GMM pose prior, angle prior, 3D term,
//                                            gradient-dead IMU term, temporal smoothness in 2D and 3D)
//   net/smplify/prior.py:164-179             MaxMixturePrior.merged_log_likelihood
// Here: parameters x = [T, 72 axis-angle + 3 translation]. Two kernels per evaluation, one workgroup per frame:
//   rc_smplify_fwd_kernel   primal: rotations, FK, 33 landmarks, projection, per-frame loss terms, prior argmin
//   rc_smplify_grad_kernel  wave 0 forms the adjoint lambda = dL/d(landmark) of its frame (incl. the smoothness terms
//                           that couple it to frames t-1 and t+1) and contracts it with the constant skinning data
//                           into per-joint co-factors; 72 threads then push one TANGENT each (d/d axis-angle
//                           component) through Rodrigues and the kinematic chain (only the descendants of the joint
//                           move) and dot it with the co-factors -- forward-mode through the tree, reverse-mode
//                           through everything after it. ~2 MFLOP per frame per evaluation; HBM traffic is the 75
//                           parameters in, 75 gradients out and ~1.5 KB of landmarks per frame.
#include "rc_device.h"

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

// ============================================================================================== forward pass
__global__ __launch_bounds__(64) void rc_smplify_fwd_kernel(SmplifyArgs A, const BodyConst* __restrict__ body_g) {
    __shared__ WaveScratch s;
    __shared__ __attribute__((aligned(16))) BodyConst s_body;
    __shared__ float s_d[SM_DIM];
    const int t = blockIdx.x, lane = threadIdx.x;
    stage_body(&s_body, body_g, lane, 64);
    __syncthreads();
    const BodyConst* body = &s_body;
    const float* aa = A.aa + (long long)t * 72;
    const float tr[3] = {A.tran[t * 3], A.tran[t * 3 + 1], A.tran[t * 3 + 2]};
    frame_primal(body, s, aa, tr, lane);

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
    // GMM prior (prior.py:164-179): min_m 0.5 d^T P_m d - log(nll_w_m)
    float best = 0.0f;
    int bi = 0;
    for (int m = 0; m < SM_NG; ++m) {
        __syncthreads();
        for (int i = lane; i < SM_DIM; i += 64) s_d[i] = aa[3 + i] - A.means[m * SM_DIM + i];
        __syncthreads();
        float q = 0.0f;
        for (int i = lane; i < SM_DIM; i += 64) {
            const float* pr = A.prec + ((long long)m * SM_DIM + i) * SM_DIM;
            float r = 0.0f;
            for (int j = 0; j < SM_DIM; ++j) r += pr[j] * s_d[j];
            q += r * s_d[i];
        }
        const float ll = 0.5f * wave_sum(q) - A.lognll[m];
        if (m == 0 || ll < best) { best = ll; bi = m; }
    }
    loss += 0.01f * best;                                                               // pose_prior_weight ** 2
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
    if (lane == 0) { A.frame_loss[t] = loss; A.imu_loss[t] = imu; A.argmin[t] = bi; }
}

// ============================================================================================= gradient pass
#define TAN_LD 80     // tangent threads (72) padded: [joint][component][thread] layout keeps LDS accesses conflict-free

__global__ __launch_bounds__(128) void rc_smplify_grad_kernel(SmplifyArgs A, const BodyConst* __restrict__ body_g) {
    __shared__ WaveScratch s;
    __shared__ __attribute__((aligned(16))) BodyConst s_body;
    __shared__ float s_lam[33][3];
    __shared__ float s_M[24][9], s_m[24][3], s_o[24][3];
    __shared__ float s_dG[24 * 9 * TAN_LD], s_dP[24 * 3 * TAN_LD];
    const int t = blockIdx.x, tid = threadIdx.x, T = A.T;
    stage_body(&s_body, body_g, tid, 128);
    __syncthreads();
    const BodyConst* body = &s_body;
    const float* aa = A.aa + (long long)t * 72;
    const float tr[3] = {A.tran[t * 3], A.tran[t * 3 + 1], A.tran[t * 3 + 2]};
    frame_primal(body, s, aa, tr, tid < 64 ? tid : 64);              // every thread hits the barriers inside

    // ---- adjoint of the loss with respect to the 33 landmarks of THIS frame ---------------------------------
    float sm = 0.0f;
    if (tid < 33) {
        const int v = tid;
        const float x = s.J33[v][0], y = s.J33[v][1], z = s.J33[v][2];
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
        if (v >= 1) {                                                                  // 3D term, own part
            const float* r = A.ref3d + (long long)t * 99;
#pragma unroll
            for (int c = 0; c < 3; ++c) lam[c] += 2.0f * ((s.J33[v][c] - s.J33[0][c]) - (r[3 * v + c] - r[c]));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) s_lam[v][c] = lam[c];
    }
    if (tid < 64) {
        const float tot = wave_sum(sm);
        if (tid == 0) A.smooth_loss[t] = tot;
    }
    __syncthreads();
    if (tid < 3) {                                                                     // 3D term, landmark 0 part
        const float* r = A.ref3d + (long long)t * 99;
        float acc = 0.0f;
        for (int v = 1; v < 33; ++v) acc += 2.0f * ((s.J33[v][tid] - s.J33[0][tid]) - (r[3 * v + tid] - r[tid]));
        s_lam[0][tid] -= acc;
    }
    __syncthreads();
    // ---- contract lambda with the constant skinning data into per-joint co-factors ---------------------------
    if (tid < 24) {
        const int i = tid;
        float M[9], m3[3], o3[3];
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = 0.0f;
        m3[0] = m3[1] = m3[2] = o3[0] = o3[1] = o3[2] = 0.0f;
        for (int v = 0; v < 33; ++v) {
            const int oj = body->override_joint[v];
            if (oj >= 0) {
                if (oj == i) { o3[0] += s_lam[v][0]; o3[1] += s_lam[v][1]; o3[2] += s_lam[v][2]; }
                continue;
            }
            const float w = body->w33[v][i];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                m3[a] += w * s_lam[v][a];
#pragma unroll
                for (int b = 0; b < 3; ++b) M[3 * a + b] += w * s_lam[v][a] * body->v33[v][b];
            }
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) s_M[i][q] = M[q];
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_m[i][c] = m3[c]; s_o[i][c] = o3[c]; }
    }
    __syncthreads();

    float* gout = A.grad_aa + (long long)t * 72;
    if (tid < 72) {
        // ---- one tangent per thread: d/d aa[joint j][axis a] through Rodrigues and the descendants of j --------
        const int j = tid / 3, a = tid % 3, k = tid;
        float dRl[9];
        batch_rodrigues_tangent(aa + 3 * j, a, dRl);
        unsigned moved = 0u;
        float g = 0.0f;
        for (int i = j; i < 24; ++i) {
            const int p = body->parent[i];
            float dG[9], dP[3];
            if (i == j) {
                if (j == 0) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) dG[q] = dRl[q];
                } else mat3_mul(s.G[p], dRl, dG);
                dP[0] = dP[1] = dP[2] = 0.0f;
            } else if (moved & (1u << p)) {
                float dGp[9], dPp[3];
#pragma unroll
                for (int q = 0; q < 9; ++q) dGp[q] = s_dG[(p * 9 + q) * TAN_LD + k];
#pragma unroll
                for (int q = 0; q < 3; ++q) dPp[q] = s_dP[(p * 3 + q) * TAN_LD + k];
                mat3_mul(dGp, s.Rl[i], dG);
                mat3_vec(dGp, body->bone[i], dP);
#pragma unroll
                for (int q = 0; q < 3; ++q) dP[q] += dPp[q];
            } else continue;
            moved |= 1u << i;
#pragma unroll
            for (int q = 0; q < 9; ++q) s_dG[(i * 9 + q) * TAN_LD + k] = dG[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) s_dP[(i * 3 + q) * TAN_LD + k] = dP[q];
            float gj[3];
            mat3_vec(dG, body->jrest[i], gj);                         // dT = dP - dG * jrest   (model.py:235)
#pragma unroll
            for (int q = 0; q < 9; ++q) g += s_M[i][q] * dG[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) g += s_m[i][q] * (dP[q] - gj[q]) + s_o[i][q] * dP[q];
        }
        // priors act on the 69 non-root components
        if (k >= 3) {
            const int m = A.argmin[t], i = k - 3;
            const float* pr = A.prec + ((long long)m * SM_DIM) * SM_DIM;
            float r = 0.0f;
            for (int q = 0; q < SM_DIM; ++q) {
                const float d = aa[3 + q] - A.means[m * SM_DIM + q];
                r += (pr[i * SM_DIM + q] + pr[q * SM_DIM + i]) * d;   // d/dx of x^T P x  =  (P + P^T) x
            }
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

void rc_launch_smplify(const SmplifyArgs& A, const BodyConst* body, hipStream_t st) {
    if (A.T <= 0) return;
    hipLaunchKernelGGL(rc_smplify_fwd_kernel, dim3(A.T), dim3(64), 0, st, A, body);
    hipLaunchKernelGGL(rc_smplify_grad_kernel, dim3(A.T), dim3(128), 0, st, A, body);
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
    hipLaunchKernelGGL(rc_vec_dots_kernel, dim3((unsigned)nb, (unsigned)n_jobs), dim3(256), 0, st, jobs_dev, n, nb, partial);
}
