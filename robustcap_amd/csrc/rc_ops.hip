// Per-op kernels behind the rest of the reference's ParametricModel / art.math call surface:
//   forward_kinematics_R            articulate/math/spatial.py:170-194 (_forward_tree with bmm), model.py:131-145
//   bone_vector_to_joint_position   spatial.py:126-145          joint_position_to_bone_vector   spatial.py:148-167
//   get_zero_pose_joint_and_vertex  model.py:78-93 (shape=None)
//   rotation_matrix_to_r6d          articulate/math/angular.py:267-274
//   normalize_tensor, lerp          articulate/math/general.py:15-39
//   angle_between                   angular.py:128-141 (rotation matrices; norm of the Rodrigues vector of R1^T R2)
//   bbox normalisation              net/sig_mp.py:150-152 with get_bbox_scale L277-284
// HBM-bound sweeps (a few hundred bytes per item), one thread or one wave per item; used by the drop-in Python surface and
// by the per-op parity tests against the reference's captured vectors.
#include "rc_internal.h"
#include "rc_device.h"

// R_global[i] = R_global[parent[i]] . R_local[i]; one wave per body, joints of one tree level in parallel
__global__ __launch_bounds__(64) void rc_fk_r_kernel(const BodyConst* __restrict__ body, const float* Rl, float* Rg) {
    __shared__ float sL[24][9];
    __shared__ float sG[24][9];
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < 216; e += 64) sL[e / 9][e % 9] = Rl[b * 216 + e];
    __syncthreads();
    for (int lvl = 0; lvl < 10; ++lvl) {
        if (lane < 24 && body->level[lane] == lvl) {
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) sG[0][k] = sL[0][k];
            } else {
                float M[9];
                mat3_mul(sG[body->parent[lane]], sL[lane], M);
#pragma unroll
                for (int k = 0; k < 9; ++k) sG[lane][k] = M[k];
            }
        }
        __syncthreads();
    }
    for (int e = lane; e < 216; e += 64) Rg[b * 216 + e] = sG[e / 9][e % 9];
}

// joint[i] = joint[parent[i]] + bone[i] (root: bone[0]); the adds run root-to-leaf like the reference's _forward_tree
__global__ __launch_bounds__(192) void rc_bone_to_joint_kernel(const BodyConst* __restrict__ body, const float* bone, float* joint, long long n) {
    __shared__ float acc[24][192];                       // running sums of this thread's (body, component), parent-indexed
    const long long idx = (long long)blockIdx.x * 192 + threadIdx.x;
    const long long b = idx / 3;
    const int c = (int)(idx % 3), t = threadIdx.x;
    if (b >= n) return;
    acc[0][t] = bone[b * 72 + c];
    for (int i = 1; i < 24; ++i) acc[i][t] = acc[body->parent[i]][t] + bone[b * 72 + 3 * i + c];
    for (int i = 0; i < 24; ++i) joint[b * 72 + 3 * i + c] = acc[i][t];
}

// bone[i] = (-joint[parent[i]]) + joint[i]  (root: joint[0])
__global__ void rc_joint_to_bone_kernel(const BodyConst* __restrict__ body, const float* joint, float* bone, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = idx / 72;
    const int e = (int)(idx % 72), i = e / 3, c = e % 3;
    if (b >= n) return;
    const float v = joint[b * 72 + e];
    bone[b * 72 + e] = i == 0 ? v : (-joint[b * 72 + 3 * body->parent[i] + c]) + v;
}

// model.py:86-87: j = J - J[:1], v = v_template - J[:1]
__global__ void rc_zero_pose_kernel(const BodyConst* __restrict__ body, const float* __restrict__ vt, int V, float* joint, float* vert) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 72) joint[idx] = body->jrest[idx / 3][idx % 3];
    if (vert && idx < 3 * V) vert[idx] = vt[idx] - body->jroot[idx % 3];
}

// r[:, :, :2].transpose(1, 2) -> (R00, R10, R20, R01, R11, R21)
__global__ void rc_rotmat_to_r6d_kernel(const float* R, float* r6d, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = idx / 6;
    const int e = (int)(idx % 6);
    if (i >= n) return;
    r6d[idx] = R[9 * i + 3 * (e % 3) + e / 3];
}

// a * w1 + b * w2 with w1 = float(1 - t), w2 = float(t) formed on the host in double (tensor * python float)
__global__ void rc_lerp_kernel(const float* a, const float* b, float w1, float w2, float* out, long long n) {
#pragma clang fp contract(off)                           // torch: two multiplies and an add, no fma
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * w1 + b[i] * w2;
}

// x / |x| over the last dimension (width <= 64 per wave pass; wider rows loop), optional norms
__global__ __launch_bounds__(64) void rc_normalize_rows_kernel(const float* x, float* out, float* norm, int width) {
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    float ss = 0.f;
    for (int k = lane; k < width; k += 64) { const float v = x[r * width + k]; ss += v * v; }
    const float nrm = sqrtf(wave_sum(ss));
    for (int k = lane; k < width; k += 64) out[r * width + k] = x[r * width + k] / nrm;
    if (norm && lane == 0) norm[r] = nrm;
}

// |Rodrigues vector of R1^T R2| (angular.py:137-140)
__global__ void rc_angle_between_kernel(const float* R1, const float* R2, float* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float D[9], aa[3];
    mat3T_mul(R1 + 9 * i, R2 + 9 * i, D);
    rotmat_to_aa(D, aa);
    out[i] = norm3(aa);
}

// sig_mp.py:150-152: xy / max(bbox width, height); every row but 23 relative to row 23; confidence untouched
__global__ __launch_bounds__(64) void rc_bbox_normalise_kernel(const float* kp, float* out) {
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    float x = 0.f, y = 0.f, cf = 0.f;
    if (lane < 33) { x = kp[(b * 33 + lane) * 3]; y = kp[(b * 33 + lane) * 3 + 1]; cf = kp[(b * 33 + lane) * 3 + 2]; }
    float xn, yn;
    bbox_normalise(x, y, lane, xn, yn);
    if (lane < 33) { out[(b * 33 + lane) * 3] = xn; out[(b * 33 + lane) * 3 + 1] = yn; out[(b * 33 + lane) * 3 + 2] = cf; }
}

// get_zero_pose_joint_and_vertex(shape) before the root alignment (articulate/model.py:88-91):
//   v = tensordot(shape, shapedirs) + v_template  (one thread per vertex coordinate, 10 blend directions)
//   j = J_regressor . v                           (one workgroup per (joint, coordinate), reduced over the V vertices)
__global__ void rc_shape_vertices_kernel(const float* __restrict__ vt, const float* __restrict__ sd, const float* __restrict__ beta,
                                         int V, float* v) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * V) return;
    float acc = 0.f;
#pragma unroll
    for (int b = 0; b < 10; ++b) acc += beta[b] * sd[(long long)idx * 10 + b];
    v[idx] = acc + vt[idx];
}
__global__ __launch_bounds__(256) void rc_shape_joints_kernel(const float* __restrict__ Jr, const float* __restrict__ v, int V, float* j) {
    __shared__ float part[4];
    const int joint = blockIdx.x / 3, c = blockIdx.x % 3;
    float acc = 0.f;
    for (int k = threadIdx.x; k < V; k += 256) acc += Jr[(long long)joint * V + k] * v[3 * k + c];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) j[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

static inline unsigned blocks_of(long long n, int per) { return (unsigned)((n + per - 1) / per); }

void rc_launch_shape_body(const float* vt, const float* sd, const float* beta, const float* Jr, int V, float* v, float* j, hipStream_t st) {
    hipLaunchKernelGGL(rc_shape_vertices_kernel, dim3(blocks_of(3ll * V, 256)), dim3(256), 0, st, vt, sd, beta, V, v);
    hipLaunchKernelGGL(rc_shape_joints_kernel, dim3(72), dim3(256), 0, st, Jr, v, V, j);
}

void rc_launch_fk_r(const BodyConst* body, const float* Rl, float* Rg, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(rc_fk_r_kernel, dim3((unsigned)n), dim3(64), 0, st, body, Rl, Rg);
}
void rc_launch_bone_to_joint(const BodyConst* body, const float* bone, float* joint, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(rc_bone_to_joint_kernel, dim3(blocks_of(n * 3, 192)), dim3(192), 0, st, body, bone, joint, n);
}
void rc_launch_joint_to_bone(const BodyConst* body, const float* joint, float* bone, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(rc_joint_to_bone_kernel, dim3(blocks_of(n * 72, 256)), dim3(256), 0, st, body, joint, bone, n);
}
void rc_launch_zero_pose(const BodyConst* body, const float* vt, int V, float* joint, float* vert, hipStream_t st) {
    const long long n = vert ? (3ll * V > 72 ? 3ll * V : 72) : 72;
    hipLaunchKernelGGL(rc_zero_pose_kernel, dim3(blocks_of(n, 256)), dim3(256), 0, st, body, vt, V, joint, vert);
}
void rc_launch_rotmat_to_r6d(const float* R, float* r6d, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(rc_rotmat_to_r6d_kernel, dim3(blocks_of(n * 6, 256)), dim3(256), 0, st, R, r6d, n);
}
void rc_launch_lerp(const float* a, const float* b, float w1, float w2, float* out, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(rc_lerp_kernel, dim3(blocks_of(n, 256)), dim3(256), 0, st, a, b, w1, w2, out, n);
}
void rc_launch_normalize_rows(const float* x, float* out, float* norm, long long rows, int width, hipStream_t st) {
    if (rows > 0) hipLaunchKernelGGL(rc_normalize_rows_kernel, dim3((unsigned)rows), dim3(64), 0, st, x, out, norm, width);
}
void rc_launch_angle_between(const float* R1, const float* R2, float* out, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(rc_angle_between_kernel, dim3(blocks_of(n, 256)), dim3(256), 0, st, R1, R2, out, n);
}
void rc_launch_bbox_normalise(const float* kp, float* out, long long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(rc_bbox_normalise_kernel, dim3((unsigned)n), dim3(64), 0, st, kp, out);
}
