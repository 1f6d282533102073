// IMU synthesis of the reference's dataset preparation on the device, gfx950.
//
// Reference: preprocess.py:22-33 `_syn_acc` (second differences of vertex positions at 60 fps, wide stencil in the
// interior) and preprocess.py:206-214 (FK with mesh -> imu_ori = gp[:, ji_mask], imu_acc = _syn_acc(vert[:, vi_mask])).
// The reference skins all 6,890 vertices to keep six; here a 64-lane wave per frame chains the joints and skins only
// the six IMU vertices (read straight from the mesh arrays of rc_set_mesh), then a stencil kernel forms the
// accelerations. HBM: 876 B in, 576 B out per frame.
#include "rc_device.h"

struct ImuPick { int vid[6]; int jid[6]; };

__global__ __launch_bounds__(64) void rc_imu_frame_kernel(const BodyConst* __restrict__ body, const float* __restrict__ vt,
                                                          const float* __restrict__ w, ImuPick pick, const float* pose,
                                                          const float* tran, float* ori, float* joint, float* vert6) {
    __shared__ WaveScratch s;
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < 216; e += 64) s.Rl[e / 9][e % 9] = pose[b * 216 + e];
    const float t[3] = {tran[b * 3], tran[b * 3 + 1], tran[b * 3 + 2]};
    __syncthreads();
    wave_body_fk(body, s, t, lane);
    if (lane < 54) ori[b * 54 + lane] = s.G[pick.jid[lane / 9]][lane % 9];
    if (joint && lane < 24) {
#pragma unroll
        for (int c = 0; c < 3; ++c) joint[(b * 24 + lane) * 3 + c] = s.P[lane][c] + t[c];
    }
    if (lane < 6) {                                                   // same arithmetic as the full-mesh sweep (rc_metrics.hip)
        const int v = pick.vid[lane];
        float A[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) A[k] = 0.0f;
        const float* wv = w + (long long)v * 24;
        for (int j = 0; j < 24; ++j) {
            const float wj = wv[j];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                A[4 * r + 0] += wj * s.G[j][3 * r + 0];
                A[4 * r + 1] += wj * s.G[j][3 * r + 1];
                A[4 * r + 2] += wj * s.G[j][3 * r + 2];
                A[4 * r + 3] += wj * s.T[j][r];
            }
        }
        const float x = vt[3 * v] - body->jroot[0], y = vt[3 * v + 1] - body->jroot[1], z = vt[3 * v + 2] - body->jroot[2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            vert6[(b * 6 + lane) * 3 + r] = (((A[4 * r] * x + A[4 * r + 1] * y) + A[4 * r + 2] * z) + A[4 * r + 3]) + t[r];
    }
}

// _syn_acc (preprocess.py:22-33) on v[T, width]: one lane per element; evaluation order of the reference kept
// ((a + b) - 2 c) * 3600 [/ n^2] so the stencil is bit-exact.
__global__ void rc_syn_acc_kernel(const float* __restrict__ v, float* __restrict__ acc, long long T, long long width, int n) {
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * width) return;
    const long long t = idx / width;
    float out = 0.0f;
    if (n / 2 != 0 && t >= n && t < T - n)
        out = (((v[idx - n * width] + v[idx + n * width]) - 2.0f * v[idx]) * 3600.0f) / (float)(n * n);
    else if (t >= 1 && t < T - 1)
        out = ((v[idx - width] + v[idx + width]) - 2.0f * v[idx]) * 3600.0f;
    acc[idx] = out;
}

void rc_launch_imu_frames(const BodyConst* body, const float* vt, const float* w, const int* vid, const int* jid, const float* pose,
                          const float* tran, float* ori, float* joint, float* vert6, long long T, hipStream_t st) {
    if (T <= 0) return;
    ImuPick pick;
    for (int i = 0; i < 6; ++i) { pick.vid[i] = vid[i]; pick.jid[i] = jid[i]; }
    hipLaunchKernelGGL(rc_imu_frame_kernel, dim3((unsigned)T), dim3(64), 0, st, body, vt, w, pick, pose, tran, ori, joint, vert6);
}
void rc_launch_syn_acc(const float* v, float* acc, long long T, long long width, int n, hipStream_t st) {
    if (T <= 0 || width <= 0) return;
    hipLaunchKernelGGL(rc_syn_acc_kernel, dim3((unsigned)((T * width + 255) / 256)), dim3(256), 0, st, v, acc, T, width, n);
}
