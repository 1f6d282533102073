// Per-frame fusion logic of sig_mp on gfx950: everything of Net.forward_online (net/sig_mp.py:113-274) that is
// not a sub-net GEMM, as three wave-per-body kernels with all branching on device (the reference syncs the host
// with .item() six times per frame):
//   rc_prep_kernel  L138-152  mean confidence -> regime / row flags, root-frame IMU transform, bbox
//                             normalisation, concatenated sub-net inputs
//   rc_fuse_kernel  L154-167, L178-180  camera->root rotation of rnn4's joints, lerp, init_net trigger
//   rc_tail_kernel  L173-273  6D->R, IK, foot FK, translation / contact / floor logic, full FK + 33-landmark
//                             skinning + sync_mp3d, vision-updater inputs, init_net state write
// One 64-lane wave owns one body: lanes map to keypoints (33), joints (24) or matrix entries; reductions are
// DPP/shuffle butterflies; intermediates live in LDS. HBM traffic per body-frame is the 171-float input row,
// the 219-float output row and ~3 KB of padded sub-net input rows -- these kernels are latency-, not
// bandwidth-bound, which is why they are kept to three launches.
#include "rc_internal.h"
#include <hip/hip_ext.h>
#include <cstdlib>

#include "rc_device.h"
#include "rc_frame_dev.h"     // prep / tail of one row (shared with rc_live.hip)

__global__ __launch_bounds__(64) void rc_prep_kernel(FrameBuffers fb, FrameIO io, rc_params_dev prm, int B, int first_frame) {
    const int row = blockIdx.x;
    prep_body(fb, io, prm, row, threadIdx.x, first_frame, fb.pend[row], fb.uv_count[row]);
}

// Per-row-cursor wavefront engine (rc_api.cpp: run_wave2_segment): the prep of one TICK. Row `row` starts frame
// w.frame_at[row] of the call in ring slot `fb`, or nothing (-1: the row waits for a feedback step of an earlier frame, or has
// no frame left). It opens the step of every sub-net the frame will take -- the step NUMBER travels with the slot (wsteps), so
// that the stages of several frames of a row can be in flight while the row's counters move on. On the first tick of a
// segment a deferred updater step left pending by the frames before it (fb.pend) becomes a rider of this slot: its inputs are
// copied from the context's own buffers and the row joins the slot's rnn4 / rnn6 launches (net/sig_mp.py:264-271).
// Four rows per workgroup, a wave each (no LDS, no barrier): the kernel runs beside the wide GEMM tiles, whose 512-register waves
// need a whole CU -- a one-wave workgroup parked on one SIMD keeps the other three idle and the next tile waiting.
template <int RPB>
__global__ __launch_bounds__(64 * RPB) void rc_prep_wave_kernel(FrameBuffers fb, FrameIO io, rc_params_dev prm, int B, WavePrep w) {
    const int row = blockIdx.x * RPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const int f = w.frame_at[row];
    const bool rider = w.first_tick && prm.use_vision_updater && fb.pend[row] != 0;
    unsigned fl = 0;
    if (f >= 0) {
        FrameIO iof = io;
        iof.j2d += (long long)f * 99; iof.acc += (long long)f * 18; iof.ori += (long long)f * 54;
        fl = prep_body(fb, iof, prm, row, lane, 0, 0, 0, RC_ROW2_VALID);
    }
    if (lane == 0) {
        fb.frame[row] = f;
        unsigned f2 = f >= 0 ? (RC_ROW2_VALID | ((fl & RC_ROW_VIS) ? RC_ROW2_M4 : 0u) | ((fl & RC_ROW_PC) ? RC_ROW2_M6 : 0u)) : 0u;
        if (f < 0) fb.flags[row] = 0;
        else {
            const int always[4] = {0, 1, 4, 5};                            // rnn2, rnn3, rnn7, rnn8 step on every frame
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int n = always[q]; fb.wsteps[n * B + row] = ++w.steps[n][row]; }
            if (fl & RC_ROW_VIS) fb.wsteps[2 * B + row] = ++w.steps[2][row];   // rnn4 on camera keypoints (L149-153)
            if (fl & RC_ROW_PC) fb.wsteps[3 * B + row] = ++w.steps[3][row];    // rnn6 (L161 / L165)
        }
        if (rider) {
            fb.wsteps[2 * B + row] = ++w.steps[2][row];
            fb.wsteps[3 * B + row] = ++w.steps[3][row];
            f2 |= RC_ROW2_M4 | RC_ROW2_M6;
            fb.pend[row] = 0;
        }
        fb.flags2[row] = (unsigned char)f2;
    }
    if (rider) {
        for (int k = lane; k < 256; k += 64) {
            fb.x4l[rc_pk(row, k, LD_X4)] = w.cx4l[rc_pk(row, k, LD_X4)];
            fb.x6l[rc_pk(row, k, LD_X6)] = w.cx6l[rc_pk(row, k, LD_X6)];
        }
    }
}

// =================================================================================== fuse (L154-167, L178-180)
__device__ __forceinline__ void fuse_body(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, const int B, const int idx) {
    const int row = idx / 24, j = idx % 24;
    if (row >= B) return;
    int frame = 0;
    if (fb.frame) { frame = fb.frame[row]; if (frame < 0) return; }       // ring slot of the per-row-cursor engine: bubble
    const int regime = fb.regime[row];
    if (j == 23) {                                                        // L178-180
        if (regime == 2 && prm.use_imu_updater && fb.first_reach[row]) {
            fb.first_reach[row] = 0;
            fb.flags[row] |= RC_ROW_REACH;
            fb.trace[row * 8 + 4] = 1;
        }
        return;
    }
    const float* R = io.ori + row * io.s_ori + (long long)frame * 54 + 45;
    float vc[3], vi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        vc[c] = fb.x6[rc_pk(row, 171 + 3 * j + c, LD_X6)];               // j3dc (rnn4 output)
        vi[c] = fb.x3[rc_pk(row, 72 + 3 * j + c, LD_X3)];                // j3dr_i (rnn2 output)
    }
    float out[3];
    if (regime == 0) {
        out[0] = vi[0]; out[1] = vi[1]; out[2] = vi[2];
    } else {
        float v[3];                                                        // j3dc.view(23,3).mm(Rcr), L154
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (vc[0] * R[c] + vc[1] * R[3 + c]) + vc[2] * R[6 + c];
        if (regime == 2) {
            out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
        } else {                                                           // lerp with a python-double weight, L163-164
            const double k = fb.kconf[row];
            const float w1 = (float)(1.0 - k), w2 = (float)k;
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = vi[c] * w1 + v[c] * w2;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fb.x78[rc_pk(row, 72 + 3 * j + c, LD_X78)] = out[c];
        fb.xi[rc_pk(row, 3 * j + c, LD_XI)] = out[c];
    }
}

__global__ __launch_bounds__(256) void rc_fuse_kernel(FrameBuffers fb, FrameIO io, rc_params_dev prm, int B) {
    fuse_body(fb, io, prm, B, blockIdx.x * 256 + threadIdx.x);
}

// Wavefront engine: fuse (the frame started 4 ticks ago) and tail (the one started 8 ticks ago) of a tick are independent -- different ring
// slots, disjoint state -- and each of the two launches queued for a CU behind the shared-weight workgroups of the layer steps (fuse 26-46 us
// for 4 us of work, profiles/r06_timeline_final_high.txt). One launch: the tail's workgroups first, the fuse's behind them.
__global__ __launch_bounds__(256) void rc_fuse_tail_kernel(FrameBuffers fb_tail, FrameBuffers fb_fuse, FrameIO io, rc_params_dev prm,
                                                           const BodyConst* __restrict__ body_g, int B, WaveTail wt, int n_tail_blocks) {
    __shared__ WaveScratch s_all[4];
    __shared__ __attribute__((aligned(16))) BodyConst s_body;
    if ((int)blockIdx.x >= n_tail_blocks) {
        fuse_body(fb_fuse, io, prm, B, ((int)blockIdx.x - n_tail_blocks) * 256 + (int)threadIdx.x);
        return;
    }
    tail_impl<4, false>(fb_tail, io, prm, body_g, B, 0, io, 0, wt, s_all, s_body, nullptr);
}

// ================================================================================================== reset
struct ResetArgs {
    float* h[6];
    float* c[6];
    long long h_elems[6];   // elements of h per (layer, copy) = round_up(B, 32) * H   (rc_pk order)
    long long c_elems[6];   // elements of c per layer = B * H                            (row-major)
    int H[6];
};
__global__ __launch_bounds__(256) void rc_reset_kernel(FrameBuffers fb, ResetArgs a, const unsigned char* mask, int B) {
    const int row = blockIdx.x;
    if (mask && !mask[row]) return;
    for (int n = 0; n < 6; ++n) {
        const int H = a.H[n];
        for (int e = threadIdx.x; e < H; e += 256) {
#pragma unroll
            for (int q = 0; q < 2 * RC_HBUF; ++q) a.h[n][q * a.h_elems[n] + rc_pk(row, e, H)] = 0.0f;   // 2 layers x RC_HBUF copies
            a.c[n][(long long)row * H + e] = 0.0f;
            a.c[n][a.c_elems[n] + (long long)row * H + e] = 0.0f;
        }
    }
    if (threadIdx.x == 0) {                                               // net/sig_mp.py:95-104
        fb.has_last[row] = 0;
        fb.n_floor[row] = 0;
        fb.first_reach[row] = 1;
        fb.pend[row] = 0;
    }
}

// ============================================================================================ per-op kernels
__global__ void rc_r6d_kernel(const float* r6d, float* R, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[6], M[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = r6d[6 * i + k];
    r6d_to_R(v, M);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[9 * i + k] = M[k];
}

__global__ void rc_ik_kernel(const BodyConst* __restrict__ body, const float* Rg, float* Rl, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = idx / 24;
    const int j = (int)(idx % 24);
    if (b >= n) return;
    const float* g = Rg + (b * 24 + j) * 9;
    float M[9];
    if (j == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) M[k] = g[k];
    } else {
        mat3T_mul(Rg + (b * 24 + body->parent[j]) * 9, g, M);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) Rl[(b * 24 + j) * 9 + k] = M[k];
}

__global__ __launch_bounds__(64) void rc_fk_bone_kernel(const BodyConst* __restrict__ body, const float* Rg, float* joints, long long n) {
    __shared__ float sR[24][9];
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < 216; e += 64) sR[e / 9][e % 9] = Rg[b * 216 + e];
    __syncthreads();
    if (lane < 24) {
        float o[3];
        bone_chain(body, sR, lane, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) joints[(b * 24 + lane) * 3 + c] = o[c];
    }
}

__global__ __launch_bounds__(64) void rc_body_fk_kernel(const BodyConst* __restrict__ body, const float* pose, const float* tran,
                                                        float* grot, float* joint, float* j33) {
    __shared__ WaveScratch s;
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < 216; e += 64) s.Rl[e / 9][e % 9] = pose[b * 216 + e];
    const float t[3] = {tran[b * 3], tran[b * 3 + 1], tran[b * 3 + 2]};
    __syncthreads();
    wave_body_fk(body, s, t, lane);
    if (grot) for (int e = lane; e < 216; e += 64) grot[b * 216 + e] = s.G[e / 9][e % 9];
    if (lane < 24) {
#pragma unroll
        for (int c = 0; c < 3; ++c) joint[(b * 24 + lane) * 3 + c] = s.P[lane][c] + t[c];
    }
    if (lane < 33) {
#pragma unroll
        for (int c = 0; c < 3; ++c) j33[(b * 33 + lane) * 3 + c] = s.J33[lane][c];
    }
}

// art.math.axis_angle_to_rotation_matrix (articulate/math/angular.py:221-233): c I + (1-c) a a^T + s [a]x, zero -> I
__global__ void rc_aa2R_kernel(const float* aa, float* R, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a[3] = {aa[3 * i], aa[3 * i + 1], aa[3 * i + 2]};
    const float th = norm3(a);
    float k[3] = {a[0] / th, a[1] / th, a[2] / th};
#pragma unroll
    for (int q = 0; q < 3; ++q) if (!(fabsf(k[q]) <= 3.0e38f)) k[q] = 0.0f;          // NaN / inf -> 0
    const float c = cosf(th), sn = sinf(th), t = 1.0f - c;
    const float K[9] = {0.f, -k[2], k[1], k[2], 0.f, -k[0], -k[1], k[0], 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            R[9 * i + 3 * r + q] = ((r == q ? c : 0.0f) + t * (k[r] * k[q])) + sn * K[3 * r + q];
}

// art.math.rotation_matrix_to_axis_angle (angular.py:236-246 loops cv2.Rodrigues on the host). Restated from the
// Rodrigues formula in float64: theta = atan2(|v|, (tr - 1) / 2), v = vee(R - R^T) / 2; near pi the axis comes
// from the symmetric part. PARITY UNPINNED against OpenCV (absent); validated by round trip.
__global__ void rc_R2aa_kernel(const float* Rm, float* aa, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rotmat_to_aa(Rm + 9 * i, aa + 3 * i);
}

// evaluate.py:38-51,70-72: per-frame camera-frame inputs from pixel keypoints and world-frame IMU readings
__global__ void rc_camera_inputs_kernel(const float* kp, const float* acc, const float* ori, CamConst cam, float* j2dc,
                                        float* accc, float* oric, long long n) {
    const long long f = blockIdx.x;
    const int t = threadIdx.x;
    if (t < 33) {
        const float u = kp[(f * 33 + t) * 3], v = kp[(f * 33 + t) * 3 + 1], cf = kp[(f * 33 + t) * 3 + 2];
        float* o = j2dc + (f * 33 + t) * 3;
        o[0] = (cam.Kinv[0] * u + cam.Kinv[1] * v) + cam.Kinv[2];
        o[1] = (cam.Kinv[3] * u + cam.Kinv[4] * v) + cam.Kinv[5];
        o[2] = cf;
    } else if (t < 33 + 18) {
        const int e = t - 33, i = e / 3, r = e % 3;
        const float* a = acc + (f * 6 + i) * 3;
        accc[f * 18 + e] = (cam.R[3 * r] * a[0] + cam.R[3 * r + 1] * a[1]) + cam.R[3 * r + 2] * a[2];
    } else if (t < 33 + 18 + 54) {
        const int e = t - 51, i = e / 9, r = (e % 9) / 3, c = e % 3;
        const float* o = ori + (f * 6 + i) * 9;
        oric[f * 54 + e] = (cam.R[3 * r] * o[c] + cam.R[3 * r + 1] * o[3 + c]) + cam.R[3 * r + 2] * o[6 + c];
    }
}

// The same preparation for EVERY (sequence, camera) row of an evaluation in one launch (evaluate.py:32-51,66-73 loops rows
// and frames on the host): row r reads the pixel keypoints of its camera, the IMU data of its sequence seq_of_row[r] (shared
// by that sequence's cameras) and its own camera constants cams[r]; frames t >= len[r] of the [n_rows, Tmax] layout are
// padding (zero keypoints / accelerations, identity orientations) so that the batched net call sees well-formed rotations.
__global__ __launch_bounds__(128) void rc_camera_inputs_rows_kernel(const float* kp, const float* acc, const float* ori,
                                                                    const int* seq_of_row, const int* len, const CamConst* cams,
                                                                    float sx, float sy, int Tmax, float* j2dc, float* accc, float* oric) {
    const long long r = blockIdx.y, f = blockIdx.x;
    const int t = threadIdx.x;
    const long long o = r * Tmax + f, in = (long long)seq_of_row[r] * Tmax + f;
    const bool pad = f >= len[r];
    const CamConst& cam = cams[r];
    if (t < 33) {
        const float* k = kp + (o * 33 + t) * 3;
        float* d = j2dc + (o * 33 + t) * 3;
        const float u = pad ? 0.f : k[0] * sx, v = pad ? 0.f : k[1] * sy;           // evaluate.py:43-44: pixels from the image-size normalisation
        d[0] = pad ? 0.f : (cam.Kinv[0] * u + cam.Kinv[1] * v) + cam.Kinv[2];
        d[1] = pad ? 0.f : (cam.Kinv[3] * u + cam.Kinv[4] * v) + cam.Kinv[5];
        d[2] = pad ? 0.f : k[2];
    } else if (t < 33 + 18) {
        const int e = t - 33, i = e / 3, q = e % 3;
        const float* a = acc + (in * 6 + i) * 3;
        accc[o * 18 + e] = pad ? 0.f : (cam.R[3 * q] * a[0] + cam.R[3 * q + 1] * a[1]) + cam.R[3 * q + 2] * a[2];
    } else if (t < 33 + 18 + 54) {
        const int e = t - 51, i = e / 9, q = (e % 9) / 3, c = e % 3;
        const float* m = ori + (in * 6 + i) * 9;
        oric[o * 54 + e] = pad ? (q == c ? 1.f : 0.f) : (cam.R[3 * q] * m[c] + cam.R[3 * q + 1] * m[3 + c]) + cam.R[3 * q + 2] * m[6 + c];
    }
}

// smplify forward residual: conf^2 * sum_xy gmof(K (j/z) - kp), sigma^2 d^2 / (sigma^2 + d^2)
// (net/smplify/losses.py:6-12, 36-37, 43-46; ignored landmarks temporal_smplify.py:92,204)
__global__ __launch_bounds__(64) void rc_residual_kernel(const BodyConst* __restrict__ body, const float* pose, const float* tran,
                                                         const float* kp, const float* K, float sigma, unsigned long long ign_mask,
                                                         float* loss) {
    __shared__ WaveScratch s;
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < 216; e += 64) s.Rl[e / 9][e % 9] = pose[b * 216 + e];
    const float t[3] = {tran[b * 3], tran[b * 3 + 1], tran[b * 3 + 2]};
    __syncthreads();
    wave_body_fk(body, s, t, lane);
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

// row-major [B, cols] (leading dimension src_ld) -> rc_pk order with leading dimension ld (rc_lstm_step staging)
__global__ void rc_pack_rows_kernel(const float* src, int src_ld, int cols, float* dst, int ld, int B) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row = idx / cols;
    const int k = (int)(idx % cols);
    if (row >= B) return;
    dst[rc_pk(row, k, ld)] = src[row * src_ld + k];
}

// rc_get_state: turn every pending updater step into a transition step to be run right now
__global__ void rc_flush_flags_kernel(FrameBuffers fb, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    fb.flags2[row] = fb.pend[row] ? RC_ROW2_FLUSH : 0;
    fb.pend[row] = 0;
}

// rc_sequence pre-pass: confidence regime of every (frame, row) -- the SAME arithmetic as rc_prep_kernel (float butterfly
// mean, double compares) so that the host's launch plan and the device's row flags can never disagree.
// codes[t * B + row] = 0 (c <= lo), 1 (lo < c < hi), 2 (c >= hi). Four waves per workgroup, one (row, frame) each.
__global__ __launch_bounds__(256) void rc_scan_conf_kernel(const float* j2d, long long row_stride, int B, int T, double conf_lo,
                                                          double conf_hi, signed char* codes) {
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= (long long)B * T) return;
    const int t = (int)(item / B), row = (int)(item % B);
    const float* kp = j2d + row * row_stride + (long long)t * 99;
    const float cf = lane < 33 ? kp[3 * lane + 2] : 0.f;
    const float c = wave_sum(cf) / 33.0f;
    const double c64 = (double)c;
    if (lane == 0) codes[item] = c64 >= conf_hi ? 2 : (c64 > conf_lo ? 1 : 0);
}

// end of a sequence-mode segment: every sub-net stepped n_frames times on every row (the counters stood still meanwhile)
struct StepPtrs { int* p[6]; };
__global__ void rc_advance_steps_kernel(StepPtrs sp, int n_frames, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * B) return;
    sp.p[i / B][i % B] += n_frames;
}

// ================================================================================================ launchers
// rows per workgroup of the wavefront engine's per-row kernels (RC_SEQ_ROWS_PER_WG = 1: one-wave workgroups, A/B runs)
static int rc_wave_rows_per_wg() {
    static const int v = [] { const char* e = std::getenv("RC_SEQ_ROWS_PER_WG"); return (e && std::atoi(e) == 1) ? 1 : 4; }();
    return v;
}

void rc_launch_scan_conf(const float* j2d, long long row_stride, int B, int T, double conf_lo, double conf_hi, signed char* codes,
                         hipStream_t st) {
    const long long items = (long long)B * T;
    if (items <= 0) return;
    hipLaunchKernelGGL(rc_scan_conf_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, j2d, row_stride, B, T, conf_lo, conf_hi, codes);
}
void rc_launch_advance_steps(int* const* steps6, int n_frames, int B, hipStream_t st) {
    StepPtrs sp;
    for (int i = 0; i < 6; ++i) sp.p[i] = steps6[i];
    hipLaunchKernelGGL(rc_advance_steps_kernel, dim3((6 * B + 255) / 256), dim3(256), 0, st, sp, n_frames, B);
}
void rc_launch_flush_flags(const FrameBuffers& fb, int B, hipStream_t st) {
    hipLaunchKernelGGL(rc_flush_flags_kernel, dim3((B + 255) / 256), dim3(256), 0, st, fb, B);
}
void rc_launch_pack_rows(const float* src, int src_ld, int cols, float* dst, int ld, int B, hipStream_t st) {
    const long long n = (long long)B * cols;
    hipLaunchKernelGGL(rc_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, src_ld, cols, dst, ld, B);
}
void rc_launch_prep(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, int B, int first_frame, hipStream_t st) {
    hipLaunchKernelGGL(rc_prep_kernel, dim3(B), dim3(64), 0, st, fb, io, prm, B, first_frame);
}
void rc_launch_fuse(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, int B, hipStream_t st) {
    hipLaunchKernelGGL(rc_fuse_kernel, dim3((B * 24 + 255) / 256), dim3(256), 0, st, fb, io, prm, B);
}
void rc_launch_tail(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, const BodyConst* body, int B,
                    int first_frame, hipStream_t st, const FrameIO* io_next, const WaveTail* wt, hipEvent_t stop) {
    const WaveTail w = wt ? *wt : WaveTail{};
    const FrameIO nx = io_next ? *io_next : io;
    const int has_next = io_next ? 1 : 0;
    if (w.on && rc_wave_rows_per_wg() == 4) {              // wavefront engine: four rows per workgroup
        const dim3 g((B + 3) / 4), b(256);
        if (stop) hipExtLaunchKernelGGL(rc_tail_kernel<4>, g, b, 0, st, nullptr, stop, 0, fb, io, prm, body, B, first_frame, nx, has_next, w);
        else hipLaunchKernelGGL(rc_tail_kernel<4>, g, b, 0, st, fb, io, prm, body, B, first_frame, nx, has_next, w);
    } else {
        if (stop) hipExtLaunchKernelGGL(rc_tail_kernel<1>, dim3(B), dim3(64), 0, st, nullptr, stop, 0, fb, io, prm, body, B, first_frame, nx, has_next, w);
        else hipLaunchKernelGGL(rc_tail_kernel<1>, dim3(B), dim3(64), 0, st, fb, io, prm, body, B, first_frame, nx, has_next, w);
    }
}
bool rc_launch_fuse_tail(const FrameBuffers& fb_tail, const FrameBuffers& fb_fuse, const FrameIO& io, const rc_params_dev& prm, const BodyConst* body, int B,
                         const WaveTail& wt, hipStream_t st, hipEvent_t stop) {
    if (rc_wave_rows_per_wg() != 4) return false;
    const int n_tail = (B + 3) / 4, n_fuse = (B * 24 + 255) / 256;
    const dim3 g(n_tail + n_fuse), b(256);
    if (stop) hipExtLaunchKernelGGL(rc_fuse_tail_kernel, g, b, 0, st, nullptr, stop, 0, fb_tail, fb_fuse, io, prm, body, B, wt, n_tail);
    else hipLaunchKernelGGL(rc_fuse_tail_kernel, g, b, 0, st, fb_tail, fb_fuse, io, prm, body, B, wt, n_tail);
    return true;
}
void rc_launch_prep_wave(const FrameBuffers& slot, const FrameIO& io0, const rc_params_dev& prm, int B, const WavePrep& w, hipStream_t st) {
    if (rc_wave_rows_per_wg() == 4) hipLaunchKernelGGL(rc_prep_wave_kernel<4>, dim3((B + 3) / 4), dim3(256), 0, st, slot, io0, prm, B, w);
    else hipLaunchKernelGGL(rc_prep_wave_kernel<1>, dim3(B), dim3(64), 0, st, slot, io0, prm, B, w);
}
void rc_launch_reset(const FrameBuffers& fb, float* const* h, float* const* c, const int* hidden, const unsigned char* mask,
                     int B, hipStream_t st) {
    ResetArgs a;
    const long long Bp = (B + 31) / 32 * 32;
    for (int n = 0; n < 6; ++n) {
        a.h[n] = h[n]; a.c[n] = c[n]; a.H[n] = hidden[n];
        a.h_elems[n] = Bp * hidden[n]; a.c_elems[n] = (long long)B * hidden[n];
    }
    hipLaunchKernelGGL(rc_reset_kernel, dim3(B), dim3(256), 0, st, fb, a, mask, B);
}
void rc_launch_r6d(const float* r6d, float* R, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_r6d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r6d, R, n);
}
void rc_launch_camera_inputs(const float* kp, const float* acc, const float* ori, const CamConst& cam, float* j2dc, float* accc,
                             float* oric, long long n, hipStream_t st) {
    hipLaunchKernelGGL(rc_camera_inputs_kernel, dim3((unsigned)n), dim3(128), 0, st, kp, acc, ori, cam, j2dc, accc, oric, n);
}
void rc_launch_camera_inputs_rows(const float* kp, const float* acc, const float* ori, const int* seq_of_row, const int* len,
                                  const CamConst* cams, float sx, float sy, int n_rows, int Tmax, float* j2dc, float* accc,
                                  float* oric, hipStream_t st) {
    if (n_rows <= 0 || Tmax <= 0) return;
    hipLaunchKernelGGL(rc_camera_inputs_rows_kernel, dim3((unsigned)Tmax, (unsigned)n_rows), dim3(128), 0, st, kp, acc, ori, seq_of_row,
                       len, cams, sx, sy, Tmax, j2dc, accc, oric);
}
void rc_launch_aa2R(const float* aa, float* R, long long n, hipStream_t st) {
    hipLaunchKernelGGL(rc_aa2R_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, aa, R, n);
}
void rc_launch_R2aa(const float* R, float* aa, long long n, hipStream_t st) {
    hipLaunchKernelGGL(rc_R2aa_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, R, aa, n);
}
void rc_launch_ik(const BodyConst* body, const float* Rg, float* Rl, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_ik_kernel, dim3((unsigned)((n * 24 + 255) / 256)), dim3(256), 0, st, body, Rg, Rl, n);
}
void rc_launch_fk_bone(const BodyConst* body, const float* Rg, float* joints, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_fk_bone_kernel, dim3((unsigned)n), dim3(64), 0, st, body, Rg, joints, n);
}
void rc_launch_body_fk(const BodyConst* body, const float* pose, const float* tran, float* grot, float* joint, float* j33,
                       long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(rc_body_fk_kernel, dim3((unsigned)n), dim3(64), 0, st, body, pose, tran, grot, joint, j33);
}
void rc_launch_residual(const BodyConst* body, const float* pose, const float* tran, const float* kp, const float* K, float sigma,
                        unsigned long long ign_mask, float* loss, long long T, hipStream_t st) {
    if (T <= 0) return;
    hipLaunchKernelGGL(rc_residual_kernel, dim3((unsigned)T), dim3(64), 0, st, body, pose, tran, kp, K, sigma, ign_mask, loss);
}
