// Per-row device logic of the sig_mp frame shared by rc_frame.hip (frame-stepped / wavefront kernels) and rc_live.hip (the lean
// live frame): the prep of a row (net/sig_mp.py:138-152) and the tail of a row (L173-273). One 64-lane wave per body.
#pragma once
#include "rc_internal.h"
#include "rc_device.h"

// -DRC_LIVE_TRACE (tools/live_trace.py; never in the product library): thread 0 of block 0 stamps points inside the lean live
// frame's kernels with the 100 MHz wall clock -- where a latency-bound kernel spends its microseconds.
#ifdef RC_LIVE_TRACE
__device__ unsigned long long g_live_tt[4][16];
#define RC_LT(k, i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_live_tt[k][i] = wall_clock64(); } while (0)
#else
#define RC_LT(k, i) do { } while (0)
#endif

#define LD_X2 128
#define LD_X3 256
#define LD_X4 256
#define LD_X6 256
#define LD_X78 256
#define LD_XI 128

// =========================================================================================== prep (L138-152)
// One wave, one body. pend_in / uv_count: the row's pending-updater mark and landmark refresh counter as they stand BEFORE this
// frame (read from the state by rc_prep_kernel, handed over in registers when the previous frame's tail runs this in the same
// wave -- rc_tail_kernel with a next frame).
// Split in two so that a caller can request the frame's inputs long before it needs them (rc_tail_kernel asks for the NEXT
// frame's inputs at its very top: every global read of a wave is then one latency, not a chain of them).
struct PrepIn {
    float x, y, cf;         // keypoint `lane` (< 33)
    float a3[3], al;        // accelerations of IMU lane / 3 (lanes < 18) and element `lane`
    float o3[3], ol;        // orientation column entries of IMU lane / 9 (lanes < 54) and element `lane`
    float rv;               // lanes 0..8: root orientation (IMU 5), L139 -- broadcast by prep_compute
};
__device__ __forceinline__ void prep_load(PrepIn& in, const FrameIO& io, const int row, const int lane) {
    const float* kp = io.j2d + row * io.s_j2d;
    const float* acc = io.acc + row * io.s_acc;
    const float* ori = io.ori + row * io.s_ori;
    in.x = 0.f; in.y = 0.f; in.cf = 0.f; in.al = 0.f; in.ol = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) { in.a3[q] = 0.f; in.o3[q] = 0.f; }
    if (lane < 33) { in.x = kp[3 * lane]; in.y = kp[3 * lane + 1]; in.cf = kp[3 * lane + 2]; }
    if (lane < 18) {
        const int i = lane / 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) in.a3[q] = acc[3 * i + q];
        in.al = acc[lane];
    }
    if (lane < 54) {
        const int i = lane / 9, cc = lane % 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) in.o3[q] = ori[9 * i + 3 * q + cc];
        in.ol = ori[lane];
    }
    // nine wave-uniform values: one vector load by lanes 0..8 and lane broadcasts (uniform reads would become scalar loads, each a
    // round trip of its own behind s_waitcnt lgkmcnt(0); vector loads are all in flight together)
    in.rv = lane < 9 ? ori[45 + lane] : 0.f;
}

// What the prep of a row computes (per lane), apart from where it is stored: prep_compute = prep_values + prep_store; the lean
// live frame (rc_live.hip) keeps the values in LDS as the A operand of its first linear1 launch instead.
struct PrepVals {
    float accr, orir;       // lanes < 18 / < 54: root-frame acceleration / orientation entry `lane` (L142-143)
    float xn, yn;           // lanes < 33: bbox-normalised keypoint (L150-152)
    unsigned f, f2;         // RC_ROW_* / RC_ROW2_* bytes of the row (wave-uniform)
    int regime;             // 0 low / 1 mid / 2 high
    double kconf;           // (c - lo) / (hi - lo), L163
};
__device__ __forceinline__ PrepVals prep_values(const PrepIn& in, const rc_params_dev& prm, const int lane, const int first_frame,
                                                const int pend_in, const int uv_count, const unsigned flags2_extra = 0u) {
    PrepVals v;
    const float c = wave_sum(in.cf) / 33.0f;                              // L138
    const double c64 = (double)c;                                         // python-double compares
    const bool gt_lo = c64 > prm.conf_lo, is_hi = c64 >= prm.conf_hi;
    const bool refresh = !prm.live || uv_count == 0;
    bbox_normalise(in.x, in.y, lane, v.xn, v.yn);                         // L150-152
    float Rcr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rcr[k] = lane_bcast(in.rv, k);
    unsigned f = 0;
    const bool vis = gt_lo || first_frame;
    if (vis) f |= RC_ROW_VIS;                                             // L149
    if (gt_lo) f |= RC_ROW_PC;                                            // L161 / L165
    if (!gt_lo && refresh && prm.use_vision_updater) f |= RC_ROW_UPD;     // L264
    // deferred updater steps of the previous frame (see RC_ROW2_*)
    const bool pend = pend_in != 0;
    unsigned f2 = flags2_extra;
    if (pend && vis) f2 |= RC_ROW2_TR;
    if (vis || pend) f2 |= RC_ROW2_M4;
    if (gt_lo || (pend && !vis)) f2 |= RC_ROW2_M6;
    v.f = f; v.f2 = f2;
    v.regime = is_hi ? 2 : (gt_lo ? 1 : 0);
    v.kconf = (c64 - prm.conf_lo) / (prm.conf_hi - prm.conf_lo);         // L163
    v.accr = 0.f; v.orir = 0.f;
    if (lane < 18) {                                                      // accr = accc . Rcr, L142
        const int j = lane % 3;
        v.accr = (in.a3[0] * Rcr[j] + in.a3[1] * Rcr[3 + j]) + in.a3[2] * Rcr[6 + j];
    }
    if (lane < 54) {                                                      // orir = Rcr^T . oric, L143
        const int r = (lane % 9) / 3;
        v.orir = (Rcr[r] * in.o3[0] + Rcr[3 + r] * in.o3[1]) + Rcr[6 + r] * in.o3[2];
    }
    return v;
}
__device__ __forceinline__ void prep_store(const FrameBuffers& fb, const PrepIn& in, const PrepVals& v, const int row, const int lane) {
    if (lane == 0) {
        fb.flags[row] = (unsigned char)v.f;
        fb.flags2[row] = (unsigned char)v.f2;
        fb.regime[row] = (unsigned char)v.regime;
        fb.kconf[row] = v.kconf;
        int* tr = fb.trace + row * 8;
        tr[0] = v.regime;
        tr[1] = 0; tr[2] = 0; tr[4] = 0; tr[5] = 0; tr[6] = 0; tr[7] = 0;
    }
    if (lane < 18) {
        fb.x2[rc_pk(row, lane, LD_X2)] = v.accr;
        fb.x3[rc_pk(row, lane, LD_X3)] = v.accr;
        fb.x78[rc_pk(row, lane, LD_X78)] = v.accr;
        fb.x4[rc_pk(row, lane, LD_X4)] = in.al;
        fb.x6[rc_pk(row, lane, LD_X6)] = in.al;
    }
    if (lane < 54) {
        fb.x2[rc_pk(row, 18 + lane, LD_X2)] = v.orir;
        fb.x3[rc_pk(row, 18 + lane, LD_X3)] = v.orir;
        fb.x78[rc_pk(row, 18 + lane, LD_X78)] = v.orir;
        fb.x4[rc_pk(row, 18 + lane, LD_X4)] = in.ol;
        fb.x6[rc_pk(row, 18 + lane, LD_X6)] = in.ol;
    }
    if (lane < 33) {
        const int k = 72 + 3 * lane;
        fb.x4[rc_pk(row, k, LD_X4)] = v.xn; fb.x4[rc_pk(row, k + 1, LD_X4)] = v.yn; fb.x4[rc_pk(row, k + 2, LD_X4)] = in.cf;
        fb.x6[rc_pk(row, k, LD_X6)] = in.x; fb.x6[rc_pk(row, k + 1, LD_X6)] = in.y; fb.x6[rc_pk(row, k + 2, LD_X6)] = in.cf;
    }
}
// Returns the RC_ROW_* byte of the row (the same value on every lane); flags2_extra is OR-ed into the second flag byte.
__device__ __forceinline__ unsigned prep_compute(const FrameBuffers& fb, const PrepIn& in, const rc_params_dev& prm, const int row,
                                                 const int lane, const int first_frame, const int pend_in, const int uv_count,
                                                 const unsigned flags2_extra = 0u) {
    const PrepVals v = prep_values(in, prm, lane, first_frame, pend_in, uv_count, flags2_extra);
    prep_store(fb, in, v, row, lane);
    return v.f;
}
__device__ __forceinline__ unsigned prep_body(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, const int row,
                                              const int lane, const int first_frame, const int pend_in, const int uv_count,
                                              const unsigned flags2_extra = 0u) {
    PrepIn in;
    prep_load(in, io, row, lane);
    return prep_compute(fb, in, prm, row, lane, first_frame, pend_in, uv_count, flags2_extra);
}


// ============================================================================================ tail (L173-273)
// has_next: the same wave goes on with the prep of the NEXT frame of its row (io_next) -- in a frame-stepped sequence the two
// kernels are back to back on the stream anyway, and the row's state they share travels in registers.
// wt.on: ring slot of the per-row-cursor engine -- the row's frame index comes from the slot (bubbles exit), and the vision
// updater's inputs go to the slot that starts at this tick (see WaveTail).
// RPB rows per workgroup, a wave each. 1: the frame-stepped launches (256 workgroups spread over the chip: shortest latency).
// 4: the wavefront engine, where the kernel runs beside the wide GEMM tiles (see rc_prep_wave_kernel); the body constants are
// staged once per workgroup behind its only block-wide barrier, after which the rows are independent (own wave, own scratch,
// wave-local synchronisation) and a bubble row's wave simply leaves.
// LIVE (rc_live.hip: the last kernel of the lean live frame): a 256-thread workgroup per row whose four waves have just summed the
// second stage's linear2 partial sums into `sub` (LDS); all of them stage the body constants, then wave 0 runs the row.
struct LiveSub { float r6d[144]; float pc[4]; float vr[4]; float ct[4]; };
// The per-row words of the tail, GATHERED: lane l loads word l of this list with ONE vector load and the values are handed out with
// readlane -- as wave-uniform reads they would be scalar loads, each waiting for the previous one (s_waitcnt lgkmcnt(0)).
struct TailRegs { float gv; unsigned bv; float acc_l, ori_l; };
template <bool LIVE>
__device__ __forceinline__ void tail_request(TailRegs& t, const FrameBuffers& fb, const FrameIO& io, const int row, const int lane) {
    const float* ori = io.ori + row * io.s_ori;
    const float* acc = io.acc + row * io.s_acc;
    const bool ft_given = io.first_tran != nullptr;
    // LIVE: the frame's inputs sit in pinned HOST memory (a PCIe round trip per read); the first kernel of the frame has left the IMU
    // data in the prefix of the rnn6 input row (prep_store: x6[0..17] = acc, x6[18..71] = ori) -- read that copy instead
    const float* x6 = fb.x6;
    const float* gp = nullptr;
    if (lane < 6) gp = fb.last_pfoot + row * 6 + lane;
    else if (lane < 9) gp = fb.last_tran + row * 3 + (lane - 6);
    else if (lane < 12) gp = fb.gravity + row * 3 + (lane - 9);
    else if (lane < 15) gp = LIVE ? nullptr : fb.pc + row * 4 + (lane - 12);       // (LIVE: the sub-net outputs come from LDS)
    else if (lane < 18) gp = LIVE ? nullptr : fb.vr + row * 4 + (lane - 15);
    else if (lane < 20) gp = LIVE ? nullptr : fb.contact + row * 2 + (lane - 18);
    else if (lane < 38) gp = fb.floor + row * 33 + 15 + (lane - 20);              // floor samples 5..10 (L213: mean of the last six)
    else if (lane < 47) gp = LIVE ? x6 + rc_pk(row, 18 + 45 + (lane - 38), LD_X6) : ori + 45 + (lane - 38);   // Rcr, L139
    else if (lane < 50) gp = ft_given ? io.first_tran + row * 3 + (lane - 47) : nullptr;
    else if (lane < 52) gp = reinterpret_cast<const float*>(fb.kconf + row) + (lane - 50);
    else if (lane == 52) gp = reinterpret_cast<const float*>(fb.has_last + row);
    else if (lane == 53) gp = reinterpret_cast<const float*>(fb.n_floor + row);
    else if (lane == 54) gp = reinterpret_cast<const float*>(fb.uv_count + row);
    t.gv = gp ? *gp : 0.f;
    const unsigned char* bp = lane == 0 ? fb.flags + row : (lane == 1 ? fb.regime + row : nullptr);
    t.bv = bp ? (unsigned)*bp : 0u;
    t.acc_l = lane < 18 ? (LIVE ? x6[rc_pk(row, lane, LD_X6)] : acc[lane]) : 0.f;  // this frame's IMU data (updater inputs)
    t.ori_l = lane < 54 ? (LIVE ? x6[rc_pk(row, 18 + lane, LD_X6)] : ori[lane]) : 0.f;
}
// LIVE: the caller (rc_live_tail_kernel) has staged the body constants, requested the row's words (`pre`) and summed the sub-net
// outputs into `sub` (LDS) behind ONE batch of loads; wave 0 runs the row with wave-local synchronisation.
template <int RPB, bool LIVE>
__device__ __forceinline__ void tail_impl(FrameBuffers fb, FrameIO io, const rc_params_dev& prm, const BodyConst* __restrict__ body_g,
                                          const int B, const int first_frame, const FrameIO& io_next, const int has_next, WaveTail wt,
                                          WaveScratch* s_all, BodyConst& s_body, const LiveSub* sub, const TailRegs* pre = nullptr) {
    constexpr bool WL = RPB > 1 || LIVE;
    const int row = LIVE ? (int)blockIdx.x : (int)(blockIdx.x * RPB + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    WaveScratch& s = s_all[LIVE ? 0 : (threadIdx.x >> 6)];
    if constexpr (WL && !LIVE) {
        BodyStage<64 * RPB> bsa;
        bsa.load(body_g, threadIdx.x);
        bsa.store(&s_body, threadIdx.x);
        __syncthreads();
        if (row >= B) return;
    }
    int frame = 0;
    if (wt.on) {
        frame = fb.frame[row];
        if (frame < 0) return;                                             // bubble: the whole wave leaves
        io.j2d += (long long)frame * 99; io.acc += (long long)frame * 18; io.ori += (long long)frame * 54;
        io.pose_out += (long long)frame * 216; io.tran_out += (long long)frame * 3;
    }
    const bool wave_ride = wt.on && frame != wt.t_last;                    // updater inputs ride the target slot
    // ---- every global read of this wave, requested up front (one memory latency instead of a chain of ~10: this kernel is a
    // dependent-latency chain, 1,560 B of I/O per body). The state reads are safe to hoist: only this wave writes its row.
    BodyStage<64> bst;
    if constexpr (!WL) bst.load(body_g, lane);
    const bool ft_given = io.first_tran != nullptr;
    TailRegs tr_;
    if constexpr (LIVE) tr_ = *pre; else tail_request<false>(tr_, fb, io, row, lane);
    const float gv = tr_.gv;
    const unsigned bv = tr_.bv;
    float r6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (lane < 24) {
#pragma unroll
        for (int k = 0; k < 6; ++k) r6[k] = LIVE ? sub->r6d[6 * lane + k] : fb.r6d[row * 144 + 6 * lane + k];
    }
    const float acc_l = tr_.acc_l, ori_l = tr_.ori_l;
    PrepIn nin;
    if (has_next) prep_load(nin, io_next, row, lane);
    if constexpr (!WL) bst.store(&s_body, lane);                          // (LDS: visible to the wave after the first barrier)
    const BodyConst* body = &s_body;
    float lpf[6], ltr[3], g[3], pc[3], vr[3], flr[6][3], Rcr[9], ftr[3];
#pragma unroll
    for (int c = 0; c < 6; ++c) lpf[c] = lane_bcast(gv, c);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ltr[c] = lane_bcast(gv, 6 + c); g[c] = lane_bcast(gv, 9 + c); pc[c] = lane_bcast(gv, 12 + c); vr[c] = lane_bcast(gv, 15 + c);
        ftr[c] = lane_bcast(gv, 47 + c);
    }
    float ct0 = lane_bcast(gv, 18), ct1 = lane_bcast(gv, 19);
    if constexpr (LIVE) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { pc[c] = sub->pc[c]; vr[c] = sub->vr[c]; }
        ct0 = sub->ct[0]; ct1 = sub->ct[1];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) flr[q][c] = lane_bcast(gv, 20 + 3 * q + c);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rcr[k] = lane_bcast(gv, 38 + k);
    const double k64 = __hiloint2double(__float_as_int(lane_bcast(gv, 51)), __float_as_int(lane_bcast(gv, 50)));
    const bool has_last = __float_as_int(lane_bcast(gv, 52)) != 0;
    int n_floor = __float_as_int(lane_bcast(gv, 53));
    const int uvc = __float_as_int(lane_bcast(gv, 54));
    const unsigned flags = (unsigned)__builtin_amdgcn_readlane((int)bv, 0);
    const int regime = __builtin_amdgcn_readlane((int)bv, 1);
    if constexpr (LIVE) RC_LT(2, 3);

    // L173: 6D -> global rotations (root-relative frame)
    if (lane < 24) {
        float R[9];
        r6d_to_R(r6, R);
#pragma unroll
        for (int k = 0; k < 9; ++k) s.Rg[lane][k] = R[k];
    }
    rc_sync<WL>();
    if constexpr (LIVE) RC_LT(2, 4);
    // L174-175: local rotations, root replaced by the pelvis IMU orientation
    if (lane < 24) {
        float R[9];
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = Rcr[k];
        } else {
            mat3T_mul(s.Rg[body->parent[lane]], s.Rg[lane], R);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s.Rl[lane][k] = R[k];                   // (written out below, 256 contiguous bytes per store)
    }
    // L186: feet from the predicted GLOBAL rotations, rotated to the camera frame. Even lanes walk the chain of joint 10, odd lanes
    // that of joint 11 (two dependent chains of LDS reads side by side instead of one after the other), then both are broadcast.
    float pf[2][3];
    {
        float jf[3], pl[3];
        bone_chain(body, s.Rg, 10 + (lane & 1), jf);
#pragma unroll
        for (int c = 0; c < 3; ++c) pl[c] = (jf[0] * Rcr[3 * c] + jf[1] * Rcr[3 * c + 1]) + jf[2] * Rcr[3 * c + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pf[0][c] = lane_bcast(pl[c], 0); pf[1][c] = lane_bcast(pl[c], 1); }
    }
    if constexpr (LIVE) RC_LT(2, 5);

    // L187-203: root translation
    const float c0 = sigmoidf_(ct0), c1 = sigmoidf_(ct1);                  // L170
    const float cmax = fmaxf(c0, c1);
    const int foot = c1 > c0 ? 1 : 0;
    const bool use_vel = (cmax < prm.contact_threshold) || !has_last;
    float tran[3];
    {
        float v[3];
        if (use_vel) {
            mat3_vec(Rcr, vr, v);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = v[c] * 3.0f / 60.0f;        // vel_scale / 60, L188
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (foot ? lpf[3 + c] : lpf[c]) - pf[foot][c];   // L190
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) tran[c] = has_last ? ltr[c] + v[c] : v[c];
    }
    bool far = false;
    if (regime == 2) {                                                     // L196-203
        const double kf = k64 > 1.0 ? 1.0 : k64;
        const float d[3] = {pc[0] - tran[0], pc[1] - tran[1], pc[2] - tran[2]};
        far = norm3(d) > prm.distance_threshold || prm.tran_filter_num > 1.0;
        if (far) {
            tran[0] = pc[0]; tran[1] = pc[1]; tran[2] = pc[2];
        } else {
            const double w = prm.tran_filter_num * kf;
            const float w1 = (float)(1.0 - w), w2 = (float)w;
#pragma unroll
            for (int c = 0; c < 3; ++c) tran[c] = tran[c] * w1 + pc[c] * w2;
        }
    }
    // L206-221: floor height along gravity
    const bool on_ground = cmax > prm.contact_threshold;
    float p0[3], p1[3], pick[3] = {0.f, 0.f, 0.f};
    int appended = -1;
    {
        const float d0 = ((pf[0][0] + tran[0]) * g[0] + (pf[0][1] + tran[1]) * g[1]) + (pf[0][2] + tran[2]) * g[2];
        const float d1 = ((pf[1][0] + tran[0]) * g[0] + (pf[1][1] + tran[1]) * g[1]) + (pf[1][2] + tran[2]) * g[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) { p0[c] = d0 * g[c]; p1[c] = d1 * g[c]; }
    }
    const bool p0_lt_p1 = norm3(p0) < norm3(p1);
    if (n_floor < 11 && !first_frame && !ft_given && on_ground && prm.use_flat_floor && regime == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) pick[c] = p0_lt_p1 ? p1[c] : p0[c];
        appended = n_floor;
        n_floor += 1;
    }
    if (prm.use_flat_floor && n_floor > 10 && on_ground) {
        float m[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 5; q < 11; ++q) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e = (q == appended) ? pick[c] : flr[q - 5][c];
                m[c] = (q == 5) ? e : m[c] + e;
            }
        }
        float d0[3], d1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { m[c] = m[c] / 6.0f; d1[c] = m[c] - p1[c]; d0[c] = m[c] - p0[c]; }
        if (p0_lt_p1 && norm3(d1) < prm.height_threshold) {
#pragma unroll
            for (int c = 0; c < 3; ++c) tran[c] += d1[c];
        } else if (norm3(d0) < prm.height_threshold) {
#pragma unroll
            for (int c = 0; c < 3; ++c) tran[c] += d0[c];
        }
    }
    if (ft_given) {                                                        // L222-225
#pragma unroll
        for (int c = 0; c < 3; ++c) tran[c] = ftr[c];
    } else if (first_frame) {
        tran[0] = pc[0]; tran[1] = pc[1]; tran[2] = pc[2];
    }
    const bool live = prm.live != 0;
    const bool refresh = !live || uvc == 0;
    const int uvc_next = (live && (prm.use_reproj_opt || prm.use_vision_updater)) ? (refresh ? prm.update_vision_freq : uvc - 1) : uvc;
    const int pend_next = ((flags & RC_ROW_UPD) && !wave_ride) ? 1 : 0;
    if constexpr (LIVE) RC_LT(2, 6);
    rc_sync<WL>();   // all lanes have read the per-row state; lane 0 may now overwrite it
    {
        float* po = io.pose_out + row * io.s_pose;                         // L174-175 -> output (s.Rl is complete behind the barrier above)
        const float* rl = &s.Rl[0][0];
#pragma unroll
        for (int e = lane; e < 216; e += 64) po[e] = rl[e];
    }
    if constexpr (LIVE) RC_LT(2, 7);
    if (lane == 0) {                                                       // L227, L273
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            fb.last_pfoot[row * 6 + c] = pf[0][c];
            fb.last_pfoot[row * 6 + 3 + c] = pf[1][c];
        }
        fb.has_last[row] = 1;
        fb.n_floor[row] = n_floor;
        if (appended >= 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) fb.floor[row * 33 + 3 * appended + c] = pick[c];
        }
        // L228, L234-242: the refresh counter only moves while one of its two consumers is switched on
        if (live && (prm.use_reproj_opt || prm.use_vision_updater)) fb.uv_count[row] = uvc_next;
        fb.pend[row] = pend_next;                                         // L264-271 run at the start of the next frame
        if ((flags & RC_ROW_UPD) && wave_ride) {                          // the two updater steps join the target slot's launches
            wt.flags2[row] |= (unsigned char)(RC_ROW2_M4 | RC_ROW2_M6);
            wt.wsteps[2 * B + row] = ++wt.steps4[row];
            wt.wsteps[3 * B + row] = ++wt.steps6[row];
        }
        int* tr = fb.trace + row * 8;
        tr[0] = regime;                                                   // (also written by prep / fuse; with several frames of
        tr[4] = (flags & RC_ROW_REACH) ? 1 : 0;                           //  a row in flight the LAST tail must own every field)
        tr[1] = ((flags & RC_ROW_VIS) ? 1 : 0) + ((flags & RC_ROW_UPD) ? 1 : 0);
        tr[2] = (first_frame ? 1 : 0) + ((flags & RC_ROW_PC) ? 1 : 0) + ((flags & RC_ROW_UPD) ? 1 : 0);
        tr[3] = n_floor;
        tr[5] = use_vel ? 1 : 0;
        tr[6] = foot;
        tr[7] = (far && regime == 2) ? 1 : 0;
    }

    // L228-242: mesh landmarks from the LOCAL pose chained from the camera-frame root. The reference skins the mesh on every frame;
    // its landmarks are READ only by the vision updater (L264-271), the optional re-projection refinement (L245-261) and the live
    // mode's landmark cache (L234-242) -- on every other frame (a visible row, the common case) the chain and the skinning are
    // skipped: the frame's outputs do not depend on them. (Wave-uniform: flags, regime and the counters are per row.)
    const bool need_mesh = (flags & RC_ROW_UPD) || (prm.use_reproj_opt && regime >= 1) ||
                           (live && refresh && (prm.use_reproj_opt || prm.use_vision_updater));
    if (need_mesh) wave_body_fk<WL>(body, s, tran, lane);
    if constexpr (LIVE) RC_LT(2, 8);
    if (need_mesh && live && (prm.use_reproj_opt || prm.use_vision_updater)) {
        if (lane < 33) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (refresh) fb.j_temp[row * 99 + 3 * lane + c] = s.J33[lane][c];
                else s.J33[lane][c] = fb.j_temp[row * 99 + 3 * lane + c];
            }
        }
        rc_sync<WL>();
    }

    // L245-261 (use_reproj_opt, default off): closed-form refinement of the translation from the 2D residual
    if (prm.use_reproj_opt && regime >= 1) {
        const float* kp = io.j2d + row * io.s_j2d;
        float p = 0.f, u = 0.f, v = 0.f, jx = 0.f, jy = 0.f, jz = 1.f;
        if (lane < 33) { u = kp[3 * lane]; v = kp[3 * lane + 1]; p = kp[3 * lane + 2]; jx = s.J33[lane][0]; jy = s.J33[lane][1]; jz = s.J33[lane][2]; }
        const bool on = lane < 33;
        const float ax = wave_sum(on ? p / (jz * jz) : 0.f) + prm.smooth;
        const float bx = wave_sum(on ? p * (-jx / (jz * jz) + u / jz) : 0.f);
        const float by = wave_sum(on ? p * (-jy / (jz * jz) + v / jz) : 0.f);
        const float dx = bx / ax, dy = by / ax;
        jx += dx; jy += dy;
        const float z2 = jz * jz;
        const float az = wave_sum(on ? p * (jx * jx + jy * jy) / (z2 * z2) : 0.f) + prm.smooth;
        const float bz = wave_sum(on ? p * ((jx / jz - u) * jx / z2 + (jy / jz - v) * jy / z2) : 0.f);
        const float dz = bz / az;
        tran[0] = (tran[0] + dx) + 0.0f; tran[1] = (tran[1] + dy) + 0.0f; tran[2] = (tran[2] + 0.0f) + dz;
        if (on) { s.J33[lane][0] = jx; s.J33[lane][1] = jy; s.J33[lane][2] = (jz + 0.0f) + dz; }
        rc_sync<WL>();
    }
    if (lane == 0) {                                                       // L273 (after the optional refinement)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            fb.last_tran[row * 3 + c] = tran[c];
            io.tran_out[row * io.s_tran + c] = tran[c];
        }
    }

    // L264-271: inputs of the vision updater (rnn6 on raw re-projection, rnn4 on the normalised one)
    if (flags & RC_ROW_UPD) {
        // the updater's sub-net steps run at the start of the NEXT frame: this frame's IMU data goes with them
        float* const x4l = wt.on ? (wave_ride ? wt.x4l : wt.cx4l) : fb.x4l;
        float* const x6l = wt.on ? (wave_ride ? wt.x6l : wt.cx6l) : fb.x6l;
        if (lane < 18) { x4l[rc_pk(row, lane, LD_X4)] = acc_l; x6l[rc_pk(row, lane, LD_X6)] = acc_l; }
        if (lane < 54) { x4l[rc_pk(row, 18 + lane, LD_X4)] = ori_l; x6l[rc_pk(row, 18 + lane, LD_X6)] = ori_l; }
        float x = 0.f, y = 0.f, z1 = 0.f;
        if (lane < 33) {
            const float z = s.J33[lane][2];
            x = s.J33[lane][0] / z; y = s.J33[lane][1] / z; z1 = z / z;    // L265
            const int k = 72 + 3 * lane;
            x6l[rc_pk(row, k, LD_X6)] = x; x6l[rc_pk(row, k + 1, LD_X6)] = y; x6l[rc_pk(row, k + 2, LD_X6)] = z1;
        }
        if (lane >= 1 && lane < 24) {                                     // L266: joint[1:] - joint[:1]
#pragma unroll
            for (int c = 0; c < 3; ++c)
                x6l[rc_pk(row, 171 + 3 * (lane - 1) + c, LD_X6)] = (s.P[lane][c] + tran[c]) - (s.P[0][c] + tran[c]);
        }
        float xn, yn;
        bbox_normalise(x, y, lane, xn, yn);                                // L268-270
        if (lane < 33) {
            const int k = 72 + 3 * lane;
            x4l[rc_pk(row, k, LD_X4)] = xn; x4l[rc_pk(row, k + 1, LD_X4)] = yn; x4l[rc_pk(row, k + 2, LD_X4)] = z1;
        }
    }
    // L181-183: rnn2 state <- init_net(j3dr); takes effect from the next frame
    if (flags & RC_ROW_REACH) {
        const int cur = (wt.on ? fb.wsteps[row] : fb.steps2[row]) % RC_HBUF;   // copy the rnn2 step of this frame wrote
        const float* src = fb.init_out + row * 2048;
        for (int e = lane; e < 512; e += 64) {
            fb.h2[cur * fb.h2_par_stride + rc_pk(row, e, 512)] = src[e];
            fb.h2[fb.h2_layer_stride + cur * fb.h2_par_stride + rc_pk(row, e, 512)] = src[512 + e];
            fb.c2[row * 512 + e] = src[1024 + e];
            fb.c2[fb.c2_layer_stride + row * 512 + e] = src[1536 + e];
        }
    }
    if (has_next) prep_compute(fb, nin, prm, row, lane, 0, pend_next, uvc_next);
    if constexpr (LIVE) RC_LT(2, 9);
}

template <int RPB>
__global__ __launch_bounds__(64 * RPB) void rc_tail_kernel(FrameBuffers fb, FrameIO io, rc_params_dev prm,
                                                           const BodyConst* __restrict__ body_g, int B, int first_frame, FrameIO io_next,
                                                           int has_next, WaveTail wt) {
    __shared__ WaveScratch s_all[RPB];
    __shared__ __attribute__((aligned(16))) BodyConst s_body;
    tail_impl<RPB, false>(fb, io, prm, body_g, B, first_frame, io_next, has_next, wt, s_all, s_body, nullptr);
}

