/*
 * robustcap_hip.h -- C ABI of the MI355X (gfx950) implementation of RobustCap's sig_mp per-frame path.
 *
 * The reference (shaohua-pan/RobustCap) is pure Python/PyTorch and has no FFI of its own; every entry point
 * below names the reference Python interface it replaces (paths relative to the reference repo root). The
 * Python host (robustcap_amd/) binds these with ctypes; see INTEGRATION.md for the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - All tensors are float32, row-major, resident in device memory unless a parameter says "host".
 *   - The caller owns every I/O buffer and passes raw device pointers plus the HIP stream to enqueue on
 *     (hipStream_t is passed as void*). The library owns only its context (packed weights, recurrent state,
 *     scratch). No call synchronises the device except rc_create / rc_finalize_weights / rc_set_body / rc_set_mesh /
 *     rc_shape_body / rc_get_state / rc_get_trace / rc_destroy; rc_sequence synchronises `stream` ONCE per call when its
 *     launch planner is on AND the call has at least min_frames frames (rc_set_sequence_mode; shorter calls and mode 0 are
 *     fully asynchronous), rc_camera_inputs_rows once (constant upload), the
 *     *_host-mean variants of the metric calls and rc_smplify_* as documented with them.
 *   - Every function returns 0 on success or a negative rc_status; rc_last_error() gives a message. Nothing
 *     throws across the boundary. A context is not re-entrant; distinct contexts on distinct streams are
 *     independent.
 *   - "row" = one body (one sequence/camera) of the batch; rows never interact.
 */
#ifndef ROBUSTCAP_HIP_H
#define ROBUSTCAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rc_ctx rc_ctx;

typedef enum {
    RC_OK = 0,
    RC_ERR_INVALID = -1,    /* bad argument                                  */
    RC_ERR_HIP = -2,        /* a HIP runtime call failed                     */
    RC_ERR_STATE = -3,      /* call order (weights / body not loaded yet)    */
    RC_ERR_UNKNOWN_KEY = -4 /* rc_load_weight: not a sig_mp state_dict key   */
} rc_status;

/* Behaviour switches = the class attributes callers poke on the reference Net (net/sig_mp.py:27-45, 91-93). */
typedef struct {
    double conf_lo, conf_hi;       /* Net.conf_range: (0.7, 0.8); (0.85, 0.9) when constructed live      */
    float contact_threshold;       /* 0.7                                                                */
    float distance_threshold;      /* Net.distrance_threshold = 10                                       */
    float height_threshold;        /* Net.height_threhold = 0.15                                         */
    double tran_filter_num;        /* 0.05; 0.01 when constructed live                                   */
    int32_t use_flat_floor;        /* 1                                                                  */
    int32_t use_vision_updater;    /* 1                                                                  */
    int32_t use_imu_updater;       /* 1                                                                  */
    int32_t live;                  /* Net.live: landmark refresh every update_vision_freq+1 frames       */
    int32_t update_vision_freq;    /* 30                                                                 */
    int32_t use_reproj_opt;        /* 0; Net.use_reproj_opt: closed-form translation refinement L245-261  */
    float smooth;                  /* Net.smooth = 1 (regulariser of that refinement)                    */
    int32_t reserved;
} rc_params;

/* rc_step flags */
#define RC_FLAG_FIRST_FRAME 1u /* forward_online(first_frame=True), net/sig_mp.py:114,149,155,208,224 */

/* ---- lifecycle ------------------------------------------------------------------------------------------ */
/* Replaces Net() / Net().to(device) (net/sig_mp.py:47-93). batch = number of independent rows (>= 1).
 * live_at_construction mirrors "Net.live = True" BEFORE construction (evaluate.py:392). */
int rc_create(int32_t batch, int32_t live_at_construction, rc_ctx** out);
int rc_destroy(rc_ctx* ctx);
const char* rc_last_error(const rc_ctx* ctx); /* ctx may be NULL: last error of a failed rc_create */
int rc_default_params(int32_t live_at_construction, rc_params* out);
int rc_get_params(const rc_ctx* ctx, rc_params* out);
int rc_set_params(rc_ctx* ctx, const rc_params* p);

/* Replaces Net.load_state_dict (evaluate.py:58): one call per state_dict entry, key names and shapes as in
 * the reference ("rnn2.rnn.weight_ih_l0", "rnn4.linear1.bias", "rnn2.init_net.4.weight", ...).
 * `host` points to numel float32 values in HOST memory (torch layout). rc_finalize_weights repacks everything
 * to the kernel layout and uploads; it fails if a key is missing. The library keeps NO host copy afterwards (254 MB per
 * context otherwise): a later reload -- also of a single tensor -- passes every tensor again before the next finalize. */
int rc_load_weight(rc_ctx* ctx, const char* key, const float* host, int64_t numel);
int rc_finalize_weights(rc_ctx* ctx);

/* Replaces art.ParametricModel(paths.smpl_file) as consumed by the path (articulate/model.py:29-40,78-93 and
 * config.mp_mask): parent[24] (parent[0] ignored), rest joints J[24,3], and for the 33 landmark vertices the
 * skinning weights w33[33,24] and template positions v33[33,3]. HOST pointers. */
int rc_set_body(rc_ctx* ctx, const int32_t* parent, const float* J, const float* w33, const float* v33);

/* Replaces "net.gravityc = ..." (evaluate.py:73): per-row gravity direction in the camera frame, HOST [batch,3]. */
int rc_set_gravity(rc_ctx* ctx, const float* gravity_host);

/* Replaces Net.reset_states (net/sig_mp.py:95-104). row_mask: DEVICE uint8[batch] (non-zero = reset) or NULL
 * for all rows. Like the reference it does not touch update_vision_count / j_temp / gravity. */
int rc_reset(rc_ctx* ctx, const uint8_t* row_mask, void* stream);

/* ---- the hot path --------------------------------------------------------------------------------------- */
/* Replaces Net.forward_online (net/sig_mp.py:113-274) for `batch` rows at once; row b == the reference run
 * alone on sequence b. j2dc[batch,33,3] (x/z, y/z, conf), accc[batch,6,3], oric[batch,6,3,3]; first_tran
 * [batch,3] or NULL; pose_out[batch,24,3,3] (local rotations, root = camera-frame IMU), tran_out[batch,3].
 * Enqueues ~14 kernels on `stream`; no host synchronisation, all branching is device side. */
int rc_step(rc_ctx* ctx, const float* j2dc, const float* accc, const float* oric, const float* first_tran,
            uint32_t flags, float* pose_out, float* tran_out, void* stream);

/* The evaluate.py per-sequence loop (evaluate.py:75-83) for all rows: frames t = 0..T-1, inputs and outputs
 * indexed [row][t] with element strides  row_stride_* (between rows) and T contiguous frames per row:
 *   j2dc + row*rs_j2d + t*99, accc + row*rs_acc + t*18, oric + row*rs_ori + t*54,
 *   pose_out + row*rs_pose + t*216, tran_out + row*rs_tran + t*3.
 * Frame 0 uses first_tran (if not NULL) and `flags`; later frames use neither. See rc_set_sequence_mode for how the
 * frames are scheduled (and for the one synchronisation of `stream` the default mode costs per call). */
int rc_sequence(rc_ctx* ctx, int32_t T, const float* j2dc, int64_t rs_j2d, const float* accc, int64_t rs_acc,
                const float* oric, int64_t rs_ori, const float* first_tran, uint32_t flags, float* pose_out,
                int64_t rs_pose, float* tran_out, int64_t rs_tran, void* stream);

/* Arithmetic of the GEMM products of EVERY launch of a context (linear layers, LSTM gate GEMMs, init_net):
 *   mode 0: v_mfma_f32_16x16x4_f32 -- fp32 operands, bitwise an fma chain per output element;
 *   mode 1: split-bf16 products on v_mfma_f32_16x16x32_bf16 -- every fp32 operand is the exact sum of three bf16 numbers
 *           (weights split once at rc_finalize_weights, activations on the fly) and every product the fp32-accumulated sum
 *           of the partial products down to 2^-16 of it (what is dropped is below 2^-23 of a product, i.e. below the
 *           rounding of the running fp32 sum); operands, accumulators, gates and state stay fp32. Same accuracy class as
 *           mode 0 (parity with the reference: tests), 2.7x fewer MFMA cycles per product.
 * Default: mode 1 for contexts of batch >= 48, mode 0 below (weight-streaming bound: 6 B instead of 4 B per weight would
 * slow them). Within one mode a row's result does not depend on the batch, the tile shape or the engine
 * (bitwise); between the two modes results differ by fp32 rounding noise. rc_get_gemm_mode returns the mode. */
int rc_set_gemm_mode(rc_ctx* ctx, int32_t mode);
int rc_get_gemm_mode(const rc_ctx* ctx);
int rc_default_gemm_mode(int32_t total_rows);   /* the default of a context of that many rows: sharded runs pin every shard to the TOTAL's */

/* Sequence mode of rc_sequence (on by default). With mode >= 1 every rc_sequence call of T >= 2 frames first classifies
 * each (frame, row) on the device (the arithmetic of the per-frame prep kernel), reads the codes back -- ONE
 * synchronisation of `stream` per call -- and plans the launches on the host:
 *   - the PER-ROW-CURSOR WAVEFRONT engine runs every frame but one that takes first_frame / first_tran: the stages of a
 *     frame (prep | linear1 | LSTM l0 | l1 | linear2 of {rnn2, rnn4} + fuse | the same four of {rnn6, rnn3, rnn7, rnn8} +
 *     tail) are skewed over consecutive ticks and a 16-slot ring, two merged gate-GEMM launches per tick carry the stages of
 *     up to eight frames -- one on `stream`, the other (from 48 rows) on a context-owned stream -- and the per-row kernels
 *     run beside them on another context-owned stream; all of it is joined back into `stream` before the call returns, so
 *     the caller sees ordinary stream order. Rows are independent, so each
 *     row has its own frame cursor: the vision updater's feedback (net/sig_mp.py:264-271) and the one-shot init_net
 *     (L178-183) make only THAT row wait (8 / 6 ticks, once per occlusion / once per sequence) while the batch keeps
 *     ticking; the updater's rnn6 / rnn4 steps ride the launches of the ring slot that starts when the frame's tail runs.
 *     This is the full-sequence form of the recurrence (the reference's own is RNN.forward over packed sequences,
 *     articulate/utils/torch/rnn.py:129-133); every row's outputs and states are bitwise those of the frame-stepped launches;
 *   - mode 1 takes that engine when the plan's cost estimate beats the frame-stepped launches (short calls with lagging
 *     rows do not) and the call has >= min_frames frames; mode 2 takes it whenever the call has >= min_frames frames;
 *   - frame-stepped frames run without the three transition launches when the plan proves that no row needs one.
 * mode = 0: frame-stepped launches only, no pre-pass, no synchronisation. Live contexts (rc_params.live) always behave
 * like mode 0. rc_get_sequence_stats: frames run by each engine and ticks launched since rc_create (any may be NULL).
 * rc_plan_sequence (transition-launch marks of frame-stepped frames) and rc_plan_wave are the planners alone on HOST data
 * (tests): codes[T*B] (0: c <= lo, 1: mid, 2: c >= hi; frame-major), first_reach[B], pend[B] (state in front of frame 0 /
 * t0) -> mode_out[T] (0 frame-stepped with the three transition launches, 1 without);
 * rc_plan_wave -> *n_ticks, *n_prep (ticks that start frames or riders), frame_at[n_prep*B] (frame row b starts at tick k,
 * or -1), counts[4*n_prep] (rows starting a frame / of them visible / riders / init_net rows, per tick; may be NULL),
 * est_us[2] (cost estimates wavefront / frame-stepped; may be NULL). RC_ERR_INVALID with *n_prep set when frame_at_cap
 * (ints) is too small. */
int rc_set_sequence_mode(rc_ctx* ctx, int32_t mode, int32_t min_frames);
int rc_get_sequence_stats(rc_ctx* ctx, int64_t* wave_frames, int64_t* stepped_frames, int64_t* ticks);
/* Launch counts since rc_create: launches of the shared-weight kernel of the LSTM layer steps (rc_gemm_lds_kernel, round 6: contexts of
 * >= RC_LDS_MIN_BATCH rows in split-product mode), and launches of the other wide / mid-tile kernels (either may be NULL). Replaces
 * nothing in the reference: its per-frame loop (net/sig_mp.py:126-129) has no launch structure to mirror; this is introspection for
 * bench.py and the tests. (Round 5 counted its one-launch-per-tick kernel in the first slot; removed, profiles/r06_tick_path_removed.diff.) */
int rc_get_launch_stats(rc_ctx* ctx, int64_t* lds_launches, int64_t* other_wide_launches);
/* ... and, of `other_wide_launches`, those that ran on rc_gemm_split48_w32_kernel (contexts of 33-64 rows: one-reader launches stream the fp32
 * weights): bench.py names the kernel whose launches it timed (round-5 advisor item). */
int rc_get_launch_stats_w32(rc_ctx* ctx, int64_t* w32_launches);
/* Resident layer-step kernel of the wavefront engine (round 6; north_star: "fused persistent kernel ... across timesteps"). With enable != 0
 * a planned rc_sequence call of a context of 65 .. 256 rows in split-product mode runs the LSTM layer steps AND the linear1 layers of
 * ALL its ticks (net/sig_mp.py:126-129 over every frame of the call; articulate/utils/torch/rnn.py:129-133 is the reference's own
 * whole-sequence form) in ONE launch of `workgroups` (default 224, at most 240; 0 keeps the current value) resident workgroups that take work items from a
 * queue in device memory, ordered by counters instead of stream events; prep / linear2 / fuse / tail stay launches of the second stream.
 * Results are bitwise those of the stream engine. Default off (RC_SEQ_RESIDENT=1 switches it on at rc_create): measured slower than the
 * three-stream ticks. A wait inside the kernel that runs out (RC_SEQ_RESIDENT_BOUND_MS, 2000) marks the call failed: the NEXT
 * rc_sequence returns RC_ERR_STATE once and counts an abort. rc_get_resident_stats: segments run / aborts seen (either may be NULL). */
int rc_set_resident(rc_ctx* ctx, int32_t enable, int32_t workgroups);
int rc_get_resident_stats(rc_ctx* ctx, int64_t* segments, int64_t* aborts);
int rc_plan_sequence(const int8_t* codes, int32_t B, int32_t T, const int32_t* pend, uint32_t flags, int32_t use_vision_updater,
                     uint8_t* mode_out);
int rc_plan_wave(const int8_t* codes, int32_t B, int32_t T, int32_t t0, const int32_t* first_reach, const int32_t* pend,
                 int32_t use_imu_updater, int32_t use_vision_updater, int32_t* frame_at, int64_t frame_at_cap,
                 int32_t* n_ticks, int32_t* n_prep, int32_t* counts, double* est_us);

/* ---- live / streaming mode (BASELINE config 5) ------------------------------------------------------------- */
/* The live_server.py loop (live_server.py:40-48): one frame per call, HOST tensors in and out exactly like
 * forward_online's CPU tensors. rc_live_begin captures the steady-state frame (H2D of the 171 input floats per row,
 * the 14 kernels, D2H of the 219 output floats; for batch <= 16 the kernels access the pinned host buffers directly and
 * the copies disappear) into a hipGraph on a private stream -- twice, with and without the three transition launches;
 * rc_live_step replays the short one when the confidences it was handed rule out a transition step, else the full one
 * (frames with first_tran / RC_FLAG_FIRST_FRAME take the ordinary enqueue path) and returns when the outputs are in
 * host memory. j2dc[batch,33,3], accc[batch,6,3], oric[batch,6,3,3], first_tran[batch,3]|NULL -> pose[batch,24,3,3],
 * tran[batch,3].
 * Round 6: before the pre-built AQL packet chain of the lean frame (batch <= 4) is trusted, rc_live_begin runs ONE probe frame through it
 * and through the graph replay of the same launches from the same state (every small device buffer of the context is saved and put back:
 * the check leaves no trace) and requires the same bits; otherwise the chain is dropped and live frames replay the graph
 * (rc_get_live_backend: aql = 0 and the reason). RC_LIVE_AQL_SELFCHECK=0 skips the check. */
int rc_live_begin(rc_ctx* ctx);
int rc_live_step(rc_ctx* ctx, const float* j2dc_host, const float* accc_host, const float* oric_host,
                 const float* first_tran_host, uint32_t flags, float* pose_host, float* tran_host);
int rc_live_end(rc_ctx* ctx);
/* For batch <= 4 rc_live_begin also captures the LEAN plan of the steady-state frame (csrc/rc_live.hip): seven dependent launches --
 * prep + linear1 | LSTM l0 | LSTM l1 + linear2 partial sums | (partials, fuse) + linear1 | LSTM l0 | LSTM l1 + partials | (partials) +
 * tail -- replayed for every frame that needs neither a transition step nor rnn2.init_net (L178-183) and is not a sequence start;
 * the other frames take the captures above. Same layer-step arithmetic as rc_step; linear2 sums per-tile partial products in a
 * fixed order instead of one MFMA chain, so a lean frame equals an rc_step frame to rounding (<= 1e-6), not bit for bit.
 * RC_LIVE_LEAN=0 in the environment of rc_create switches it off. Counters: frames replayed from the lean / the full captures. */
int rc_get_live_stats(rc_ctx* ctx, int64_t* lean_frames, int64_t* full_frames);
/* Round 5, the idle-time pre-step of a paced live stream (BASELINE config 5: "60 fps"; live_server.py:40-48 receives one camera frame
 * every 16.6 ms): when the caller left the device idle in front of a frame (>= RC_LIVE_PRESTEP_IDLE_US, default 500 us), rc_live_step
 * enqueues, behind that frame, the recurrent halves W_hh h(t-1) of the NEXT frame's twelve LSTM layer steps (half of the 243 MB of weights,
 * one launch, while the device would idle); the next lean frame then starts from them and streams only the input halves. Bitwise the same
 * layer steps. Any eager call in between (reset, rc_step, ...) discards them. presteps: enqueued since rc_create; available: the AQL chain
 * carries the pre-step programs (either may be NULL). */
int rc_get_live_prestep(rc_ctx* ctx, int64_t* presteps, int32_t* available);
/* Lean live frames that turned out not to be the lean plan's -- a row needed a transition step or triggered init_net (net/sig_mp.py:178-183,
 * L264-271) and the conservative host-side mirror of those flags in rc_live_step did not foresee it. The plan's first kernel checks both on the
 * device; such a frame changes nothing and rc_live_step replays it on the full capture, so the caller only sees a slower frame. Expected: 0. */
int rc_get_live_replayed(rc_ctx* ctx, int64_t* frames);
/* RC_LIVE_SPIN=1 (opt-in since round 6; a back-to-back caller's queue-ahead, RC_LIVE_SPIN_B2B, stays on): rc_live_step launches the first kernel of the NEXT lean frame before it returns; that kernel waits on the device for the
 * frame's inputs, which the next call writes straight into device memory. taken: frames that started from a waiting kernel; lost: waiting
 * kernels sent away (the next frame was not the lean plan's, something touched the state in between) or that gave up after 100 ms. */
int rc_get_live_spin(rc_ctx* ctx, int64_t* taken, int64_t* lost);
/* Host time of rc_live_step averaged over the lean frames so far, microseconds: {staging the inputs + choosing the capture, enqueue
 * (hipGraphLaunch), waiting for the frame, copying the outputs}. The frame's GPU time is inside the third. */
int rc_get_live_profile(rc_ctx* ctx, double* avg_us4);
/* The same split for the MOST RECENT lean frame, plus {1 if it started from a kernel waiting on the device, 1 if it used a pre-step}:
 * what bench.py attributes the slowest frames of a paced run with (host wake-up vs device time). Introspection; the reference's
 * live loop (live_server.py:40-48) has no counterpart. */
int rc_get_live_last_profile(rc_ctx* ctx, double* us6);
/* What rc_live_step uses for steady-state frames after rc_live_begin: *lean_captured = the seven-launch capture exists;
 * *aql = its dispatches are pre-built AQL packets on an HSA queue of the context's own (csrc/rc_aql.cpp: ~0.5 us of host time per
 * frame instead of hipGraphLaunch's ~7 us; RC_LIVE_AQL=0 switches it off); note: why not, when it is not (may be NULL). With one row the
 * last kernel of the chain stores the frame's sequence number to a pinned host word once every write of the frame is released at system
 * scope, and rc_live_step returns on that word (RC_LIVE_DONE_FLAG=0: on the queue's completion signal) -- outputs and the context's device
 * state are complete either way; rc_live_end / rc_destroy wait for the queue itself to drain. */
int rc_get_live_backend(rc_ctx* ctx, int32_t* lean_captured, int32_t* aql, char* note, int32_t note_len);

/* ---- per-op entry points (tests, harness; each a single kernel) ------------------------------------------ */
/* art.math.r6d_to_rotation_matrix (articulate/math/angular.py:249-264): r6d[n,6] -> R[n,3,3]. */
int rc_r6d_to_rotmat(const float* r6d, float* R, int64_t n, void* stream);
/* art.math.axis_angle_to_rotation_matrix (articulate/math/angular.py:221-233): aa[n,3] -> R[n,3,3]. */
int rc_axis_angle_to_rotmat(const float* aa, float* R, int64_t n, void* stream);
/* art.math.rotation_matrix_to_axis_angle (articulate/math/angular.py:236-246; the reference loops cv2.Rodrigues on
 * the host): R[n,3,3] -> aa[n,3], angle in [0, pi]. Restates the matrix -> vector branch of OpenCV 4.2's cvRodrigues2 in
 * float64: range check, nearest orthonormal matrix, acos, the s < 1e-5 branches (zero near the identity, sqrt of the
 * diagonal near pi). */
int rc_rotmat_to_axis_angle(const float* R, float* aa, int64_t n, void* stream);
/* art.math.rotation_matrix_to_r6d (articulate/math/angular.py:267-274): R[n,3,3] -> r6d[n,6] = first two columns. */
int rc_rotmat_to_r6d(const float* R, float* r6d, int64_t n, void* stream);
/* art.math.angle_between(rot1, rot2) for rotation matrices (angular.py:128-141): |Rodrigues(R1^T R2)| -> out[n]. */
int rc_angle_between(const float* R1, const float* R2, float* out, int64_t n, void* stream);
/* art.math.lerp(a, b, t) with a Python-double weight (articulate/math/general.py:15-24): out[n] = a * float(1 - t) +
 * b * float(t), the weights formed in double on the host exactly like `tensor * python_float`. */
int rc_lerp(const float* a, const float* b, double t, float* out, int64_t n, void* stream);
/* art.math.normalize_tensor(x, dim=-1, return_norm) (general.py:27-39): x[rows,width] -> out = x / |x|, norm[rows]|NULL. */
int rc_normalize_rows(const float* x, float* out, float* norm, int64_t rows, int32_t width, void* stream);
/* Keypoint normalisation of forward_online (net/sig_mp.py:150-152 with get_bbox_scale L277-284): kp[n,33,3] -> out[n,33,3]
 * (xy / max(bbox width, height), rows != 23 relative to row 23, confidence copied). */
int rc_bbox_normalise(const float* kp, float* out, int64_t n, void* stream);
/* Shape blendshapes of ParametricModel.get_zero_pose_joint_and_vertex(shape) (articulate/model.py:88-91), before its root
 * alignment: verts_out[V,3] = tensordot(beta, shapedirs) + v_template, joints_out[24,3] = J_regressor . verts_out. All
 * pointers HOST (v_template[V,3], shapedirs[V,3,10], J_regressor[24,V] dense, beta[10]); computed on the device,
 * synchronous. The caller hands the shaped joints / landmark vertices to rc_set_body (and the vertices to rc_set_mesh):
 * every entry point of the context then works on the shaped body, which is how forward_kinematics(shape=...) and
 * smplify_runner(shape=...) are served (one shape per context, as TemporalSMPLify holds one per sequence). */
int rc_shape_body(rc_ctx* ctx, const float* v_template, const float* shapedirs, const float* J_regressor, const float* beta,
                  int32_t V, float* verts_out, float* joints_out);
/* ParametricModel.forward_kinematics_R (articulate/math/spatial.py:170-194): Rl[n,24,3,3] local -> Rg[n,24,3,3] global. */
int rc_fk_r(rc_ctx* ctx, const float* Rlocal, float* Rglobal, int64_t n, void* stream);
/* ParametricModel.bone_vector_to_joint_position / joint_position_to_bone_vector (spatial.py:126-167): [n,24,3] both. */
int rc_bone_to_joint(rc_ctx* ctx, const float* bone, float* joint, int64_t n, void* stream);
int rc_joint_to_bone(rc_ctx* ctx, const float* joint, float* bone, int64_t n, void* stream);
/* ParametricModel.get_zero_pose_joint_and_vertex(shape=None) (articulate/model.py:78-93): joint[24,3] = J - J[0] and,
 * when vert is not NULL (needs rc_set_mesh), vert[V,3] = v_template - J[0]. */
int rc_zero_pose(rc_ctx* ctx, float* joint, float* vert, void* stream);
/* ParametricModel.inverse_kinematics_R (articulate/math/spatial.py:197-221): Rg[n,24,3,3] -> Rl[n,24,3,3]. */
int rc_ik_r(rc_ctx* ctx, const float* Rglobal, float* Rlocal, int64_t n, void* stream);
/* fk() of forward_online (net/sig_mp.py:131-135): joints[n,24,3] from GLOBAL rotations + rest bone vectors. */
int rc_fk_bone(rc_ctx* ctx, const float* Rglobal, float* joints, int64_t n, void* stream);
/* ParametricModel.forward_kinematics(calc_mesh=True) restricted to the landmarks + sync_mp3d
 * (articulate/model.py:209-241, net/sig_mp.py:287-299): pose[n,24,3,3] local, tran[n,3] ->
 * grot[n,24,3,3] (may be NULL), joint[n,24,3], j33[n,33,3]. */
int rc_body_fk(rc_ctx* ctx, const float* pose, const float* tran, float* grot, float* joint, float* j33,
               int64_t n, void* stream);
/* One step of sub-net `net` ("rnn2".."rnn8") = f(i, x) of forward_online (net/sig_mp.py:126-129) on its
 * recurrent state: x[batch, in] -> y[batch, out]. row_mask DEVICE uint8[batch] or NULL (all rows). */
int rc_lstm_step(rc_ctx* ctx, const char* net, const float* x, const uint8_t* row_mask, float* y, void* stream);
/* Full-mesh skinning for the metrics of evaluate.py:120-133 (cal_mpjpe: PVE and regressor joints need every vertex).
 * rc_set_mesh uploads v_template[V,3] and weights[V,24] (HOST pointers, same J / parent as rc_set_body);
 * rc_body_mesh = ParametricModel.forward_kinematics(pose, tran, calc_mesh=True)[2] (articulate/model.py:229-241):
 * pose[n,24,3,3] local, tran[n,3] -> vert[n,V,3]. */
int rc_set_mesh(rc_ctx* ctx, const float* v_template_host, const float* weights_host, int32_t V);
int rc_body_mesh(rc_ctx* ctx, const float* pose, const float* tran, float* vert, int64_t n, void* stream);
/* Metrics of evaluate.py:120-133 (cal_mpjpe) fused on the device: both meshes skinned with zero translation,
 * keypoints = J_regressor[:n_used] . vertices (rc_set_regressor: HOST [n_rows,V] row-major, n_used = 14 in the
 * reference; without a regressor the 24 SMPL joints stand in), pelvis alignment, then per frame
 * {MPJPE, PVE, PA-MPJPE (Procrustes, utils.py:138-203)} -> per_frame DEVICE [n,3]. mean_host (HOST double[3] or NULL)
 * receives the means over the n frames (synchronises `stream`). */
int rc_set_regressor(rc_ctx* ctx, const float* j_regressor_host, int32_t n_rows, int32_t n_used);
int rc_mesh_metrics(rc_ctx* ctx, const float* pose, const float* gt_pose, int64_t n, float* per_frame, double* mean_host,
                    void* stream);
/* IMU synthesis of the dataset preparation (preprocess.py:22-33 `_syn_acc`, :206-214): pose[T,24,3,3] local rotations,
 * tran[T,3] DEVICE; vertex_ids[6] (config.vi_mask) and joint_ids[6] (config.ji_mask) HOST; needs rc_set_mesh.
 * -> imu_ori[T,6,3,3] = global rotations of joint_ids, vert6[T,6,3] = skinned vertices, imu_acc[T,6,3] =
 * _syn_acc(vert6, smooth_n), joint3d[T,24,3] (or NULL). rc_syn_acc is the stencil alone on v[T,width]: second
 * differences * 3600 with zero end frames, interior [n:-n] from the wide stencil / n^2 when smooth_n >= 2 (bit-exact).
 * Like the reference, smooth_n >= 2 needs T >= 2 * smooth_n + 1. */
int rc_synth_imu(rc_ctx* ctx, const float* pose, const float* tran, const int32_t* vertex_ids_host,
                 const int32_t* joint_ids_host, int64_t T, int32_t smooth_n, float* imu_ori, float* imu_acc, float* joint3d,
                 float* vert6, void* stream);
int rc_syn_acc(const float* v, int64_t T, int64_t width, int32_t smooth_n, float* acc, void* stream);
/* reconstruction_error(S1, S2, reduction=None) (utils.py:189-203) on raw point sets: S1, S2 DEVICE [n, n_points, 3]
 * -> err DEVICE [n] = mean point distance after the optimal scale * rotation + translation of S1 onto S2. */
int rc_procrustes_error(const float* S1, const float* S2, int64_t n, int32_t n_points, float* err, void* stream);
/* PositionErrorEvaluator (articulate/evaluator.py:100-129): p[n,3], t[n,3] DEVICE -> dist[n] DEVICE and, if mean_host
 * is not NULL, their mean (synchronises `stream`). */
int rc_position_error(const float* p, const float* t, int64_t n, float* dist, double* mean_host, void* stream);
/* smplify forward residual (net/smplify/temporal_smplify.py:198-220 -> losses.py:36-37,43-46): pose[T,24,3,3],
 * tran[T,3], kp[T,33,3] in pixels, K[3,3] (DEVICE) -> loss[T,33]. The confidences of landmarks
 * {1..9,31,32} count as zero. */
/* Landmarks whose confidence smplify zeroes (temporal_smplify.py:92-94): default {1..9, 31, 32}; `use_head=True` of the
 * reference = {31, 32}. ids HOST int32[n], each in 0..32. Applies to rc_reproj_residual and the optimiser. */
int rc_set_ignored_landmarks(rc_ctx* ctx, const int32_t* ids_host, int32_t n);
int rc_reproj_residual(rc_ctx* ctx, const float* pose, const float* tran, const float* kp, const float* K,
                       float sigma, float* loss, int64_t T, void* stream);

/* ---- smplify optimiser ------------------------------------------------------
 * Replaces net/smplify/run.py:6-34 (smplify_runner) -> temporal_smplify.py:97-196 (TemporalSMPLify.__call__):
 * L-BFGS (torch.optim.LBFGS, strong Wolfe) over [body_pose (T*72 axis-angle), tran (T*3)] of the whole sequence on
 * temporal_body_fitting_loss (net/smplify/losses.py:23-87). Loss and analytic gradient run on the device
 * (rc_smplify.hip), the line-search scalars on the host. */

/* GMM pose prior of MaxMixturePrior (net/smplify/prior.py:102-147): HOST means[8,69], precisions[8,69,69]
 * (inverse covariances) and nll_weights[8] (the weights already divided by the determinant term, prior.py:137-141). */
int rc_smplify_set_prior(rc_ctx* ctx, const float* means_host, const float* precisions_host, const float* nll_weights_host);

/* One closure evaluation (temporal_smplify.py:150-166): x DEVICE [T*72 | T*3] flat parameter vector, kp[T,33,3]
 * pixels + confidence, ref3d[T,33,3] landmarks of the initial prediction, imu_aa[T,18] axis-angle of the IMU
 * orientations (all DEVICE), K[3,3] HOST. Writes the total loss (HOST double) and grad DEVICE [T*72 | T*3].
 * Synchronises `stream`. */
int rc_smplify_loss_grad(rc_ctx* ctx, const float* x, const float* kp, const float* ref3d, const float* imu_aa,
                         const float* K_host, int64_t T, double* loss_host, float* grad, void* stream);

/* With shape=..., the reference keeps the preserved 3D landmarks of the INITIAL prediction on the MEAN-shape body
 * (temporal_smplify.py:112 calls forward_kinematics without shape) while everything else uses the shaped body. A context
 * holds one body, so the caller computes those landmarks on a mean-shape model and hands them over: ref3d DEVICE [T,33,3],
 * used by the NEXT rc_smplify_run instead of the landmarks of the context's own body (then forgotten); NULL cancels. */
int rc_smplify_set_ref3d(rc_ctx* ctx, const float* ref3d);

typedef struct rc_smplify_info {
    int32_t status;          /* 0: rejected by the pre-check (run.py:27-29), 1: optimised */
    int32_t n_iter, n_eval;  /* L-BFGS iterations / closure evaluations */
    int32_t reserved;
    double first_loss, final_loss;   /* closure value at the prediction / at the accepted point */
    double host_ms, device_ms;       /* wall time of the call / HIP-event time of the closure evaluations */
} rc_smplify_info;

/* smplify_runner(pred_pose, pred_tran, j2dc, imu_ori, batch_size=T, cam_k, lr, opt_steps=1, use_lbfgs=True,
 * loss_threshold): pose[T,24,3,3] local rotations, tran[T,3], kp[T,33,3], imu_ori[T,6,3,3] DEVICE; K HOST.
 * Outputs pose_out / tran_out DEVICE (the inputs copied through when the pre-check rejects the sequence) and
 * update HOST uint8[T] (new per-frame mean residual < old; all zero when rejected -- the reference returns None,
 * see info->status). max_iter = 20 and max_eval = 25 are torch's defaults the reference leaves untouched. */
int rc_smplify_run(rc_ctx* ctx, const float* pose, const float* tran, const float* kp, const float* imu_ori,
                   const float* K_host, int64_t T, float lr, int32_t max_iter, float loss_threshold, float* pose_out,
                   float* tran_out, uint8_t* update_host, rc_smplify_info* info, void* stream);

/* The same for n_rows independent (sequence, camera) rows at once (evaluate.py:86-90 loops them): every row runs the optimiser
 * of rc_smplify_run as a fiber of the CALLING thread (a stack of its own behind a guard page; RC_SMPLIFY_THREADS=1: a host thread per
 * row instead), the device work of all rows goes out in lock-step rounds -- one launch per kind
 * of operation over all rows, one read-back and one synchronisation per round -- so 72 rows cost about what the longest row's ~46
 * rounds cost. Per row the arithmetic is that of rc_smplify_run (same n_iter / n_eval / losses). Arrays of n_rows: T, DEVICE
 * pointers pose / tran / kp / imu_ori / pose_out / tran_out, HOST pointers update (uint8[T_r] each), HOST K[n_rows,9], infos
 * (host_ms / device_ms are those of the whole batch; reserved = rounds). Rows rejected by the pre-check are copied through. */
int rc_smplify_run_batch(rc_ctx* ctx, int32_t n_rows, const int64_t* T_rows, const float* const* pose, const float* const* tran,
                         const float* const* kp, const float* const* imu_ori, const float* K_host, float lr, int32_t max_iter,
                         float loss_threshold, float* const* pose_out, float* const* tran_out, uint8_t* const* update_host,
                         rc_smplify_info* infos, void* stream);

/* The optimiser alone, in float64, on a caller-supplied objective (HOST): same algorithm object as rc_smplify_run
 * with Real = double. tests/ pin it against torch.optim.LBFGS evaluation by evaluation. objective returns the loss
 * at x[n] and fills grad[n]. x is updated in place; losses_out (capacity cap) receives every objective value in
 * evaluation order. */
typedef double (*rc_objective_fn)(void* user, const double* x, double* grad, int64_t n);
int rc_lbfgs_minimize(rc_objective_fn objective, void* user, int64_t n, double* x, double lr, int32_t max_iter,
                      int32_t max_eval, int32_t history_size, double tolerance_grad, double tolerance_change,
                      int32_t* n_iter_out, int32_t* n_eval_out, double* losses_out, int64_t cap);

/* Harness input preparation of evaluate.py:38-51,70-73 for ONE (sequence, camera) of n frames:
 *   j2dc = K^-1 [u, v, 1] with the confidence copied into the last channel, accc = R_cw acc_w, oric = R_cw ori_w,
 *   gravity_out (HOST float[3]) = R_cw [0, -1, 0].
 * kp_pix[n,33,3] = (u, v, conf) in pixels, imu_acc_w[n,6,3], imu_ori_w[n,6,3,3] DEVICE; K[3,3], Tcw[4,4] HOST. */
int rc_camera_inputs(const float* kp_pix, const float* imu_acc_w, const float* imu_ori_w, const float* K_host,
                     const float* Tcw_host, float* j2dc, float* accc, float* oric, float* gravity_out_host, int64_t n,
                     void* stream);

/* The same preparation for ALL rows of an evaluation in one launch (evaluate.py:32-51,66-73 loops sequences, cameras and
 * frames on the host). Row r = one (sequence, camera): kp_norm[n_rows,Tmax,33,3] = (u / image_w, v / image_h, conf) as the
 * preprocessed dataset stores them (evaluate.py:43-44 multiplies the image size back), imu_acc_w[n_seq,Tmax,6,3] and
 * imu_ori_w[n_seq,Tmax,6,3,3] per SEQUENCE, seq_of_row[n_rows] (which sequence a row belongs to), len[n_rows] (valid frames of
 * the row; later frames are written as padding: zero keypoints and accelerations, identity orientations) -- all DEVICE;
 * K[n_rows,3,3], Tcw[n_rows,4,4] HOST. Outputs j2dc[n_rows,Tmax,33,3], accc[n_rows,Tmax,6,3], oric[n_rows,Tmax,6,3,3] DEVICE,
 * gravity_out[n_rows,3] HOST. cam_scratch: caller-owned DEVICE buffer of n_rows * 72 bytes (camera constants).
 * Synchronises `stream` once (upload of the camera constants). */
int rc_camera_inputs_rows(const float* kp_norm, const float* imu_acc_w, const float* imu_ori_w, const int32_t* seq_of_row,
                          const int32_t* len, const float* K_host, const float* Tcw_host, float image_w, float image_h,
                          int32_t n_rows, int32_t Tmax, float* j2dc, float* accc, float* oric, float* gravity_out_host,
                          void* cam_scratch, void* stream);

/* ---- state access (tests / checkpointing of a running sequence) ------------------------------------------ */
/* Copy the (h, c) state of sub-net `net` to HOST buffers h[2,batch,H], c[2,batch,H]. Synchronises `stream`. */
int rc_get_state(rc_ctx* ctx, const char* net, float* h_host, float* c_host, void* stream);
/* Per-row fusion state (net/sig_mp.py:85-104 and the class attributes L43-45) for tests and debugging: out[batch,5] =
 * {last_tran is set, len(floor_y), first_reach, update_vision_count, a deferred vision-updater step is pending}. Synchronises. */
int rc_get_fusion_state(rc_ctx* ctx, int32_t* out_host, void* stream);

/* Per-row branch trace of the last step, HOST int32[batch,8]:
 * {regime (0 low,1 mid,2 high), rnn4 steps, rnn6 steps, floor samples held, reach fired, used velocity branch,
 *  stance foot, jump reset}. Synchronises `stream`. */
int rc_get_trace(rc_ctx* ctx, int32_t* trace_host, void* stream);

/* Timing hook for bench.py: accumulate HIP-event time of the gate-GEMM launches on their own stream.
 * enable = 1 records every gate-GEMM launch, enable = 2 only those of the wide-tile kernel rc_gemm_kernel (the 16-row
 * launches run on rc_gemm_small_kernel), enable = 3 only those of the shared-weight kernel rc_gemm_lds_kernel (round 6: the LSTM layer
 * steps of contexts of more than 64 rows), 0 stops; rc_gemm_timing_read returns total milliseconds and launch count so far. */
int rc_gemm_timing(rc_ctx* ctx, int32_t enable);
int rc_gemm_timing_read(rc_ctx* ctx, double* total_ms, int64_t* launches);
/* Time (ms) during which at least one of the launches read so far was running: the wavefront engine issues the two wide launches
 * of a tick on two streams, so their durations overlap and the sum above counts that time twice. Valid after rc_gemm_timing_read. */
int rc_gemm_timing_busy(rc_ctx* ctx, double* busy_ms);

#ifdef __cplusplus
}
#endif
#endif /* ROBUSTCAP_HIP_H */
