// Internal declarations shared by the HIP translation units of librobustcap_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- GEMM tiling -------------------------------------------------------------------------------------------
// One workgroup = RC_NW waves = one 32-row x (16*NC)-column output tile; the K range is split across the waves
// (in-workgroup split-K) and reduced through LDS. Each wave runs v_mfma_f32_16x16x4_f32 on 2 row blocks x NC
// column blocks (A: lane l = row l&15, k-quarter l>>4; B: k-quarter l>>4, column l&15).
#define RC_MT 32          // rows per workgroup tile of dense layers (2 MFMA row blocks); LSTM tiles: 16 * mr
#define RC_NT 64          // columns per workgroup tile of dense layers (NC = 4 blocks of 16)
#ifndef RC_NW
#define RC_NW 4           // waves per workgroup (K split)
#endif
#define RC_KC 16          // k per chunk: one dwordx4 (4 consecutive k) per lane per operand block
#define RC_KALIGN 128     // padded K granularity (>= 2 * RC_KC * RC_NW: an even number of chunks per wave)
#define RC_HBUF 3          // copies of every hidden state h: step s writes copy s % 3 and reads copy (s - 1) % 3. Two would do for the
                           // recurrence; the third lets the first wide launch of a wavefront tick (which writes copy (s + 2) % 3)
                           // start while linear2 of the previous tick -- second stream -- still reads copy s % 3
#define RC_MAX_PROB 14    // problems fused in one launch (GemmLaunch travels as a kernel argument: 14 x 232 B = 3.2 KB; no scratch copy: checked in the ISA)

// Every GEMM A operand (sub-net inputs, relu(linear1), hidden states) is stored in MFMA A-fragment order so that a
// wave's A load is one contiguous 1 KiB piece, exactly like the packed weights: for 16-row block row / 16 and
// 16-wide k chunk k / 16 the 256 floats [k-quarter (k/4)&3][row & 15][k & 3] are contiguous. (Row-major rows would
// make each dwordx4 load touch many cache lines; the vector-memory front end, not HBM, was the bound -- profiles/.)
__host__ __device__ inline long long rc_pk(long long row, int k, int ld) {
    return (row >> 4) * (16ll * ld) + (long long)(k >> 4) * 256 + ((k >> 2) & 3) * 64 + (row & 15) * 4 + (k & 3);
}

enum { RC_EPI_DENSE = 0, RC_EPI_RELU = 1, RC_EPI_LSTM = 2 };
enum { RC_PAR_NONE = 0, RC_PAR_SRC = 1, RC_PAR_DST = 2 };

// row flags (one byte per row, rebuilt every frame)
#define RC_ROW_VIS 1u     // rnn4 steps on the camera keypoints   (c > lo or first_frame)
#define RC_ROW_PC 2u      // rnn6 steps on the camera keypoints   (c > lo)
#define RC_ROW_UPD 4u     // vision updater: rnn6/rnn4 step on re-projected landmarks (c <= lo)
#define RC_ROW_REACH 8u   // rnn2 state re-initialised by init_net this frame
#define RC_ROW_MASK 16u   // caller-supplied mask (rc_lstm_step)
// second flag byte (flags2): scheduling of the DEFERRED vision-updater steps. The updater's rnn6/rnn4 steps of
// frame t only change state that frame t+1 reads, so they are executed at the start of frame t+1: merged into that
// frame's own rnn4 / rnn6 launch when the row does not step there anyway (the row just reads its "late" input
// buffer), or in a small transition launch first when it does (regime change low -> high).
#define RC_ROW2_TR 1u     // pending updater step AND the row steps again this frame: transition launch
#define RC_ROW2_M4 2u     // rows of the merged rnn4 launch  (VIS | pending)
#define RC_ROW2_M6 4u     // rows of the merged rnn6 launch  (PC  | pending)
#define RC_ROW2_FLUSH 8u  // rc_get_state: run every pending step now
#define RC_ROW2_VALID 16u // per-row-cursor wavefront engine: the row carries a frame in this ring slot (not a bubble)

struct GemmSeg {
    const float* base;      // activation matrix [rows, ld] in rc_pk order
    long long par_stride;   // elements between the RC_HBUF copies (0 if not multi-buffered)
    int ld;
    int K;                  // padded length of this K segment (multiple of RC_KALIGN; 0 = absent)
    int par_mode;           // RC_PAR_*
    int pad_;
};

struct GemmProblem {
    GemmSeg seg[2];         // A = [seg0 | seg1] along K
    const float* W;         // packed weights, see pack_weights() in rc_api.cpp
    const void* Ws;         // the same weights as three bf16 planes (split-bf16 products), see pack_weights_split()
    const float* bias;      // [n_tiles * 64] in packed column order
    float* out;             // dense: out[row * ldo + col0 + n], or rc_pk(row, col0 + n, ldo) if out_packed
    float* hstate;          // lstm: h[parity][row][H]
    float* cstate;          // lstm: c[row][H]
    int* steps;             // per-row step counter of this net (copy = steps % RC_HBUF)
    const unsigned char* flags;
    const float* alt_base;  // seg[0] of rows WITHOUT sel_bit in sel_flags reads here (deferred updater input)
    const unsigned char* sel_flags;
    const unsigned char* out_flags;   // dense: only rows with out_bit set are written (0 = all active rows)
    long long h_par_stride;
    int ldo, N, H;
    int out_col0, out_packed;
    int sel_bit, out_bit;
    int flag_bit;           // 0 = all rows
    int epi;                // RC_EPI_*
    int open_step;          // linear1 opens a step: the n_tile 0 workgroup increments steps[row]
    int n_tiles, m_tiles, wg_base, Kp;
    int nc;                 // 16-column blocks per tile
    int mr;                 // 16-row blocks per tile (2 or 4); (mr, nc) must be one of the instantiated shapes
    int trace_base;         // first record slot of this launch (read by -DRC_TRACE_TILES builds only)
    int nt;                 // live frames: stream THIS problem's weights with non-temporal loads (rc_gemm_small_nt_kernel)
    int step_off;           // added to steps[row] wherever the step parity is formed. Frame-stepped launches: 0 (linear1
                            // has already incremented the counter). Sequence mode: 1 + (frame - first frame of the
                            // segment), with open_step = 0: the counters stand still while stages of several frames are in
                            // flight and are advanced once, after the segment.
};

struct GemmLaunch {
    int n;                  // problems
    int B;                  // rows in the batch
    int split;              // 1: products as split-bf16 partial products on the bf16 MFMA (rc_gemm.hip: mma_kblock), W -> Ws
    int live;               // 1: launch of a live frame (16-row launches then stream their weights with non-temporal loads)
    GemmProblem p[RC_MAX_PROB];
};

#define RC_TICK_PROB 24   // GEMM problems of one sequence-mode tick (6 sub-nets x {linear1, LSTM l0, LSTM l1, linear2})

// ---- shared-weight gate GEMM of the LSTM layer steps (rc_gemm_lds.hip): 256-row x 128-column tiles, 8 waves, the weight planes of
// a k-block staged once per workgroup in LDS, K halved across two workgroups (seg[0] | seg[1]) that meet through a slab in device memory
#define RC_LDS_MAXP 12    // layer steps fused in one launch
#define RC_LDS_SLAB_FLOATS 65536   // per tile: two half sums of 256 x 128 floats
struct LdsProblem {
    GemmSeg seg[2];         // A = [seg0 | seg1] along K, each H long: the layer's input and its own h(t - 1)
    const void* Ws;         // weight planes (pack_weights_split)
    const float* bias;
    float* hstate;
    float* cstate;
    const int* steps;
    const unsigned char* flags;
    float* slab;            // [tiles][2][256 x 128]: the half sums
    int* tickets;           // [tiles], zero between launches
    long long h_par_stride;
    int H, flag_bit, n_tiles, m_tiles, wg_base, Qs, step_off, ksplit;
    // relu(linear1) as an item of the resident kernel (epi = RC_EPI_RELU; ksplit 1, seg[0] | seg[1] = the two halves of ONE input): out
    // [rows, ldo] in rc_pk order; rows without sel_bit in sel_flags read alt[0] | alt[1] instead (a rider's deferred input)
    float* out;
    const float* alt[2];
    const unsigned char* sel_flags;
    int epi, ldo, sel_bit, pad_;
};
struct LdsLaunch {
    int n, B;
    LdsProblem p[RC_LDS_MAXP];
};
#define RC_RES_MAXP 20    // problems of a tick of the resident kernel: twelve layer steps + six linear1
void rc_launch_gemm_lds(const LdsLaunch& L, int total_wg, hipStream_t s, hipEvent_t stop = nullptr);

// ---- resident layer-step kernel (rc_gemm_lds.hip: rc_gemm_resident_kernel) ----------------------------------
// One launch carries the LSTM layer steps of EVERY tick of a wavefront-engine segment: its workgroups stay on their CUs and take work
// items (the items of rc_gemm_lds_kernel, tick after tick, longest first inside a tick) from one queue in device memory. What stream
// order and events did between launches, counters in device memory do between items (rc_api.cpp: run_wave2_resident).
struct ResidentTick {
    int n, B;                             // the tick's problems, as one launch of the shared-weight kernel would carry them
    LdsProblem p[RC_RES_MAXP];
    int n_items;                          // work items of the tick (every problem's range padded to a multiple of 8)
    int dep[RC_RES_MAXP][2];              // per problem: up to two problems of the PREVIOUS tick it reads (own h(t - 1); its input: layer 0's h, relu(linear1)), -1 = none
    int dep_items[RC_RES_MAXP][2];        // ... and their item counts
    int need_tail[RC_RES_MAXP];           // 1: the problem reads what the previous tick's second-stream chain wrote (linear1: fuse / tail / prep; rnn2 behind an init_net state write)
};
struct ResidentArgs {
    const ResidentTick* ticks;
    const int* item_base;                 // [n_ticks + 1] first queue position of every tick
    int n_ticks;
    int* head;                            // queue head
    int* done;                            // [n_ticks][RC_RES_MAXP] items finished per (tick, problem)
    int* tick_done;                       // [n_ticks] items finished per tick
    const int* flag_tail;                 // second stream: tails finished (the chain of tick k done -> k + 1)
    int* abort;                           // set by whoever waited longer than the bound; everybody else stops waiting
    unsigned long long spin_bound;        // wall_clock64 ticks (100 MHz)
};
void rc_launch_gemm_resident(const ResidentArgs& R, int workgroups, hipStream_t s);
void rc_launch_flag_set(int* flag, int value, hipStream_t s);
void rc_launch_flag_wait(const int* counter, int target, int* abort, unsigned long long spin_bound, hipStream_t s);

// ---- per-frame small kernels -------------------------------------------------------------------------------
struct BodyConst {          // device copy of the body constants the path needs
    int parent[24];
    int level[24];          // depth in the kinematic tree
    float bone[24][3];      // rest bone vectors  (j_rest[i] - j_rest[parent[i]])
    float jrest[24][3];     // rest joints, root at the origin
    float jroot[3];         // rest position of the root joint (subtracted from the template, model.py:87)
    float w33[33][24];
    float v33[33][3];       // landmark vertices, root-relative
    int override_joint[33]; // sync_mp3d: landmark row -> joint id, or -1
    int nchild[24];         // children of a joint, ascending (the reverse sweep of the smplify gradient pulls from them)
    int child[24][4];
};

struct FrameBuffers {       // device pointers owned by the context (all [B, ld] row-major)
    float *x2, *x3, *x4, *x6, *x78, *x4l, *x6l, *xi;   // concatenated sub-net inputs (rc_pk order, padded, pads stay zero)
    float *vr, *pc, *r6d, *contact;                     // sub-net outputs consumed by the fusion logic
    float *init_out;                                    // rnn2.init_net output [B, 2048]
    unsigned char* flags;                               // RC_ROW_* per row
    unsigned char* flags2;                              // RC_ROW2_* per row
    unsigned char* pend;                                // 1 = updater step of the previous frame still to run
    unsigned char* regime;                              // 0 low / 1 mid / 2 high
    double* kconf;                                      // (c - lo) / (hi - lo) per row
    float* gravity;                                     // [B,3]
    // per-row fusion state (net/sig_mp.py:85-90)
    float *last_pfoot, *last_tran, *floor, *j_temp;
    int *has_last, *n_floor, *first_reach, *uv_count;
    int* trace;                                         // [B,8]
    // rnn2 state for the init_net write
    float *h2, *c2;
    int* steps2;
    long long h2_par_stride, h2_layer_stride, c2_layer_stride;
    // per-row-cursor wavefront engine (ring slots only; nullptr in the context's own frame-stepped buffers):
    int* frame;             // [B] frame index (relative to the call) the row carries in this slot, -1 = bubble
    int* wsteps;            // [6][B] step number of each sub-net for the step this slot's frame (or rider) takes
};

// rc_prep_wave_kernel: which frame every row starts at this tick (host plan), the global step counters it opens steps on,
// and the context's own updater-input buffers (a deferred updater step pending from before the segment rides the first slot)
struct WavePrep {
    const int* frame_at;    // [B] of this tick
    int* steps[6];          // the sub-nets' per-row step counters
    const float *cx4l, *cx6l;
    int first_tick;
};
// rc_tail_kernel in the per-row-cursor engine: the vision updater's two sub-net steps of a frame "ride" the ring slot that
// is initialised at the tick its tail runs (target slot); the last frame of the segment leaves them pending instead.
struct WaveTail {
    int on;                 // 0 = frame-stepped / all-visible engine
    int t_last;             // last frame of the segment
    float *x4l, *x6l;       // target slot
    unsigned char* flags2;
    int* wsteps;
    int *steps4, *steps6;   // global step counters of rnn4 / rnn6
    float *cx4l, *cx6l;     // the context's own buffers (pending step of the last frame)
};

struct FrameIO {
    const float *j2d, *acc, *ori, *first_tran;
    float *pose_out, *tran_out;
    long long s_j2d, s_acc, s_ori, s_pose, s_tran;      // row strides (elements)
};

struct rc_params_dev {
    double conf_lo, conf_hi, tran_filter_num;
    float contact_threshold, distance_threshold, height_threshold;
    int use_flat_floor, use_vision_updater, use_imu_updater, live, update_vision_freq, use_reproj_opt;
    float smooth;
};

// ---- the lean live frame (rc_live.hip; live_server.py:40-48 calls forward_online once per camera frame) -------------------------------
// Batch <= RC_LIVE_MAXB, steady-state frames only (no first frame, no transition step, init_net already run): SEVEN dependent
// launches instead of 11-14 --
//   K1 prep + linear1{rnn4, rnn2} | K2 LSTM l0 | K3 LSTM l1 + per-tile partial sums of linear2 |
//   K4 (sum of the partials, fuse) + linear1{rnn6, rnn3, rnn7, rnn8} | K5 LSTM l0 | K6 LSTM l1 + linear2 partials |
//   K7 (sum of the partials) + tail
// linear2 y = W2 h + b is computed where h is produced: the workgroup that owns 4 (or 8) hidden units multiplies them with its
// columns of W2 and stores the partial y; the consuming kernel sums the partials of all tiles in a FIXED order behind the launch
// boundary (no atomics, no fences, nothing placement-dependent). Every kernel requests its weights first and its per-row words
// (flags, step parities, cell state) in the same batch: one memory latency in front of the first MFMA instead of a chain of four.
#define RC_LIVE_MAXB 4
// Second argument of the LSTM launches of the lean frame. `hot`: the per-row words a launch needs before its first activation load --
// step number (-> which copy of h) and whether the row steps, per problem of the launch (stage 1: rnn4, rnn2; stage 2: rnn6, rnn3,
// rnn7, rnn8) -- are IN the kernel arguments: on the AQL path (rc_aql.cpp) the arguments live in device memory of the context's own,
// and the linear1 kernel that opens the steps (K1 / K4) writes them there, so the LSTM kernels have no dependent global read in
// front of the weight stream. hot = 0 (graph replay, direct launches): read from the state.
// `pre` (round 5, the idle-time pre-step): 1 = the recurrent half of this launch's layer steps -- the partial sums of waves 2 and 3 of every
// tile, W_hh . h(t - 1) over their K ranges -- was computed after the PREVIOUS frame (rc_live_pre) and sits in `prebuf`; those waves load
// it instead of streaming their half of the weights. Same MFMA chain per accumulator, same reduction order: bitwise the unhoisted step.
struct LiveGrid { int hot; int st[4][RC_LIVE_MAXB]; int act[4][RC_LIVE_MAXB]; int pre; int pre_base; const float* prebuf; };
struct LiveNet {
    const float *W1, *b1;           // linear1: pack_weights order [N/16][Kp1/16][64][4], bias
    const float *Wl[2], *bl[2];     // LSTM layers: pack_weights order, gate-interleaved columns; b_ih + b_hh
    const float *W2, *b2;           // linear2 ROW-MAJOR [out][H] (a unit's column slice is one 16-byte piece per output), bias [out]
    float *x1, *h, *c, *part;       // relu(linear1) [Bp][H] rc_pk | h [2][RC_HBUF][Bp][H] rc_pk | c [2][B][H] | partials [H/UT][RC_LIVE_MAXB][outp]
    int* steps;
    int H, out, outp, Kp1;
    long long BpH;                  // elements between the copies of h
};
struct LiveFrame {
    LiveNet net[6];                 // kNets order: rnn2, rnn3, rnn4, rnn6, rnn7, rnn8
    FrameBuffers fb;
    FrameIO io;
    rc_params_dev prm;
    const BodyConst* body;
    int* status;                    // pinned host word, set != 0 by K1 when the frame is not the lean plan's (a transition step or an init_net trigger:
                                    // the host's mirror of those flags is conservative, this is the check behind it); rc_live_step then replays the full capture
    int* abort;                     // device word K1 writes every frame (0 / 1): set, the frame's kernels change NOTHING (no state, no step counters, no outputs)
    unsigned* spin_mb;              // waiting K1 (rc_live.hip, RC_LIVE_SPIN): its mailbox (one of two, frames alternate) in host-writable device memory -- [0] the host's command (0 none, 1 go,
                                    // 2 skip), [16] the decision workgroup 0 publishes (1 go, 2 skip, 3 timed out); else null
    unsigned* spin_state;           // pinned host word: 3 when the spinning K1 gave up waiting
    LiveGrid* hot[4];               // AQL path: the LiveGrid argument blocks of K2, K3 (written by K1) and K5, K6 (by K4); else null
    unsigned* done_flag;            // AQL path, one row: pinned host word K7 stores the frame's sequence number to, behind a system-scope
    unsigned* done_seq;             // release, when everything of the frame is written (device counter of frames); else null
    int B;
    int nc;                         // 16-column blocks per LSTM tile (1 or 2)
};
#define RC_LIVE_KERNELS 7
struct LiveKernel {             // one launch of the lean frame: host function + symbol name, grid (workgroups of `wg` threads), arguments
    const void* fn;
    const char* name;
    unsigned grid;
    int has_grid;               // arguments: (LiveFrame) or (LiveFrame, LiveGrid)
    unsigned wg;                // threads per workgroup (0 = 256)
    LiveFrame F;
    LiveGrid G;
};
int rc_live_plan(const LiveFrame& F, LiveKernel* out, const float* prebuf = nullptr);   // fills RC_LIVE_KERNELS entries, returns their number;
                                                                      // prebuf: the LSTM launches take their recurrent halves from it (LiveGrid.pre)
int rc_live_pre_plan(const LiveFrame& F, float* prebuf, LiveKernel* out);   // the idle-time pre-step: rc_live_pre (+ rc_live_warm); returns 1 or 2 launches, or 0
long long rc_live_pre_floats(const LiveFrame& F);                     // size of prebuf
void rc_launch_live_frame(const LiveFrame& F, hipStream_t s, const float* prebuf = nullptr);
void rc_launch_live_pre(const LiveFrame& F, float* prebuf, hipStream_t s);

// The same chains as AQL packets on an HSA queue of the context's own (rc_aql.cpp): rc_live_step then pays ~0.5 us of host time to
// enqueue the frame instead of hipGraphLaunch's ~7 us, and polls the completion itself. A chain holds several PROGRAMS (packet lists
// with their own argument blocks) on one queue: the lean frame, the lean frame on pre-computed recurrent halves, and the pre-step.
struct AqlChain;
int rc_aql_create(int hip_device, AqlChain** out, char* err, int err_len);
// a program of n <= RC_LIVE_KERNELS launches; frame != 0: a live frame (its last kernel stores the completion word, its last packet carries
// the frame signal), else a background program (pre-step) with a signal of its own. Returns the program id (>= 0) or -1.
int rc_aql_add(AqlChain* c, const LiveKernel* k, int n, int frame, char* err, int err_len);
int rc_aql_run(AqlChain* c, int prog);                                // frame program: submit, then spin until it retired; 0 = done
int rc_aql_submit(AqlChain* c, int prog);                             // background program: submit and return
int rc_aql_alloc_shared(AqlChain* c, size_t bytes, void** ptr);         // device memory the host can write (large BAR), freed with the chain; 0 = ok
int rc_aql_submit_ahead(AqlChain* c, int prog, int beside);             // a whole frame program submitted without waiting (queued ahead: its K1 waits on the mailbox;
                                                                        // beside: that K1 may start beside the frame in front of it)
unsigned long long rc_aql_seq(const AqlChain* c);                       // number of the frame submitted last
int rc_aql_wait_seq(AqlChain* c, unsigned long long seq);               // spin until frame `seq` has retired
int rc_aql_wait_frame(AqlChain* c);                                     // ... the frame submitted last
int rc_aql_fence_background(AqlChain* c);                               // a barrier packet that counts as a background program (rc_aql_wait_background)
void rc_aql_set_mailbox(AqlChain* c, volatile unsigned* mb);            // the drain tells a K1 still spinning to leave
int rc_aql_arm(AqlChain* c);                         // RC_LIVE_ARM: a barrier-AND packet the next push releases
int rc_aql_wait_background(AqlChain* c);                              // until every submitted background program has retired
void rc_aql_destroy(AqlChain* c);

void rc_launch_gemm(const GemmLaunch& L, int total_wg, hipStream_t s, hipEvent_t stop = nullptr);
void rc_launch_scan_conf(const float* j2d, long long row_stride, int B, int T, double conf_lo, double conf_hi, signed char* codes, hipStream_t s);
void rc_launch_advance_steps(int* const* steps6, int n_frames, int B, hipStream_t s);
bool rc_gemm_is_w32(const GemmLaunch& L);       // true: the launch runs on rc_gemm_split48_w32_kernel
bool rc_gemm_is_small(const GemmLaunch& L);     // true: the launch runs on rc_gemm_small_kernel (16-row tiles only)
void rc_launch_prep(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, int B, int first_frame, hipStream_t s);
void rc_launch_fuse(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, int B, hipStream_t s);
void rc_launch_tail(const FrameBuffers& fb, const FrameIO& io, const rc_params_dev& prm, const BodyConst* body, int B,
                    int first_frame, hipStream_t s, const FrameIO* io_next = nullptr,    // io_next: also the next frame's prep
                    const WaveTail* wt = nullptr, hipEvent_t stop = nullptr);   // stop: event signalled by this dispatch
bool rc_launch_fuse_tail(const FrameBuffers& fb_tail, const FrameBuffers& fb_fuse, const FrameIO& io, const rc_params_dev& prm, const BodyConst* body, int B,
                         const WaveTail& wt, hipStream_t s, hipEvent_t stop);   // fuse + tail of a tick in one launch (false: not available, launch them apart)
void rc_launch_prep_wave(const FrameBuffers& slot, const FrameIO& io0, const rc_params_dev& prm, int B, const WavePrep& w, hipStream_t s);
void rc_launch_reset(const FrameBuffers& fb, float* const* h, float* const* c, const int* hidden, const unsigned char* mask,
                     int B, hipStream_t s);

void rc_launch_flush_flags(const FrameBuffers& fb, int B, hipStream_t s);
void rc_launch_pack_rows(const float* src, int src_ld, int cols, float* dst, int ld, int B, hipStream_t s);

void rc_launch_r6d(const float* r6d, float* R, long long n, hipStream_t s);
struct CamConst { float Kinv[9]; float R[9]; };
void rc_launch_camera_inputs(const float* kp, const float* acc, const float* ori, const CamConst& cam, float* j2dc, float* accc,
                             float* oric, long long n, hipStream_t s);
void rc_launch_shape_body(const float* vt, const float* sd, const float* beta, const float* Jr, int V, float* v, float* j, hipStream_t s);
void rc_launch_fk_r(const BodyConst* body, const float* Rl, float* Rg, long long n, hipStream_t s);
void rc_launch_bone_to_joint(const BodyConst* body, const float* bone, float* joint, long long n, hipStream_t s);
void rc_launch_joint_to_bone(const BodyConst* body, const float* joint, float* bone, long long n, hipStream_t s);
void rc_launch_zero_pose(const BodyConst* body, const float* vt, int V, float* joint, float* vert, hipStream_t s);
void rc_launch_rotmat_to_r6d(const float* R, float* r6d, long long n, hipStream_t s);
void rc_launch_lerp(const float* a, const float* b, float w1, float w2, float* out, long long n, hipStream_t s);
void rc_launch_normalize_rows(const float* x, float* out, float* norm, long long rows, int width, hipStream_t s);
void rc_launch_angle_between(const float* R1, const float* R2, float* out, long long n, hipStream_t s);
void rc_launch_bbox_normalise(const float* kp, float* out, long long n, hipStream_t s);
void rc_launch_camera_inputs_rows(const float* kp, const float* acc, const float* ori, const int* seq_of_row, const int* len,
                                  const CamConst* cams, float sx, float sy, int n_rows, int Tmax, float* j2dc, float* accc,
                                  float* oric, hipStream_t s);
void rc_launch_aa2R(const float* aa, float* R, long long n, hipStream_t s);
void rc_launch_R2aa(const float* R, float* aa, long long n, hipStream_t s);
void rc_launch_ik(const BodyConst* body, const float* Rg, float* Rl, long long n, hipStream_t s);
void rc_launch_fk_bone(const BodyConst* body, const float* Rg, float* joints, long long n, hipStream_t s);
void rc_launch_body_fk(const BodyConst* body, const float* pose, const float* tran, float* grot, float* joint,
                       float* j33, long long n, hipStream_t s);
void rc_launch_body_mesh(const BodyConst* body, const float* vt, const float* w, int V, const float* pose, const float* tran,
                         float* vert, long long n, float* scratch, hipStream_t s);   // scratch: rc_body_mesh_scratch_floats(n)
long long rc_body_mesh_scratch_floats(long long n);
void rc_launch_residual(const BodyConst* body, const float* pose, const float* tran, const float* kp, const float* K,
                        float sigma, unsigned long long ign_mask, float* loss, long long T, hipStream_t s);
#define RC_IGN_DEFAULT 0x1800003FEull     // MediaPipe landmarks {1..9, 31, 32} (temporal_smplify.py:92); use_head: {31, 32}
// kM: regressor folded with the skinning data, [nk][24][4] (rc_api.cpp: fold_regressor), or nullptr (the SMPL joints stand in)
void rc_launch_mesh_metrics(const BodyConst* body, const float* vt, const float* w, int V, const float* kM, int nk, const float* pose_p,
                            const float* pose_t, float* out, long long n, float* scratch, hipStream_t s);
long long rc_mesh_metrics_scratch_floats(int V, long long n);
void rc_launch_imu_frames(const BodyConst* body, const float* vt, const float* w, const int* vid, const int* jid, const float* pose,
                          const float* tran, float* ori, float* joint, float* vert6, long long T, hipStream_t s);
void rc_launch_syn_acc(const float* v, float* acc, long long T, long long width, int n, hipStream_t s);
void rc_launch_procrustes(const float* S1, const float* S2, int nk, float* err, long long n, hipStream_t s);
void rc_launch_point_distance(const float* a, const float* b, float* d, long long n, hipStream_t s);

// smplify optimiser (rc_smplify.hip): one evaluation = loss terms + analytic gradient of all T frames
struct SmplifyArgs {
    const float* aa;        // [T,72] axis-angle (root first)
    const float* tran;      // [T,3]
    const float* kp;        // [T,33,3] pixels + confidence (ignored landmarks count as 0)
    const float* ref3d;     // [T,33,3] landmarks of the initial prediction
    const float* imu_aa;    // [T,18] axis-angle of the measured IMU orientations
    const float* means;     // [8,69]
    const float* prec;      // [8,69,72] precision matrices, rows padded (rc_smplify_set_prior)
    const float* prec_sym;  // [8,69,72] P + P^T, rows padded
    const float* lognll;    // [8] log(nll_weights)
    float* mj;              // [T,33,3] out (fwd) / in (grad)
    float* proj;            // [T,33,2]
    float* frame_loss;      // [T] reprojection + prior + angle + 3D of frame t
    float* imu_loss;        // [T] 0.25 * |aa(imu) - aa(G[ji])|^2
    float* smooth_loss;     // [T] smoothness terms of the pair (t-1, t)
    int* argmin;            // [T] mixture component of the prior            } written by the prior kernel,
    float* prior_ll;        // [T] min_m 0.5 d^T P_m d - log(nll_w_m)        } read by the forward / gradient kernels
    float* prior_g;         // [T,69] (P_m + P_m^T) d of that mixture        }
    float* fk;              // [T,2,24,9] local and global rotations of the forward kernel's primal, read by the gradient kernel
    float* grad_aa;         // [T,72]  } one flat vector [T*72 | T*3], the optimiser's parameter order
    float* grad_tran;       // [T,3]   } (temporal_smplify.py:141: [body_pose, tran])
    float K[9];
    unsigned long long ign_mask;   // landmarks whose confidence counts as zero (bit v)
    int T;
};
void rc_launch_smplify(const SmplifyArgs& A, const BodyConst* body, hipStream_t s);
// vector kernels of the device-resident L-BFGS (rc_smplify.hip)
#define RC_LBFGS_MAX_PAIRS 100             // history of the device-resident L-BFGS = torch.optim.LBFGS's default history_size
struct VecComb { const float* v[2 * RC_LBFGS_MAX_PAIRS + 1]; float c[2 * RC_LBFGS_MAX_PAIRS + 1]; int n_vec; };   // 2.4 KB kernel argument
struct VecJob { const float* a; const float* b; int op; int pad_; };     // op 0: sum a b, 1: max |a|, 2: sum |a|
// the same for all rows of a lock-step round of the batched optimiser (descriptor tables in device memory, the row from blockIdx.y)
struct VecOp { const float* a; const float* b; const float* c; float* out; float* out2; float t; int kind; long long n; };   // kind 0 axpy, 1 pair
struct VecCombRow { VecComb c; float* out; long long n; };
struct VecJobN { const float* a; const float* b; int op; int pad_; long long n; };
void rc_launch_smplify_rows(const SmplifyArgs* rows_dev, int n_rows, int T_max, const BodyConst* body, hipStream_t s);
void rc_launch_smplify_totals(const SmplifyArgs* rows_dev, int n_rows, double* out_dev, hipStream_t s);   // total loss per table entry, the host's summation order
// set-up and wrap-up of a batch of rows (rc_smplify_run_batch) in one launch each: what rc_smplify_run does per row with
// rc_launch_residual / rc_launch_R2aa / rc_launch_body_fk / rc_launch_aa2R and two device-to-device copies
struct SmplifyRowIO {
    const float* pose;      // [T,24,3,3] initial local rotations
    const float* tran;      // [T,3]
    const float* kp;        // [T,33,3]
    const float* imu_ori;   // [T,6,3,3]
    float* x;               // [T*75] optimiser parameters [axis-angle | translation]
    float* imu_aa;          // [T,18]
    float* joint;           // [T,24,3]
    float* ref3d;           // [T,33,3]
    float* res0;            // [T,33] residual before
    float* res1;            // [T,33] residual after
    float* pose_out;        // [T,24,3,3]
    float* tran_out;        // [T,3]
    float K[9];
    int T;
    int live;               // wrap-up: 0 = the pre-check rejected the row (nothing to do)
};
void rc_launch_smplify_begin_rows(const SmplifyRowIO* rows_dev, int n_rows, int T_max, const BodyConst* body, float sigma,
                                  unsigned long long ign_mask, hipStream_t s);
void rc_launch_smplify_end_rows(const SmplifyRowIO* rows_dev, int n_rows, int T_max, const BodyConst* body, float sigma,
                                unsigned long long ign_mask, hipStream_t s);
void rc_launch_vec_ops(const VecOp* ops_dev, int n_ops, long long n_max, hipStream_t s);
void rc_launch_vec_comb_rows(const VecCombRow* rows_dev, int n_rows, long long n_max, hipStream_t s);
void rc_launch_vec_dots_rows(const VecJobN* jobs_dev, int n_jobs, int nb_max, double* partial, hipStream_t s);
void rc_launch_vec_axpy(const float* x, const float* d, float t, float* out, long long n, hipStream_t s);
void rc_launch_vec_pair(const float* g_new, const float* g_old, const float* d, float t, float* y, float* sv, long long n, hipStream_t s);
void rc_launch_vec_comb(const VecComb& c, float* out, long long n, hipStream_t s);
void rc_launch_vec_dots(const VecJob* jobs_dev, int n_jobs, long long n, double* partial, hipStream_t s);

// narrow view of the context for rc_smplify_api.cpp (the struct itself lives in rc_api.cpp)
struct rc_ctx;
struct SmplifyState;
const BodyConst* rc_ctx_body(rc_ctx* ctx);                 // nullptr until rc_set_body
int rc_ctx_fail(rc_ctx* ctx, int code, const char* msg);   // records the message, returns code
SmplifyState*& rc_ctx_smplify(rc_ctx* ctx);
unsigned long long rc_ctx_ign_mask(rc_ctx* ctx);
void rc_smplify_free(SmplifyState* s);
