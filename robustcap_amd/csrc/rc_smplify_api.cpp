// Host side of the smplify optimiser: work space, closure evaluation, L-BFGS driver, C ABI.
//
// Reference: net/smplify/run.py:6-34 (smplify_runner: pre-check, optimise, per-frame update mask) and
// net/smplify/temporal_smplify.py:97-196 (parameters, closure, torch.optim.LBFGS). One closure evaluation is
// H2D of the parameter vector (300 B/frame), two kernels (rc_smplify.hip), D2H of the gradient and the per-frame loss
// terms (312 B/frame); the L-BFGS vectors stay on the host -- at 20 iterations x 75 T floats they are noise next to
// the 26 kernel evaluations.
#include "../../include/robustcap_hip.h"
#include "rc_internal.h"
#include "rc_lbfgs.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <memory>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

constexpr size_t kPriorLd = 72, kPriorMat = 8 * 69 * kPriorLd;   // rc_smplify.hip: SM_PLD

struct SmplifyState {
    float *means = nullptr, *prec = nullptr, *lognll = nullptr;     // device; prec = [2][8][69][72]: P, then P + P^T, rows padded
    bool have_prior = false;
    int64_t cap = 0;                                                  // frames the buffers below hold
    float *x = nullptr, *grad = nullptr;                              // [cap*75] flat [aa | tran]
    float *ref3d = nullptr, *imu_aa = nullptr, *mj = nullptr, *proj = nullptr, *joint = nullptr;
    float *terms = nullptr;                                           // [3*cap] frame | imu | smooth losses
    int* argmin = nullptr;
    float *prior_ll = nullptr, *prior_g = nullptr;                    // [cap], [cap*69]: outputs of the prior kernel
    float* fk = nullptr;                                              // [cap*432]: rotations of the forward kernel's primal
    float *res0 = nullptr, *res1 = nullptr, *Kd = nullptr;            // residuals [cap,33] before / after, K on the device
    float *h_x = nullptr, *h_grad = nullptr, *h_terms = nullptr, *h_res = nullptr;   // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double device_ms = 0.0;
    // device-resident L-BFGS (minimize_on_device): vectors of the optimiser, job table and partial sums of the inner products
    int64_t lb_cap = 0;                                               // frames the buffers below hold
    int lb_pairs = 0;                                                 // curvature pairs they hold
    float *xt = nullptr, *dir = nullptr, *gslot = nullptr, *Sv = nullptr, *Yv = nullptr;   // [n], [n], [6][n], [pairs][n] x 2
    VecJob *jobs_d = nullptr, *jobs_h = nullptr;                      // device / pinned
    double *part_d = nullptr, *part_h = nullptr;
    const float* ref3d_override = nullptr;                            // rc_smplify_set_ref3d: caller-owned [T,33,3], next run only
    // arenas of rc_smplify_run_batch (grow-only: a second evaluation of the same size allocates nothing)
    char *batch_dev = nullptr, *batch_pin = nullptr;
    size_t batch_dev_cap = 0, batch_pin_cap = 0;
};

namespace {

typedef float (*cubic_fn)(float, float, float, float, float, float, bool, float, float);

#define SM_TRY(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return rc_ctx_fail(ctx, RC_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

void free_work(SmplifyState* s) {
    for (float** p : {&s->x, &s->grad, &s->ref3d, &s->imu_aa, &s->mj, &s->proj, &s->joint, &s->terms, &s->res0, &s->res1, &s->prior_ll, &s->prior_g, &s->fk})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (s->argmin) { (void)hipFree(s->argmin); s->argmin = nullptr; }
    for (float** p : {&s->h_x, &s->h_grad, &s->h_terms, &s->h_res})
        if (*p) { (void)hipHostFree(*p); *p = nullptr; }
    s->cap = 0;
    for (float** p : {&s->xt, &s->dir, &s->gslot, &s->Sv, &s->Yv})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (s->jobs_d) { (void)hipFree(s->jobs_d); s->jobs_d = nullptr; }
    if (s->part_d) { (void)hipFree(s->part_d); s->part_d = nullptr; }
    if (s->jobs_h) { (void)hipHostFree(s->jobs_h); s->jobs_h = nullptr; }
    if (s->part_h) { (void)hipHostFree(s->part_h); s->part_h = nullptr; }
    s->lb_cap = 0; s->lb_pairs = 0;
}

int state_of(rc_ctx* ctx, SmplifyState** out) {
    SmplifyState*& s = rc_ctx_smplify(ctx);
    if (!s) {
        s = new SmplifyState();
        SM_TRY(ctx, hipMalloc((void**)&s->means, 8 * 69 * sizeof(float)));
        SM_TRY(ctx, hipMalloc((void**)&s->prec, 2 * kPriorMat * sizeof(float)));
        SM_TRY(ctx, hipMalloc((void**)&s->lognll, 8 * sizeof(float)));
        SM_TRY(ctx, hipMalloc((void**)&s->Kd, 9 * sizeof(float)));
        SM_TRY(ctx, hipEventCreate(&s->ev0));
        SM_TRY(ctx, hipEventCreate(&s->ev1));
    }
    *out = s;
    return RC_OK;
}

int reserve(rc_ctx* ctx, SmplifyState* s, int64_t T) {
    if (T <= s->cap) return RC_OK;
    free_work(s);
    const size_t n = (size_t)T;
    SM_TRY(ctx, hipMalloc((void**)&s->x, n * 75 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->grad, n * 75 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->ref3d, n * 99 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->imu_aa, n * 18 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->mj, n * 99 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->proj, n * 66 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->joint, n * 72 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->terms, n * 3 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->argmin, n * sizeof(int)));
    SM_TRY(ctx, hipMalloc((void**)&s->prior_ll, n * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->prior_g, n * 69 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->fk, n * 432 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->res0, n * 33 * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->res1, n * 33 * sizeof(float)));
    SM_TRY(ctx, hipHostMalloc((void**)&s->h_x, n * 75 * sizeof(float)));
    SM_TRY(ctx, hipHostMalloc((void**)&s->h_grad, n * 75 * sizeof(float)));
    SM_TRY(ctx, hipHostMalloc((void**)&s->h_terms, n * 3 * sizeof(float)));
    SM_TRY(ctx, hipHostMalloc((void**)&s->h_res, n * 66 * sizeof(float)));
    s->cap = T;
    return RC_OK;
}

SmplifyArgs make_args(SmplifyState* s, const float* x, const float* kp, const float* ref3d, const float* imu_aa, const float* K,
                      float* grad, int64_t T, unsigned long long ign_mask) {
    SmplifyArgs A{};
    A.aa = x; A.tran = x + T * 72;
    A.kp = kp; A.ref3d = ref3d; A.imu_aa = imu_aa;
    A.means = s->means; A.prec = s->prec; A.prec_sym = s->prec + kPriorMat; A.lognll = s->lognll;
    A.mj = s->mj; A.proj = s->proj;
    A.frame_loss = s->terms; A.imu_loss = s->terms + T; A.smooth_loss = s->terms + 2 * T;
    A.argmin = s->argmin; A.prior_ll = s->prior_ll; A.prior_g = s->prior_g; A.fk = s->fk;
    A.grad_aa = grad; A.grad_tran = grad + T * 72;
    for (int q = 0; q < 9; ++q) A.K[q] = K[q];
    A.ign_mask = ign_mask;
    A.T = (int)T;
    return A;
}

// total loss from the per-frame terms (losses.py:57-87): the IMU term is summed over the sequence and then added to
// EVERY frame before the final sum, i.e. it counts T times.
double total_loss(const float* terms, int64_t T) {
    double f = 0.0, imu = 0.0, sm = 0.0;
    for (int64_t t = 0; t < T; ++t) { f += terms[t]; imu += terms[T + t]; sm += terms[2 * T + t]; }
    return f + (double)T * imu + sm;
}

// ---------------------------------------------------------------------------------- L-BFGS with device-resident vectors
// The same algorithm object as rc_lbfgs.h (torch.optim.LBFGS(max_iter, strong_wolfe), temporal_smplify.py:141-147) with the
// vectors left on the device. Per closure evaluation the host reads back the per-frame loss terms and four inner products
// (g.d, |g|_inf, |g|_1, |d|_inf: one kernel of partial sums); per iteration the inner products of the new curvature pair and of
// the gradient with the pairs kept so far, from which the two-loop recursion runs in COEFFICIENT space -- the direction is
// d = sum_i a_i s_i + b_i y_i + c g and only its (2m + 1) coefficients are computed on the host (float64), then one kernel
// forms d. No parameter or gradient vector crosses PCIe; round 2 moved 2 x 180 KB per evaluation and ran the recursion over
// 45,000-element host vectors (10 of the 13.8 ms per 600-frame row).
const int kSlots = 6, kSlotJobs = 4;

// RC_LBFGS_HISTORY: curvature pairs kept (default: torch's history_size, 100); read per call, by both formulations
int lbfgs_history_env() {
    const char* v = std::getenv("RC_LBFGS_HISTORY");
    const int h = v && *v ? std::atoi(v) : 100;
    return h < 1 ? 1 : h;
}

int reserve_lbfgs(rc_ctx* ctx, SmplifyState* s, int64_t T, int pairs) {
    if (T <= s->lb_cap && pairs <= s->lb_pairs) return RC_OK;
    for (float** p : {&s->xt, &s->dir, &s->gslot, &s->Sv, &s->Yv})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (s->jobs_d) { (void)hipFree(s->jobs_d); s->jobs_d = nullptr; }
    if (s->part_d) { (void)hipFree(s->part_d); s->part_d = nullptr; }
    if (s->jobs_h) { (void)hipHostFree(s->jobs_h); s->jobs_h = nullptr; }
    if (s->part_h) { (void)hipHostFree(s->part_h); s->part_h = nullptr; }
    s->lb_cap = 0; s->lb_pairs = 0;
    const size_t n = (size_t)T * 75, nb = (n + 4095) / 4096;
    const size_t max_jobs = (size_t)kSlots * kSlotJobs + 6 * (size_t)pairs + 8;
    SM_TRY(ctx, hipMalloc((void**)&s->xt, n * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->dir, n * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->gslot, kSlots * n * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->Sv, (size_t)pairs * n * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->Yv, (size_t)pairs * n * sizeof(float)));
    SM_TRY(ctx, hipMalloc((void**)&s->jobs_d, max_jobs * sizeof(VecJob)));
    SM_TRY(ctx, hipMalloc((void**)&s->part_d, max_jobs * nb * sizeof(double)));
    SM_TRY(ctx, hipHostMalloc((void**)&s->jobs_h, max_jobs * sizeof(VecJob)));
    SM_TRY(ctx, hipHostMalloc((void**)&s->part_h, max_jobs * nb * sizeof(double)));
    s->lb_cap = T; s->lb_pairs = pairs;
    return RC_OK;
}

struct DevResult { int n_iter = 0, n_eval = 0; float first_loss = 0, loss = 0; };

// x (device, s->x) is updated in place. kp / K / ref3d / imu_aa as for the closure.
int minimize_on_device(rc_ctx* ctx, SmplifyState* s, const BodyConst* body, const float* kp, const float* K, int64_t T, float lr,
                       int max_iter, hipStream_t st, DevResult& res) {
    const size_t n = (size_t)T * 75;
    const int nb = (int)((n + 4095) / 4096);
    // History: the last M curvature pairs (torch.optim.LBFGS: history_size = 100, old_dirs.pop(0) when full). M + 1 physical slots:
    // the candidate pair of an iteration goes to the spare one, so that rejecting it (y.s <= 1e-10) leaves the oldest pair alone;
    // accepting it into a full history makes the oldest pair's slot the next spare. RC_LBFGS_HISTORY shrinks M (A/B tests).
    const int M = std::max(1, std::min(std::min(max_iter, RC_LBFGS_MAX_PAIRS), lbfgs_history_env()));
    const int P = M + 1;
    if (int rc = reserve_lbfgs(ctx, s, T, P)) return rc;
    const int max_eval = max_iter * 5 / 4;
    const float tolerance_grad = 1e-7f, tolerance_change = 1e-9f;
    const unsigned long long ign = rc_ctx_ign_mask(ctx);
    auto slot = [&](int k) { return s->gslot + (size_t)k * n; };
    // fixed part of the job table: per gradient slot {g.d, max |g|, sum |g|, max |d|}
    for (int k = 0; k < kSlots; ++k) {
        s->jobs_h[k * kSlotJobs + 0] = VecJob{slot(k), s->dir, 0, 0};
        s->jobs_h[k * kSlotJobs + 1] = VecJob{slot(k), nullptr, 1, 0};
        s->jobs_h[k * kSlotJobs + 2] = VecJob{slot(k), nullptr, 2, 0};
        s->jobs_h[k * kSlotJobs + 3] = VecJob{s->dir, nullptr, 1, 0};
    }
    const int var0 = kSlots * kSlotJobs;                                // first job of the per-iteration (Gram) part
    SM_TRY(ctx, hipMemcpyAsync(s->jobs_d, s->jobs_h, (size_t)var0 * sizeof(VecJob), hipMemcpyHostToDevice, st));
    auto job_sum = [&](int job, bool is_max) {
        const double* p = s->part_h + (size_t)job * nb;
        double r = p[0];
        for (int b = 1; b < nb; ++b) r = is_max ? std::max(r, p[b]) : r + p[b];
        return r;
    };
    struct Eval { float f, gtd, gmax, gsum, dmax; };
    hipError_t herr = hipSuccess;
    // closure at x + t d into gradient slot k (t == 0: at x itself; the direction-dependent products are then meaningless)
    auto eval = [&](float t, int k, Eval& e) -> bool {
        const float* xp = s->x;
        if (t != 0.0f) { rc_launch_vec_axpy(s->x, s->dir, t, s->xt, (long long)n, st); xp = s->xt; }
        const SmplifyArgs A = make_args(s, xp, kp, s->ref3d, s->imu_aa, K, slot(k), T, ign);
        herr = hipEventRecord(s->ev0, st);
        rc_launch_smplify(A, body, st);
        if (herr == hipSuccess) herr = hipEventRecord(s->ev1, st);
        rc_launch_vec_dots(s->jobs_d + k * kSlotJobs, kSlotJobs, (long long)n, s->part_d + (size_t)k * kSlotJobs * nb, st);
        if (herr == hipSuccess) herr = hipMemcpyAsync(s->part_h + (size_t)k * kSlotJobs * nb, s->part_d + (size_t)k * kSlotJobs * nb,
                                                      (size_t)kSlotJobs * nb * sizeof(double), hipMemcpyDeviceToHost, st);
        if (herr == hipSuccess) herr = hipMemcpyAsync(s->h_terms, s->terms, (size_t)T * 3 * sizeof(float), hipMemcpyDeviceToHost, st);
        if (herr == hipSuccess) herr = hipStreamSynchronize(st);
        if (herr == hipSuccess) herr = hipGetLastError();
        if (herr != hipSuccess) return false;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->device_ms += ms;
        e.f = (float)total_loss(s->h_terms, T);
        e.gtd = (float)job_sum(k * kSlotJobs + 0, false);
        e.gmax = (float)job_sum(k * kSlotJobs + 1, true);
        e.gsum = (float)job_sum(k * kSlotJobs + 2, false);
        e.dmax = (float)job_sum(k * kSlotJobs + 3, true);
        return true;
    };
#define LB_FAIL() return rc_ctx_fail(ctx, RC_ERR_HIP, (std::string("smplify L-BFGS: ") + hipGetErrorString(herr)).c_str())

    int base = 0;                                                       // slot of the gradient at x
    Eval e0;
    if (!eval(0.0f, base, e0)) LB_FAIL();
    float loss = e0.f, gmax = e0.gmax, gsum = e0.gsum;
    res.first_loss = res.loss = loss;
    int evals = 1;
    if (gmax <= tolerance_grad) { res.n_eval = evals; return RC_OK; }

    // inner products among the basis {s_0.., y_0.., g} by PHYSICAL slot: index i -> s_i, P + i -> y_i, 2P -> g
    const int NB = 2 * P + 1, IG = 2 * P;
    std::vector<double> G((size_t)NB * NB, 0.0), ro(P, 0.0), al(P, 0.0), delta(NB, 0.0);
    auto Gat = [&](int a, int b) -> double& { return G[(size_t)a * NB + b]; };
    std::vector<int> order;                                             // physical slots of the pairs kept, oldest first
    int spare = 0;                                                      // physical slot of the next candidate
    double H_diag = 1.0;
    float t = 0.0f, prev_loss = loss, gtd = 0.0f, d_max = 0.0f;
    int prev_base = base;
    int n_iter = 0;
    cubic_fn cubic = rc::Lbfgs<float>::cubic;
    while (n_iter < max_iter) {
        ++n_iter;
        VecComb comb{};
        if (n_iter == 1) {
            comb.n_vec = 1; comb.v[0] = slot(base); comb.c[0] = -1.0f;     // d = -g
        } else {
            // candidate pair (slot `spare`): y = g - prev_g, s = t d
            const int c = spare;
            rc_launch_vec_pair(slot(base), slot(prev_base), s->dir, t, s->Yv + (size_t)c * n, s->Sv + (size_t)c * n, (long long)n, st);
            // inner products: the candidate against the pairs kept (and itself), the gradient against all of them
            struct Want { int a, b; };
            std::vector<Want> want;
            std::vector<int> with_c(order);
            with_c.push_back(c);
            for (int j : with_c) { want.push_back({c, j}); want.push_back({c, P + j}); want.push_back({P + c, P + j}); }
            for (int j : order) want.push_back({P + c, j});
            for (int j : with_c) { want.push_back({IG, j}); want.push_back({IG, P + j}); }
            want.push_back({IG, IG});
            auto vec_of = [&](int idx) -> const float* {
                return idx == IG ? slot(base) : (idx >= P ? s->Yv + (size_t)(idx - P) * n : s->Sv + (size_t)idx * n);
            };
            for (size_t q = 0; q < want.size(); ++q) s->jobs_h[var0 + q] = VecJob{vec_of(want[q].a), vec_of(want[q].b), 0, 0};
            herr = hipMemcpyAsync(s->jobs_d + var0, s->jobs_h + var0, want.size() * sizeof(VecJob), hipMemcpyHostToDevice, st);
            rc_launch_vec_dots(s->jobs_d + var0, (int)want.size(), (long long)n, s->part_d + (size_t)var0 * nb, st);
            if (herr == hipSuccess) herr = hipMemcpyAsync(s->part_h + (size_t)var0 * nb, s->part_d + (size_t)var0 * nb,
                                                          want.size() * nb * sizeof(double), hipMemcpyDeviceToHost, st);
            if (herr == hipSuccess) herr = hipStreamSynchronize(st);
            if (herr != hipSuccess) LB_FAIL();
            for (size_t q = 0; q < want.size(); ++q) {
                const double v = job_sum(var0 + (int)q, false);
                Gat(want[q].a, want[q].b) = v; Gat(want[q].b, want[q].a) = v;
            }
            {
                const double ys = Gat(c, P + c);
                if ((float)ys > 1e-10f) {                                 // keep the pair (torch: ys > 1e-10)
                    H_diag = ys / Gat(P + c, P + c);
                    ro[c] = 1.0 / ys;
                    if ((int)order.size() == M) {                         // history full: the oldest pair goes (old_dirs.pop(0)),
                        spare = order.front();                            // its slot takes the next candidate
                        order.erase(order.begin());
                        order.push_back(c);
                    } else {
                        order.push_back(c);
                        spare = (int)order.size();                        // slots are handed out in order until the history is full
                    }
                }
            }
            const int m = (int)order.size();
            // two-loop recursion on coefficients: q = sum_k delta[k] basis[k], starting from q = -g
            std::fill(delta.begin(), delta.end(), 0.0);
            delta[IG] = -1.0;
            auto dot_q = [&](int idx) {
                double r = delta[IG] * Gat(IG, idx);
                for (int j : order) r += delta[j] * Gat(j, idx) + delta[P + j] * Gat(P + j, idx);
                return r;
            };
            for (int i = m - 1; i >= 0; --i) { const int j = order[i]; al[j] = dot_q(j) * ro[j]; delta[P + j] -= al[j]; }
            for (double& v : delta) v *= H_diag;
            for (int i = 0; i < m; ++i) { const int j = order[i]; const double be = dot_q(P + j) * ro[j]; delta[j] += al[j] - be; }
            comb.n_vec = 0;
            for (int j : order) {
                comb.v[comb.n_vec] = s->Sv + (size_t)j * n; comb.c[comb.n_vec++] = (float)delta[j];
                comb.v[comb.n_vec] = s->Yv + (size_t)j * n; comb.c[comb.n_vec++] = (float)delta[P + j];
            }
            comb.v[comb.n_vec] = slot(base); comb.c[comb.n_vec++] = (float)delta[IG];
            gtd = (float)dot_q(IG);
        }
        rc_launch_vec_comb(comb, s->dir, (long long)n, st);
        prev_base = base;
        prev_loss = loss;
        if (n_iter == 1) {
            t = std::min(1.0f, 1.0f / gsum) * lr;
            // g.d of d = -g: the base slot's table entry pairs it with d (one more small readback, once per run)
            rc_launch_vec_dots(s->jobs_d + base * kSlotJobs, 1, (long long)n, s->part_d + (size_t)base * kSlotJobs * nb, st);
            herr = hipMemcpyAsync(s->part_h + (size_t)base * kSlotJobs * nb, s->part_d + (size_t)base * kSlotJobs * nb, (size_t)nb * sizeof(double),
                                  hipMemcpyDeviceToHost, st);
            if (herr == hipSuccess) herr = hipStreamSynchronize(st);
            if (herr != hipSuccess) LB_FAIL();
            gtd = (float)job_sum(base * kSlotJobs, false);
        } else t = lr;
        if (gtd > -tolerance_change) break;

        // ---- strong-Wolfe line search (rc_lbfgs.h: strong_wolfe) on gradient slots ------------------------------------
        struct Pt { float t, f, gtd; int k; };
        const int max_ls = max_eval - evals;
        auto free_slot = [&](std::initializer_list<int> live) {
            for (int k = 0; k < kSlots; ++k) { bool used = false; for (int v : live) used = used || v == k; if (!used) return k; }
            return -1;
        };
        const float c1 = 1e-4f, c2 = 0.9f;
        Pt cur{t, 0, 0, free_slot({base})};
        Eval ev;
        if (!eval(cur.t, cur.k, ev)) LB_FAIL();
        int ls_evals = 1;
        cur.f = ev.f; cur.gtd = ev.gtd; d_max = ev.dmax;
        Pt prev{0, loss, gtd, base};
        Pt br[2] = {prev, prev};
        int n_br = 0, ls_iter = 0;
        bool done = false;
        while (ls_iter < max_ls) {
            if (cur.f > (loss + c1 * cur.t * gtd) || (ls_iter > 1 && cur.f >= prev.f)) { br[0] = prev; br[1] = cur; n_br = 2; break; }
            if (std::fabs(cur.gtd) <= -c2 * gtd) { br[0] = cur; n_br = 1; done = true; break; }
            if (cur.gtd >= 0) { br[0] = prev; br[1] = cur; n_br = 2; break; }
            const float min_step = cur.t + 0.01f * (cur.t - prev.t), max_step = cur.t * 10;
            const float tn = cubic(prev.t, prev.f, prev.gtd, cur.t, cur.f, cur.gtd, true, min_step, max_step);
            prev = cur;
            cur.t = tn;
            cur.k = free_slot({base, prev.k});
            if (!eval(tn, cur.k, ev)) LB_FAIL();
            cur.f = ev.f; cur.gtd = ev.gtd;
            ++ls_evals;
            ++ls_iter;
        }
        if (ls_iter == max_ls) { br[0] = Pt{0, loss, gtd, base}; br[1] = cur; n_br = 2; }
        bool insuf = false;
        int lo = 0, hi = 1;
        if (n_br == 2 && !(br[0].f <= br[1].f)) { lo = 1; hi = 0; }
        while (!done && ls_iter < max_ls) {
            if (std::fabs(br[1].t - br[0].t) * d_max < tolerance_change) break;
            float tn = cubic(br[0].t, br[0].f, br[0].gtd, br[1].t, br[1].f, br[1].gtd, false, 0, 0);
            const float bmax = std::max(br[0].t, br[1].t), bmin = std::min(br[0].t, br[1].t);
            const float eps = 0.1f * (bmax - bmin);
            if (std::min(bmax - tn, tn - bmin) < eps) {
                if (insuf || tn >= bmax || tn <= bmin) {
                    tn = (std::fabs(tn - bmax) < std::fabs(tn - bmin)) ? bmax - eps : bmin + eps;
                    insuf = false;
                } else insuf = true;
            } else insuf = false;
            cur.t = tn;
            cur.k = free_slot({base, br[0].k, br[1].k});
            if (!eval(tn, cur.k, ev)) LB_FAIL();
            cur.f = ev.f; cur.gtd = ev.gtd;
            ++ls_evals;
            ++ls_iter;
            if (cur.f > (loss + c1 * tn * gtd) || cur.f >= br[lo].f) {
                br[hi] = cur;
                if (br[0].f <= br[1].f) { lo = 0; hi = 1; } else { lo = 1; hi = 0; }
            } else {
                if (std::fabs(cur.gtd) <= -c2 * gtd) done = true;
                else if (cur.gtd * (br[hi].t - br[lo].t) >= 0) br[hi] = br[lo];
                br[lo] = cur;
            }
        }
        if (n_br == 1) lo = 0;
        t = br[lo].t;
        loss = br[lo].f;
        base = br[lo].k;                                                  // gradient at the accepted point
        rc_launch_vec_axpy(s->x, s->dir, t, s->x, (long long)n, st);      // x += t d (element-wise, in place)
        // |g|_inf of the accepted point: its slot's partial sums are still in the pinned table unless the slot is the old base
        gmax = base == prev_base ? gmax : (float)job_sum(base * kSlotJobs + 1, true);
        const bool opt_cond = gmax <= tolerance_grad;
        evals += ls_evals;

        if (n_iter == max_iter) break;
        if (evals >= max_eval) break;
        if (opt_cond) break;
        if (d_max * std::fabs(t) <= tolerance_change) break;
        if (std::fabs(loss - prev_loss) < tolerance_change) break;
    }
#undef LB_FAIL
    res.n_iter = n_iter;
    res.n_eval = evals;
    res.loss = loss;
    return RC_OK;
}


// ================================================================== lock-step batch of rows (rc_smplify_run_batch, round 4)
// evaluate.py:86-90 refines the (sequence, camera) rows of an evaluation one after another; they are independent optimisation
// problems. Round 3 drove them from host threads, one context and stream each: ~300 HIP calls per row contend for the runtime,
// 72 rows of 600 frames took 0.19-0.23 s however many threads ran. Here every row's optimiser (the algorithm of
// minimize_on_device, unchanged, as a fiber of the caller's thread -- a stack of its own, see RowBatch::fibers) hands its next device request -- "evaluate the closure at x + t d
// into gradient slot k", or "form the curvature pair and these inner products" -- to ONE executor; when every live row has asked,
// the executor runs all requests with one launch per kind over all rows (descriptor tables in device memory, the row from
// blockIdx.y), one read-back, one synchronisation, and resumes the rows. A round costs what its largest kernel costs; per row the
// arithmetic is the same chain of operations as alone, so n_iter / n_eval / losses are those of the one-row-at-a-time run.
struct RowBuf {                       // device vectors of one row (carved from the batch's arena)
    int64_t T = 0;
    size_t n = 0;                     // 75 T
    int nb = 0;                       // 4,096-element blocks of a vector
    float *x = nullptr, *xt = nullptr, *dir = nullptr, *gslot = nullptr, *Sv = nullptr, *Yv = nullptr;
    float *ref3d = nullptr, *imu_aa = nullptr, *mj = nullptr, *proj = nullptr, *joint = nullptr, *res0 = nullptr, *res1 = nullptr;
    float* terms = nullptr;           // [3 T] frame | imu | smooth, device (inside the terms arena)
    const float* terms_h = nullptr;   // the same region of the pinned copy
    int* argmin = nullptr;
    float *prior_ll = nullptr, *prior_g = nullptr, *fk = nullptr;
    float *h_res0 = nullptr, *h_res1 = nullptr;   // pinned [33 T] each: residual before / after
    float K[9];
    const float* kp = nullptr;
};

struct RowReq {
    int kind = 0;                     // 1: closure evaluation, 2: curvature pair + inner products, 3: pending update of x only
    bool has_comb = false;            // dir = combination, in front of everything else (start of a line search)
    VecComb comb{};
    bool accept = false;              // x += accept_t * dir (the step the previous line search accepted)
    float accept_t = 0.0f;
    float t = 0.0f;                   // kind 1: evaluate at x + t * dir into gradient slot k
    int k = 0;
    const float *g_new = nullptr, *g_old = nullptr;   // kind 2: y = g_new - g_old -> yv, s = pair_t * dir -> sv
    float *yv = nullptr, *sv = nullptr;
    float pair_t = 0.0f;
    std::vector<VecJobN> jobs;        // inner products wanted (kind 1: the slot's five, kind 2: the Gram entries)
    int job0 = 0;                     // filled by the executor: first job of this row in the round's table
    int arg0 = -1;                    // ... and, kind 1, the row's entry of the round's closure table (its total loss comes back there)
};

class RowBatch {
  public:
    rc_ctx* ctx = nullptr;
    const BodyConst* body = nullptr;
    SmplifyState* prior = nullptr;
    hipStream_t st = nullptr;
    unsigned long long ign = 0;
    std::vector<RowBuf> row;
    int T_max = 0, nb_max = 0, max_jobs = 0;
    size_t n_max = 0;
    // arenas
    char* dev = nullptr; size_t dev_bytes = 0;
    char* pin = nullptr; size_t pin_bytes = 0;
    SmplifyArgs *args_d = nullptr, *args_h = nullptr;
    VecOp *ops_d = nullptr, *ops_h = nullptr;
    VecCombRow *comb_d = nullptr, *comb_h = nullptr;
    VecJobN *jobs_d = nullptr, *jobs_h = nullptr;
    double *part_d = nullptr, *part_h = nullptr;
    float *terms_d = nullptr, *terms_h = nullptr;
    double *total_d = nullptr, *total_h = nullptr;                       // total loss per closure-table entry of the round
    bool device_totals = true;                                           // RC_SMPLIFY_DEVICE_TOTALS=0: read the per-frame terms back and add them on the host (A/B)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double device_ms = 0.0, round_ms = 0.0;                              // closure kernels (events) / wall time inside run_round
    int rounds = 0;
    // rendezvous
    std::mutex mu;
    std::condition_variable cv;
    std::vector<RowReq*> pending;
    int n_live = 0, n_wait = 0;
    unsigned long long gen = 0;
    hipError_t herr = hipSuccess;
    // fibers (the default): every row's optimiser runs on a stack of its own inside the CALLER's thread; submit() switches back to the
    // scheduler in rc_smplify_run_batch, which runs the round when every live row has asked and resumes them one after another. No
    // thread is created, woken or left spinning (72 condition-variable wake-ups cost 0.27 ms per round, 8.7 of config 3's 40 ms).
    bool fibers = false;
    ucontext_t sched{};
    std::vector<ucontext_t> fctx;

    ~RowBatch() {                                                         // (the arenas belong to the context's SmplifyState)
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
    double job_sum(const RowReq& q, int j, int nb, bool is_max) const {
        const double* p = part_h + (size_t)(q.job0 + j) * nb_max;
        double r = p[0];
        for (int b = 1; b < nb; ++b) r = is_max ? std::max(r, p[b]) : r + p[b];
        return r;
    }
    // hand in the row's request and sleep until the round it belongs to has run; false = a HIP call failed
    bool submit(int r, RowReq* q) {
        if (fibers) {
            pending[r] = q;
            ++n_wait;
            swapcontext(&fctx[r], &sched);                                 // back when the round has run
            return herr == hipSuccess;
        }
        std::unique_lock<std::mutex> lk(mu);
        pending[r] = q;
        ++n_wait;
        if (n_wait == n_live) run_round();
        else { const unsigned long long g = gen; cv.wait(lk, [&] { return gen != g; }); }
        return herr == hipSuccess;
    }
    void leave(int r) {
        if (fibers) { pending[r] = nullptr; --n_live; return; }
        std::unique_lock<std::mutex> lk(mu);
        pending[r] = nullptr;
        --n_live;
        if (n_live > 0 && n_wait == n_live) run_round();
    }

    void run_round() {                                                   // (mu held, or fibers: every live row is waiting)
        const auto t_round = std::chrono::steady_clock::now();
        int n_args = 0, n_ops = 0, n_comb = 0, n_jobs = 0;
        for (size_t r = 0; r < row.size(); ++r) {
            RowReq* q = pending[r];
            if (!q) continue;
            const RowBuf& b = row[r];
            if (q->has_comb) { comb_h[n_comb].c = q->comb; comb_h[n_comb].out = b.dir; comb_h[n_comb].n = (long long)b.n; ++n_comb; }
            if (q->accept) ops_h[n_ops++] = VecOp{b.x, b.dir, nullptr, b.x, nullptr, q->accept_t, 0, (long long)b.n};
            if (q->kind == 1) {
                const float* xp = b.x;
                if (q->t != 0.0f) { ops_h[n_ops++] = VecOp{b.x, b.dir, nullptr, b.xt, nullptr, q->t, 0, (long long)b.n}; xp = b.xt; }
                SmplifyArgs A{};
                A.aa = xp; A.tran = xp + b.T * 72;
                A.kp = b.kp; A.ref3d = b.ref3d; A.imu_aa = b.imu_aa;
                A.means = prior->means; A.prec = prior->prec; A.prec_sym = prior->prec + kPriorMat; A.lognll = prior->lognll;
                A.mj = b.mj; A.proj = b.proj;
                A.frame_loss = b.terms; A.imu_loss = b.terms + b.T; A.smooth_loss = b.terms + 2 * b.T;
                A.argmin = b.argmin; A.prior_ll = b.prior_ll; A.prior_g = b.prior_g; A.fk = b.fk;
                float* g = b.gslot + (size_t)q->k * b.n;
                A.grad_aa = g; A.grad_tran = g + b.T * 72;
                for (int e = 0; e < 9; ++e) A.K[e] = b.K[e];
                A.ign_mask = ign; A.T = (int)b.T;
                q->arg0 = n_args;
                args_h[n_args++] = A;
            } else if (q->kind == 2) {
                ops_h[n_ops++] = VecOp{q->g_new, q->g_old, b.dir, q->yv, q->sv, q->pair_t, 1, (long long)b.n};
            }
            q->job0 = n_jobs;
            for (const VecJobN& j : q->jobs) jobs_h[n_jobs++] = j;
        }
        auto up = [&](void* d, const void* h, size_t bytes) {
            if (bytes && herr == hipSuccess) herr = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
        };
        up(comb_d, comb_h, (size_t)n_comb * sizeof(VecCombRow));
        up(ops_d, ops_h, (size_t)n_ops * sizeof(VecOp));
        up(args_d, args_h, (size_t)n_args * sizeof(SmplifyArgs));
        up(jobs_d, jobs_h, (size_t)n_jobs * sizeof(VecJobN));
        // A row's accept / evaluation-point / pair operations read dir: the combination that WRITES dir goes first. Within the
        // op table a row's accept (x += t d) stands in front of whatever reads x.
        rc_launch_vec_comb_rows(comb_d, n_comb, (long long)n_max, st);
        // two op launches: updates of x first (the evaluation point x + t d of the same request reads the new x)
        {
            int n_acc = 0;
            for (int i = 0; i < n_ops; ++i) if (ops_h[i].kind == 0 && ops_h[i].out == ops_h[i].a) ++n_acc;
            if (n_acc > 0 && n_acc < n_ops) {                            // partition: in-place updates | the rest (stable)
                std::vector<VecOp> a, b;
                for (int i = 0; i < n_ops; ++i) ((ops_h[i].kind == 0 && ops_h[i].out == ops_h[i].a) ? a : b).push_back(ops_h[i]);
                std::copy(a.begin(), a.end(), ops_h);
                std::copy(b.begin(), b.end(), ops_h + a.size());
                up(ops_d, ops_h, (size_t)n_ops * sizeof(VecOp));
                rc_launch_vec_ops(ops_d, n_acc, (long long)n_max, st);
                rc_launch_vec_ops(ops_d + n_acc, n_ops - n_acc, (long long)n_max, st);
            } else rc_launch_vec_ops(ops_d, n_ops, (long long)n_max, st);
        }
        if (n_args > 0) {
            if (herr == hipSuccess) herr = hipEventRecord(ev0, st);
            rc_launch_smplify_rows(args_d, n_args, T_max, body, st);
            if (herr == hipSuccess) herr = hipEventRecord(ev1, st);
            if (device_totals) rc_launch_smplify_totals(args_d, n_args, total_d, st);
        }
        rc_launch_vec_dots_rows(jobs_d, n_jobs, nb_max, part_d, st);
        if (n_jobs > 0 && herr == hipSuccess)
            herr = hipMemcpyAsync(part_h, part_d, (size_t)n_jobs * nb_max * sizeof(double), hipMemcpyDeviceToHost, st);
        if (n_args > 0 && herr == hipSuccess)
            herr = device_totals ? hipMemcpyAsync(total_h, total_d, (size_t)n_args * sizeof(double), hipMemcpyDeviceToHost, st)
                                 : hipMemcpyAsync(terms_h, terms_d, row.size() * (size_t)3 * T_max * sizeof(float), hipMemcpyDeviceToHost, st);
        if (herr == hipSuccess) herr = hipStreamSynchronize(st);
        if (herr == hipSuccess) herr = hipGetLastError();
        if (n_args > 0 && herr == hipSuccess) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) device_ms += ms;
        }
        ++rounds;
        round_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count();
        n_wait = 0;
        ++gen;
        cv.notify_all();
    }
};

// minimize_on_device for row r of a batch: the same algorithm, the device work requested from the batch's executor
void lbfgs_row(RowBatch& B, const int r, const float lr, const int max_iter, DevResult& res, bool& ok) {
    ok = false;
    struct Leave { RowBatch& B; int r; ~Leave() { B.leave(r); } } on_exit{B, r};
    const RowBuf& b = B.row[r];
    const size_t n = b.n;
    const int nb = b.nb;
    const int M = std::max(1, std::min(std::min(max_iter, RC_LBFGS_MAX_PAIRS), lbfgs_history_env()));
    const int P = M + 1;
    const int max_eval = max_iter * 5 / 4;
    const float tolerance_grad = 1e-7f, tolerance_change = 1e-9f;
    auto slot = [&](int k) { return b.gslot + (size_t)k * n; };
    // per evaluation five inner products: g.d, max |g|, sum |g|, max |d|, g.g
    struct Eval { float f, gtd, gmax, gsum, dmax, gg; };
    float slot_gmax[kSlots] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};           // |g|_inf of the last evaluation into each gradient slot
    RowReq q;
    bool have_comb = false, have_accept = false;
    VecComb comb{};
    float accept_t = 0.0f;
    auto attach = [&]() {
        q.has_comb = have_comb; if (have_comb) q.comb = comb;
        q.accept = have_accept; q.accept_t = accept_t;
        have_comb = false; have_accept = false;
    };
    auto eval = [&](float t, int k, Eval& e) -> bool {
        q.kind = 1; q.t = t; q.k = k;
        attach();
        q.jobs.clear();
        q.jobs.push_back(VecJobN{slot(k), b.dir, 0, 0, (long long)n});
        q.jobs.push_back(VecJobN{slot(k), nullptr, 1, 0, (long long)n});
        q.jobs.push_back(VecJobN{slot(k), nullptr, 2, 0, (long long)n});
        q.jobs.push_back(VecJobN{b.dir, nullptr, 1, 0, (long long)n});
        q.jobs.push_back(VecJobN{slot(k), slot(k), 0, 0, (long long)n});
        if (!B.submit(r, &q)) return false;
        e.f = (float)(B.device_totals ? B.total_h[q.arg0] : total_loss(b.terms_h, b.T));
        e.gtd = (float)B.job_sum(q, 0, nb, false);
        e.gmax = (float)B.job_sum(q, 1, nb, true);
        e.gsum = (float)B.job_sum(q, 2, nb, false);
        e.dmax = (float)B.job_sum(q, 3, nb, true);
        e.gg = (float)B.job_sum(q, 4, nb, false);
        slot_gmax[k] = e.gmax;
        return true;
    };

    int base = 0;
    Eval e0;
    if (!eval(0.0f, base, e0)) return;
    float loss = e0.f, gmax = e0.gmax, gsum = e0.gsum, gg0 = e0.gg;
    res.first_loss = res.loss = loss;
    int evals = 1;
    ok = true;
    if (gmax <= tolerance_grad) { res.n_eval = evals; return; }

    const int NB = 2 * P + 1, IG = 2 * P;
    std::vector<double> G((size_t)NB * NB, 0.0), ro(P, 0.0), al(P, 0.0), delta(NB, 0.0);
    auto Gat = [&](int a, int c) -> double& { return G[(size_t)a * NB + c]; };
    std::vector<int> order;
    int spare = 0;
    double H_diag = 1.0;
    float t = 0.0f, prev_loss = loss, gtd = 0.0f, d_max = 0.0f;
    int prev_base = base;
    int n_iter = 0;
    cubic_fn cubic = rc::Lbfgs<float>::cubic;
    ok = false;
    while (n_iter < max_iter) {
        ++n_iter;
        if (n_iter == 1) {
            comb = VecComb{};
            comb.n_vec = 1; comb.v[0] = slot(base); comb.c[0] = -1.0f;     // d = -g
            gtd = -gg0;                                                     // g.d of d = -g
        } else {
            const int c = spare;
            struct Want { int a, b; };
            std::vector<Want> want;
            std::vector<int> with_c(order);
            with_c.push_back(c);
            for (int j : with_c) { want.push_back({c, j}); want.push_back({c, P + j}); want.push_back({P + c, P + j}); }
            for (int j : order) want.push_back({P + c, j});
            for (int j : with_c) { want.push_back({IG, j}); want.push_back({IG, P + j}); }
            want.push_back({IG, IG});
            auto vec_of = [&](int idx) -> const float* {
                return idx == IG ? slot(base) : (idx >= P ? b.Yv + (size_t)(idx - P) * n : b.Sv + (size_t)idx * n);
            };
            q.kind = 2;
            attach();
            q.g_new = slot(base); q.g_old = slot(prev_base); q.pair_t = t;
            q.yv = b.Yv + (size_t)c * n; q.sv = b.Sv + (size_t)c * n;
            q.jobs.clear();
            for (const Want& w : want) q.jobs.push_back(VecJobN{vec_of(w.a), vec_of(w.b), 0, 0, (long long)n});
            if (!B.submit(r, &q)) return;
            for (size_t i = 0; i < want.size(); ++i) {
                const double v = B.job_sum(q, (int)i, nb, false);
                Gat(want[i].a, want[i].b) = v; Gat(want[i].b, want[i].a) = v;
            }
            {
                const double ys = Gat(c, P + c);
                if ((float)ys > 1e-10f) {
                    H_diag = ys / Gat(P + c, P + c);
                    ro[c] = 1.0 / ys;
                    if ((int)order.size() == M) { spare = order.front(); order.erase(order.begin()); order.push_back(c); }
                    else { order.push_back(c); spare = (int)order.size(); }
                }
            }
            const int m = (int)order.size();
            std::fill(delta.begin(), delta.end(), 0.0);
            delta[IG] = -1.0;
            auto dot_q = [&](int idx) {
                double rr = delta[IG] * Gat(IG, idx);
                for (int j : order) rr += delta[j] * Gat(j, idx) + delta[P + j] * Gat(P + j, idx);
                return rr;
            };
            for (int i = m - 1; i >= 0; --i) { const int j = order[i]; al[j] = dot_q(j) * ro[j]; delta[P + j] -= al[j]; }
            for (double& v : delta) v *= H_diag;
            for (int i = 0; i < m; ++i) { const int j = order[i]; const double be = dot_q(P + j) * ro[j]; delta[j] += al[j] - be; }
            comb = VecComb{};
            comb.n_vec = 0;
            for (int j : order) {
                comb.v[comb.n_vec] = b.Sv + (size_t)j * n; comb.c[comb.n_vec++] = (float)delta[j];
                comb.v[comb.n_vec] = b.Yv + (size_t)j * n; comb.c[comb.n_vec++] = (float)delta[P + j];
            }
            comb.v[comb.n_vec] = slot(base); comb.c[comb.n_vec++] = (float)delta[IG];
            gtd = (float)dot_q(IG);
        }
        have_comb = true;                                                 // formed in front of the line search's first evaluation
        prev_base = base;
        prev_loss = loss;
        t = n_iter == 1 ? std::min(1.0f, 1.0f / gsum) * lr : lr;
        if (gtd > -tolerance_change) break;

        // ---- strong-Wolfe line search (rc_lbfgs.h: strong_wolfe) on gradient slots: the code of minimize_on_device
        struct Pt { float t, f, gtd; int k; };
        const int max_ls = max_eval - evals;
        auto free_slot = [&](std::initializer_list<int> live) {
            for (int k = 0; k < kSlots; ++k) { bool used = false; for (int v : live) used = used || v == k; if (!used) return k; }
            return -1;
        };
        const float c1 = 1e-4f, c2 = 0.9f;
        Pt cur{t, 0, 0, free_slot({base})};
        Eval ev;
        if (!eval(cur.t, cur.k, ev)) return;
        int ls_evals = 1;
        cur.f = ev.f; cur.gtd = ev.gtd; d_max = ev.dmax;
        Pt prev{0, loss, gtd, base};
        Pt br[2] = {prev, prev};
        int n_br = 0, ls_iter = 0;
        bool done = false;
        while (ls_iter < max_ls) {
            if (cur.f > (loss + c1 * cur.t * gtd) || (ls_iter > 1 && cur.f >= prev.f)) { br[0] = prev; br[1] = cur; n_br = 2; break; }
            if (std::fabs(cur.gtd) <= -c2 * gtd) { br[0] = cur; n_br = 1; done = true; break; }
            if (cur.gtd >= 0) { br[0] = prev; br[1] = cur; n_br = 2; break; }
            const float min_step = cur.t + 0.01f * (cur.t - prev.t), max_step = cur.t * 10;
            const float tn = cubic(prev.t, prev.f, prev.gtd, cur.t, cur.f, cur.gtd, true, min_step, max_step);
            prev = cur;
            cur.t = tn;
            cur.k = free_slot({base, prev.k});
            if (!eval(tn, cur.k, ev)) return;
            cur.f = ev.f; cur.gtd = ev.gtd;
            ++ls_evals;
            ++ls_iter;
        }
        if (ls_iter == max_ls) { br[0] = Pt{0, loss, gtd, base}; br[1] = cur; n_br = 2; }
        bool insuf = false;
        int lo = 0, hi = 1;
        if (n_br == 2 && !(br[0].f <= br[1].f)) { lo = 1; hi = 0; }
        while (!done && ls_iter < max_ls) {
            if (std::fabs(br[1].t - br[0].t) * d_max < tolerance_change) break;
            float tn = cubic(br[0].t, br[0].f, br[0].gtd, br[1].t, br[1].f, br[1].gtd, false, 0, 0);
            const float bmax = std::max(br[0].t, br[1].t), bmin = std::min(br[0].t, br[1].t);
            const float eps = 0.1f * (bmax - bmin);
            if (std::min(bmax - tn, tn - bmin) < eps) {
                if (insuf || tn >= bmax || tn <= bmin) {
                    tn = (std::fabs(tn - bmax) < std::fabs(tn - bmin)) ? bmax - eps : bmin + eps;
                    insuf = false;
                } else insuf = true;
            } else insuf = false;
            cur.t = tn;
            cur.k = free_slot({base, br[0].k, br[1].k});
            if (!eval(tn, cur.k, ev)) return;
            cur.f = ev.f; cur.gtd = ev.gtd;
            ++ls_evals;
            ++ls_iter;
            if (cur.f > (loss + c1 * tn * gtd) || cur.f >= br[lo].f) {
                br[hi] = cur;
                if (br[0].f <= br[1].f) { lo = 0; hi = 1; } else { lo = 1; hi = 0; }
            } else {
                if (std::fabs(cur.gtd) <= -c2 * gtd) done = true;
                else if (cur.gtd * (br[hi].t - br[lo].t) >= 0) br[hi] = br[lo];
                br[lo] = cur;
            }
        }
        if (n_br == 1) lo = 0;
        t = br[lo].t;
        loss = br[lo].f;
        const int new_base = br[lo].k;
        have_accept = true; accept_t = t;                                 // x += t d: with the next request (or the final flush)
        // |g|_inf of the accepted point: from the evaluation that produced it (kept per slot)
        base = new_base;
        gmax = base == prev_base ? gmax : slot_gmax[base];
        const bool opt_cond = gmax <= tolerance_grad;
        evals += ls_evals;
        if (n_iter == max_iter) break;
        if (evals >= max_eval) break;
        if (opt_cond) break;
        if (d_max * std::fabs(t) <= tolerance_change) break;
        if (std::fabs(loss - prev_loss) < tolerance_change) break;
    }
    if (have_accept) {                                                    // the last accepted step
        q.kind = 3;
        have_comb = false;
        attach();
        q.jobs.clear();
        if (!B.submit(r, &q)) return;
    }
    res.n_iter = n_iter;
    res.n_eval = evals;
    res.loss = loss;
    ok = true;
}

}  // namespace

void rc_smplify_free(SmplifyState* s) {
    if (!s) return;
    free_work(s);
    if (s->batch_dev) (void)hipFree(s->batch_dev);
    if (s->batch_pin) (void)hipHostFree(s->batch_pin);
    for (float** p : {&s->means, &s->prec, &s->lognll, &s->Kd})
        if (*p) (void)hipFree(*p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

extern "C" {

int rc_smplify_set_prior(rc_ctx* ctx, const float* means, const float* prec, const float* nllw) {
    if (!ctx) return RC_ERR_INVALID;
    if (!means || !prec || !nllw) return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_set_prior: null buffer");
    SmplifyState* s = nullptr;
    if (int rc = state_of(ctx, &s)) return rc;
    float lg[8];
    for (int m = 0; m < 8; ++m) {
        if (!(nllw[m] > 0.0f)) return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_set_prior: nll_weights must be positive");
        lg[m] = logf(nllw[m]);
    }
    SM_TRY(ctx, hipMemcpy(s->means, means, 8 * 69 * sizeof(float), hipMemcpyHostToDevice));
    // rows padded to 72 floats (32-byte aligned for the prior kernel's scalar loads); the second half is P + P^T, the matrix of the
    // gradient of d^T P d (the fp32 sum the gradient kernel used to form per element)
    std::vector<float> pad(2 * kPriorMat, 0.0f);
    for (int m = 0; m < 8; ++m)
        for (int i = 0; i < 69; ++i)
            for (int j = 0; j < 69; ++j) {
                const float a = prec[((size_t)m * 69 + i) * 69 + j], b = prec[((size_t)m * 69 + j) * 69 + i];
                pad[((size_t)m * 69 + i) * kPriorLd + j] = a;
                pad[kPriorMat + ((size_t)m * 69 + i) * kPriorLd + j] = a + b;
            }
    SM_TRY(ctx, hipMemcpy(s->prec, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice));
    SM_TRY(ctx, hipMemcpy(s->lognll, lg, sizeof(lg), hipMemcpyHostToDevice));
    s->have_prior = true;
    return RC_OK;
}

int rc_smplify_set_ref3d(rc_ctx* ctx, const float* ref3d) {
    if (!ctx) return RC_ERR_INVALID;
    SmplifyState* s = nullptr;
    if (int rc = state_of(ctx, &s)) return rc;
    s->ref3d_override = ref3d;
    return RC_OK;
}

int rc_smplify_loss_grad(rc_ctx* ctx, const float* x, const float* kp, const float* ref3d, const float* imu_aa, const float* K,
                         int64_t T, double* loss, float* grad, void* stream) {
    if (!ctx) return RC_ERR_INVALID;
    const BodyConst* body = rc_ctx_body(ctx);
    if (!body) return rc_ctx_fail(ctx, RC_ERR_STATE, "rc_smplify_loss_grad: body not set");
    SmplifyState* s = nullptr;
    if (int rc = state_of(ctx, &s)) return rc;
    if (!s->have_prior) return rc_ctx_fail(ctx, RC_ERR_STATE, "rc_smplify_loss_grad: prior not set (rc_smplify_set_prior)");
    if (T < 0 || T > 0x7fffffff / 99 || !loss) return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_loss_grad: bad arguments");
    if (T == 0) { *loss = 0.0; return RC_OK; }
    if (!x || !kp || !ref3d || !imu_aa || !K || !grad) return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_loss_grad: null buffer");
    if (int rc = reserve(ctx, s, T)) return rc;
    hipStream_t st = (hipStream_t)stream;
    rc_launch_smplify(make_args(s, x, kp, ref3d, imu_aa, K, grad, T, rc_ctx_ign_mask(ctx)), body, st);
    SM_TRY(ctx, hipGetLastError());
    SM_TRY(ctx, hipMemcpyAsync(s->h_terms, s->terms, (size_t)T * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    SM_TRY(ctx, hipStreamSynchronize(st));
    *loss = total_loss(s->h_terms, T);
    return RC_OK;
}

int rc_smplify_run(rc_ctx* ctx, const float* pose, const float* tran, const float* kp, const float* imu_ori, const float* K,
                   int64_t T, float lr, int32_t max_iter, float loss_threshold, float* pose_out, float* tran_out,
                   uint8_t* update, rc_smplify_info* info, void* stream) {
    if (!ctx) return RC_ERR_INVALID;
    const auto t_begin = std::chrono::steady_clock::now();
    const BodyConst* body = rc_ctx_body(ctx);
    if (!body) return rc_ctx_fail(ctx, RC_ERR_STATE, "rc_smplify_run: body not set");
    SmplifyState* s = nullptr;
    if (int rc = state_of(ctx, &s)) return rc;
    if (!s->have_prior) return rc_ctx_fail(ctx, RC_ERR_STATE, "rc_smplify_run: prior not set (rc_smplify_set_prior)");
    if (T <= 0 || T > 0x7fffffff / 99 || max_iter < 1 || !(lr >= 0.0f))
        return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_run: bad arguments");
    if (!pose || !tran || !kp || !imu_ori || !K || !pose_out || !tran_out || !update || !info)
        return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_run: null buffer");
    if (int rc = reserve(ctx, s, T)) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::memset(info, 0, sizeof(*info));
    std::memset(update, 0, (size_t)T);
    s->device_ms = 0.0;

    // pre-check (run.py:24-29): mean residual of the FIRST frame against the threshold
    SM_TRY(ctx, hipMemcpyAsync(s->Kd, K, 9 * sizeof(float), hipMemcpyHostToDevice, st));
    rc_launch_residual(body, pose, tran, kp, s->Kd, 100.0f, rc_ctx_ign_mask(ctx), s->res0, T, st);
    SM_TRY(ctx, hipMemcpyAsync(s->h_res, s->res0, (size_t)T * 33 * sizeof(float), hipMemcpyDeviceToHost, st));
    SM_TRY(ctx, hipStreamSynchronize(st));
    auto frame_mean = [](const float* r) {                  // torch mean(dim=-1) of 33 fp32 values
        float acc = 0.0f;
        for (int v = 0; v < 33; ++v) acc += r[v];
        return acc / 33.0f;
    };
    if (frame_mean(s->h_res) > loss_threshold) {
        if (pose_out != pose) SM_TRY(ctx, hipMemcpyAsync(pose_out, pose, (size_t)T * 216 * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (tran_out != tran) SM_TRY(ctx, hipMemcpyAsync(tran_out, tran, (size_t)T * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
        SM_TRY(ctx, hipStreamSynchronize(st));
        info->status = 0;
        info->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return RC_OK;
    }

    // parameters and constants of the closure (temporal_smplify.py:111-139)
    rc_launch_R2aa(pose, s->x, T * 24, st);                                            // body_pose = axis-angle of the prediction
    SM_TRY(ctx, hipMemcpyAsync(s->x + T * 72, tran, (size_t)T * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    rc_launch_R2aa(imu_ori, s->imu_aa, T * 6, st);
    if (s->ref3d_override) {                                                           // see rc_smplify_set_ref3d
        SM_TRY(ctx, hipMemcpyAsync(s->ref3d, s->ref3d_override, (size_t)T * 99 * sizeof(float), hipMemcpyDeviceToDevice, st));
        s->ref3d_override = nullptr;
    } else {
        rc_launch_body_fk(body, pose, tran, nullptr, s->joint, s->ref3d, T, st);      // preserved 3D landmarks
    }
    SM_TRY(ctx, hipMemcpyAsync(s->h_x, s->x, (size_t)T * 75 * sizeof(float), hipMemcpyDeviceToHost, st));
    SM_TRY(ctx, hipStreamSynchronize(st));

    using L = rc::Lbfgs<float>;
    const size_t n = (size_t)T * 75;
    const char* hv = std::getenv("RC_SMPLIFY_HOST_LBFGS");            // read per call: tests flip it
    const bool host_vectors = hv && std::atoi(hv) != 0;
    if (!host_vectors) {
        // the optimiser's vectors stay on the device (minimize_on_device); s->x is updated in place
        DevResult dr;
        if (int rc = minimize_on_device(ctx, s, body, kp, K, T, lr, max_iter, st, dr)) return rc;
        rc_launch_aa2R(s->x, pose_out, T * 24, st);
        SM_TRY(ctx, hipMemcpyAsync(tran_out, s->x + T * 72, (size_t)T * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
        rc_launch_residual(body, pose_out, tran_out, kp, s->Kd, 100.0f, rc_ctx_ign_mask(ctx), s->res1, T, st);
        SM_TRY(ctx, hipMemcpyAsync(s->h_res + T * 33, s->res1, (size_t)T * 33 * sizeof(float), hipMemcpyDeviceToHost, st));
        SM_TRY(ctx, hipStreamSynchronize(st));
        SM_TRY(ctx, hipGetLastError());
        for (int64_t t = 0; t < T; ++t) update[t] = frame_mean(s->h_res + (T + t) * 33) < frame_mean(s->h_res + t * 33) ? 1 : 0;
        info->status = 1;
        info->n_iter = dr.n_iter;
        info->n_eval = dr.n_eval;
        info->first_loss = dr.first_loss;
        info->final_loss = dr.loss;
        info->device_ms = s->device_ms;
        info->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return RC_OK;
    }
    // RC_SMPLIFY_HOST_LBFGS=1: round 2's formulation (vectors on the host, rc_lbfgs.h run as is) for A/B runs
    L::Vec x(s->h_x, s->h_x + n);
    int hip_rc = RC_OK;
    const SmplifyArgs A = make_args(s, s->x, kp, s->ref3d, s->imu_aa, K, s->grad, T, rc_ctx_ign_mask(ctx));
    L::Objective closure = [&](const L::Vec& xv, L::Vec& g) -> float {
        std::memcpy(s->h_x, xv.data(), n * sizeof(float));
        hipError_t e = hipMemcpyAsync(s->x, s->h_x, n * sizeof(float), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(s->ev0, st);
        rc_launch_smplify(A, body, st);
        if (e == hipSuccess) e = hipEventRecord(s->ev1, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->h_grad, s->grad, n * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->h_terms, s->terms, (size_t)T * 3 * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            if (hip_rc == RC_OK) hip_rc = rc_ctx_fail(ctx, RC_ERR_HIP, (std::string("smplify closure: ") + hipGetErrorString(e)).c_str());
            std::fill(g.begin(), g.end(), 0.0f);            // zero gradient terminates the search
            return 0.0f;
        }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->device_ms += ms;
        std::memcpy(g.data(), s->h_grad, n * sizeof(float));
        return (float)total_loss(s->h_terms, T);
    };
    L::Options opt;
    opt.lr = lr;
    opt.max_iter = max_iter;
    opt.max_eval = max_iter * 5 / 4;
    opt.history_size = lbfgs_history_env();
    const L::Result r = L::minimize(closure, x, opt);
    if (hip_rc != RC_OK) return hip_rc;

    // results (temporal_smplify.py:188-196, run.py:31-34): rotation matrices, new residual, per-frame update mask
    std::memcpy(s->h_x, x.data(), n * sizeof(float));
    SM_TRY(ctx, hipMemcpyAsync(s->x, s->h_x, n * sizeof(float), hipMemcpyHostToDevice, st));
    rc_launch_aa2R(s->x, pose_out, T * 24, st);
    SM_TRY(ctx, hipMemcpyAsync(tran_out, s->x + T * 72, (size_t)T * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    rc_launch_residual(body, pose_out, tran_out, kp, s->Kd, 100.0f, rc_ctx_ign_mask(ctx), s->res1, T, st);
    SM_TRY(ctx, hipMemcpyAsync(s->h_res + T * 33, s->res1, (size_t)T * 33 * sizeof(float), hipMemcpyDeviceToHost, st));
    SM_TRY(ctx, hipStreamSynchronize(st));
    SM_TRY(ctx, hipGetLastError());
    for (int64_t t = 0; t < T; ++t) update[t] = frame_mean(s->h_res + (T + t) * 33) < frame_mean(s->h_res + t * 33) ? 1 : 0;
    info->status = 1;
    info->n_iter = r.n_iter;
    info->n_eval = r.n_eval;
    info->first_loss = r.first_loss;
    info->final_loss = r.loss;
    info->device_ms = s->device_ms;
    info->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return RC_OK;
}

namespace {
struct FiberArg { RowBatch* B; int r; float lr; int max_iter; DevResult* res; char* ok; };
void fiber_main(unsigned lo, unsigned hi) {
    FiberArg* a = reinterpret_cast<FiberArg*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    bool k = false;
    lbfgs_row(*a->B, a->r, a->lr, a->max_iter, *a->res, k);
    *a->ok = k ? 1 : 0;
}
bool smplify_threads_env() {            // RC_SMPLIFY_THREADS=1: round 4's first scheme, a host thread per row (A/B runs)
    static const bool v = [] { const char* e = std::getenv("RC_SMPLIFY_THREADS"); return e && std::atoi(e) != 0; }();
    return v;
}
}  // namespace

int rc_smplify_run_batch(rc_ctx* ctx, int32_t n_rows, const int64_t* T_rows, const float* const* pose, const float* const* tran,
                         const float* const* kp, const float* const* imu_ori, const float* K_host, float lr, int32_t max_iter,
                         float loss_threshold, float* const* pose_out, float* const* tran_out, uint8_t* const* update_host,
                         rc_smplify_info* infos, void* stream) {
    if (!ctx) return RC_ERR_INVALID;
    const auto t_begin = std::chrono::steady_clock::now();
    const BodyConst* body = rc_ctx_body(ctx);
    if (!body) return rc_ctx_fail(ctx, RC_ERR_STATE, "rc_smplify_run_batch: body not set");
    SmplifyState* s0 = nullptr;
    if (int rc = state_of(ctx, &s0)) return rc;
    if (!s0->have_prior) return rc_ctx_fail(ctx, RC_ERR_STATE, "rc_smplify_run_batch: prior not set (rc_smplify_set_prior)");
    if (n_rows < 0 || max_iter < 1 || !(lr >= 0.0f)) return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_run_batch: bad arguments");
    if (n_rows == 0) return RC_OK;
    if (!T_rows || !pose || !tran || !kp || !imu_ori || !K_host || !pose_out || !tran_out || !update_host || !infos)
        return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_run_batch: null buffer");
    for (int r = 0; r < n_rows; ++r)
        if (T_rows[r] <= 0 || T_rows[r] > 0x7fffffff / 99 || !pose[r] || !tran[r] || !kp[r] || !imu_ori[r] || !pose_out[r] || !tran_out[r] || !update_host[r])
            return rc_ctx_fail(ctx, RC_ERR_INVALID, "rc_smplify_run_batch: bad row");
    hipStream_t st = (hipStream_t)stream;
    const int M = std::max(1, std::min(std::min((int)max_iter, RC_LBFGS_MAX_PAIRS), lbfgs_history_env()));
    const int P = M + 1;

    RowBatch B;
    B.ctx = ctx; B.body = body; B.prior = s0; B.st = st; B.ign = rc_ctx_ign_mask(ctx);
    B.row.resize((size_t)n_rows);
    B.pending.assign((size_t)n_rows, nullptr);
    B.max_jobs = 6 * P + 8;
    for (int r = 0; r < n_rows; ++r) {
        RowBuf& b = B.row[r];
        b.T = T_rows[r]; b.n = (size_t)b.T * 75; b.nb = (int)((b.n + 4095) / 4096);
        for (int e = 0; e < 9; ++e) b.K[e] = K_host[9 * r + e];
        b.kp = kp[r];
        B.T_max = std::max(B.T_max, (int)b.T);
        B.n_max = std::max(B.n_max, b.n);
        B.nb_max = std::max(B.nb_max, b.nb);
    }
    // ---- two arenas (device, pinned), carved: two allocations for the whole batch
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t dev_need = 0, pin_need = 0;
    auto dtake = [&](size_t bytes) { const size_t o = dev_need; dev_need += al(bytes); return o; };
    auto ptake = [&](size_t bytes) { const size_t o = pin_need; pin_need += al(bytes); return o; };
    struct Off { size_t x, xt, dir, gs, Sv, Yv, ref, imu, mj, proj, joint, r0, r1, am, pl, pg, fk; };
    std::vector<Off> off((size_t)n_rows);
    for (int r = 0; r < n_rows; ++r) {
        const size_t n = B.row[r].n, T = (size_t)B.row[r].T;
        Off& o = off[r];
        o.x = dtake(n * 4); o.xt = dtake(n * 4); o.dir = dtake(n * 4); o.gs = dtake(kSlots * n * 4);
        o.Sv = dtake((size_t)P * n * 4); o.Yv = dtake((size_t)P * n * 4);
        o.ref = dtake(T * 99 * 4); o.imu = dtake(T * 18 * 4); o.mj = dtake(T * 99 * 4); o.proj = dtake(T * 66 * 4); o.joint = dtake(T * 72 * 4);
        o.am = dtake(T * 4); o.pl = dtake(T * 4); o.pg = dtake(T * 69 * 4); o.fk = dtake(T * 432 * 4);
    }
    // the residuals of all rows in one span each (one read-back per span; the pinned copies mirror the offsets)
    const size_t r0_begin = dev_need;
    for (int r = 0; r < n_rows; ++r) off[r].r0 = dtake((size_t)B.row[r].T * 33 * 4);
    const size_t r_bytes = dev_need - r0_begin, r1_begin = dev_need;
    for (int r = 0; r < n_rows; ++r) off[r].r1 = dtake((size_t)B.row[r].T * 33 * 4);
    const size_t p_r0 = ptake(r_bytes), p_r1 = ptake(r_bytes);
    const size_t nr = (size_t)n_rows;
    const size_t o_io = dtake(nr * sizeof(SmplifyRowIO)), p_io = ptake(nr * sizeof(SmplifyRowIO));
    const size_t o_args = dtake(nr * sizeof(SmplifyArgs)), o_ops = dtake(3 * nr * sizeof(VecOp)), o_comb = dtake(nr * sizeof(VecCombRow));
    const size_t o_jobs = dtake(nr * B.max_jobs * sizeof(VecJobN)), o_part = dtake(nr * B.max_jobs * (size_t)B.nb_max * sizeof(double));
    const size_t o_terms = dtake(nr * 3 * (size_t)B.T_max * 4);
    const size_t o_total = dtake(nr * sizeof(double)), p_total = ptake(nr * sizeof(double));
    const size_t p_args = ptake(nr * sizeof(SmplifyArgs)), p_ops = ptake(3 * nr * sizeof(VecOp)), p_comb = ptake(nr * sizeof(VecCombRow));
    const size_t p_jobs = ptake(nr * B.max_jobs * sizeof(VecJobN)), p_part = ptake(nr * B.max_jobs * (size_t)B.nb_max * sizeof(double));
    const size_t p_terms = ptake(nr * 3 * (size_t)B.T_max * 4);
    if (dev_need > s0->batch_dev_cap) {
        if (s0->batch_dev) (void)hipFree(s0->batch_dev);
        s0->batch_dev = nullptr; s0->batch_dev_cap = 0;
        SM_TRY(ctx, hipMalloc((void**)&s0->batch_dev, dev_need));
        s0->batch_dev_cap = dev_need;
    }
    if (pin_need > s0->batch_pin_cap) {
        if (s0->batch_pin) (void)hipHostFree(s0->batch_pin);
        s0->batch_pin = nullptr; s0->batch_pin_cap = 0;
        SM_TRY(ctx, hipHostMalloc((void**)&s0->batch_pin, pin_need, hipHostMallocDefault));
        s0->batch_pin_cap = pin_need;
    }
    B.dev = s0->batch_dev; B.pin = s0->batch_pin;
    B.dev_bytes = dev_need; B.pin_bytes = pin_need;
    SM_TRY(ctx, hipEventCreate(&B.ev0));
    SM_TRY(ctx, hipEventCreate(&B.ev1));
    B.args_d = (SmplifyArgs*)(B.dev + o_args); B.ops_d = (VecOp*)(B.dev + o_ops); B.comb_d = (VecCombRow*)(B.dev + o_comb);
    B.jobs_d = (VecJobN*)(B.dev + o_jobs); B.part_d = (double*)(B.dev + o_part); B.terms_d = (float*)(B.dev + o_terms);
    B.args_h = (SmplifyArgs*)(B.pin + p_args); B.ops_h = (VecOp*)(B.pin + p_ops); B.comb_h = (VecCombRow*)(B.pin + p_comb);
    B.jobs_h = (VecJobN*)(B.pin + p_jobs); B.part_h = (double*)(B.pin + p_part); B.terms_h = (float*)(B.pin + p_terms);
    B.total_d = (double*)(B.dev + o_total); B.total_h = (double*)(B.pin + p_total);
    if (const char* e = std::getenv("RC_SMPLIFY_DEVICE_TOTALS"); e && *e == '0') B.device_totals = false;
    for (int r = 0; r < n_rows; ++r) {
        RowBuf& b = B.row[r];
        const Off& o = off[r];
        auto f = [&](size_t q) { return (float*)(B.dev + q); };
        b.x = f(o.x); b.xt = f(o.xt); b.dir = f(o.dir); b.gslot = f(o.gs); b.Sv = f(o.Sv); b.Yv = f(o.Yv);
        b.ref3d = f(o.ref); b.imu_aa = f(o.imu); b.mj = f(o.mj); b.proj = f(o.proj); b.joint = f(o.joint); b.res0 = f(o.r0); b.res1 = f(o.r1);
        b.argmin = (int*)(B.dev + o.am); b.prior_ll = f(o.pl); b.prior_g = f(o.pg); b.fk = f(o.fk);
        b.h_res0 = (float*)(B.pin + p_r0 + (o.r0 - r0_begin));
        b.h_res1 = (float*)(B.pin + p_r1 + (o.r1 - r1_begin));
        b.terms = B.terms_d + (size_t)r * 3 * B.T_max;
        b.terms_h = B.terms_h + (size_t)r * 3 * B.T_max;
    }
    // ---- set-up of every row in ONE launch: residual of the initial pose (the pre-check, run.py:24-29, reads its first frame),
    // optimiser parameters, IMU orientations as axis-angle, joints and preserved landmarks (temporal_smplify.py:111-139)
    SmplifyRowIO* io_h = (SmplifyRowIO*)(B.pin + p_io);
    SmplifyRowIO* io_d = (SmplifyRowIO*)(B.dev + o_io);
    for (int r = 0; r < n_rows; ++r) {
        RowBuf& b = B.row[r];
        std::memset(&infos[r], 0, sizeof(infos[r]));
        std::memset(update_host[r], 0, (size_t)b.T);
        SmplifyRowIO& io = io_h[r];
        io.pose = pose[r]; io.tran = tran[r]; io.kp = kp[r]; io.imu_ori = imu_ori[r];
        io.x = b.x; io.imu_aa = b.imu_aa; io.joint = b.joint; io.ref3d = b.ref3d; io.res0 = b.res0; io.res1 = b.res1;
        io.pose_out = pose_out[r]; io.tran_out = tran_out[r];
        for (int e = 0; e < 9; ++e) io.K[e] = b.K[e];
        io.T = (int)b.T; io.live = 0;
    }
    SM_TRY(ctx, hipMemcpyAsync(io_d, io_h, nr * sizeof(SmplifyRowIO), hipMemcpyHostToDevice, st));
    rc_launch_smplify_begin_rows(io_d, n_rows, B.T_max, body, 100.0f, B.ign, st);
    SM_TRY(ctx, hipMemcpyAsync(B.pin + p_r0, B.dev + r0_begin, r_bytes, hipMemcpyDeviceToHost, st));
    SM_TRY(ctx, hipStreamSynchronize(st));
    auto frame_mean = [](const float* v) {
        float acc = 0.0f;
        for (int q = 0; q < 33; ++q) acc += v[q];
        return acc / 33.0f;
    };
    std::vector<int> live;
    for (int r = 0; r < n_rows; ++r) {
        RowBuf& b = B.row[r];
        if (frame_mean(b.h_res0) > loss_threshold) {
            if (pose_out[r] != pose[r]) SM_TRY(ctx, hipMemcpyAsync(pose_out[r], pose[r], (size_t)b.T * 216 * sizeof(float), hipMemcpyDeviceToDevice, st));
            if (tran_out[r] != tran[r]) SM_TRY(ctx, hipMemcpyAsync(tran_out[r], tran[r], (size_t)b.T * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
            infos[r].status = 0;
            continue;
        }
        live.push_back(r);
        io_h[r].live = 1;
    }
    SM_TRY(ctx, hipStreamSynchronize(st));
    SM_TRY(ctx, hipGetLastError());
    const auto t_opt = std::chrono::steady_clock::now();
    // ---- the optimisers: a fiber (or, RC_SMPLIFY_THREADS=1, a host thread) per row, the device work in lock-step rounds
    std::vector<DevResult> res((size_t)n_rows);
    std::vector<char> ok((size_t)n_rows, 0);
    B.n_live = (int)live.size();
    if (live.size() == 1) {
        bool k = false;
        lbfgs_row(B, live[0], lr, max_iter, res[live[0]], k);
        ok[live[0]] = k;
    } else if (!live.empty() && !smplify_threads_env()) {
        B.fibers = true;
        B.fctx.resize((size_t)n_rows);
        // a stack per row with an inaccessible page below it (round-4 advice: in one plain array an overflow ran silently into the
        // neighbouring row's stack; now it faults at the guard page)
        constexpr size_t kStack = 256u << 10;
        const size_t page = (size_t)sysconf(_SC_PAGESIZE), slot = kStack + page;
        struct Stacks {
            void* p = MAP_FAILED; size_t n = 0;
            ~Stacks() { if (p != MAP_FAILED) munmap(p, n); }
        } stacks;
        stacks.n = slot * live.size();
        stacks.p = mmap(nullptr, stacks.n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (stacks.p == MAP_FAILED) return rc_ctx_fail(ctx, RC_ERR_HIP, "smplify batch: cannot map the rows' stacks");
        for (size_t i = 0; i < live.size(); ++i) (void)mprotect((char*)stacks.p + i * slot, page, PROT_NONE);
        std::vector<FiberArg> fa(live.size());
        for (size_t i = 0; i < live.size(); ++i) {
            const int r = live[i];
            fa[i] = FiberArg{&B, r, lr, (int)max_iter, &res[r], &ok[r]};
            getcontext(&B.fctx[r]);
            B.fctx[r].uc_stack.ss_sp = (char*)stacks.p + i * slot + page;
            B.fctx[r].uc_stack.ss_size = kStack;
            B.fctx[r].uc_link = &B.sched;
            const uintptr_t p = (uintptr_t)&fa[i];
            makecontext(&B.fctx[r], (void (*)())fiber_main, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
        }
        std::vector<int> run(live);
        while (!run.empty()) {
            for (int r : run) swapcontext(&B.sched, &B.fctx[r]);          // to the row's next request, or to its end
            run.clear();
            for (int r : live) if (B.pending[r]) run.push_back(r);
            if (!run.empty()) B.run_round();                              // every live row is waiting
        }
    } else if (!live.empty()) {
        std::vector<std::thread> th;
        th.reserve(live.size());
        for (int r : live) th.emplace_back([&, r] { bool k = false; lbfgs_row(B, r, lr, max_iter, res[r], k); ok[r] = k; });
        for (std::thread& t : th) t.join();
    }
    if (B.herr != hipSuccess) return rc_ctx_fail(ctx, RC_ERR_HIP, (std::string("smplify batch: ") + hipGetErrorString(B.herr)).c_str());
    for (int r : live) if (!ok[r]) return rc_ctx_fail(ctx, RC_ERR_HIP, "smplify batch: a row's optimiser did not finish");
    const auto t_res = std::chrono::steady_clock::now();
    // ---- results in ONE launch: rotations, translation, residual after; then the per-frame update mask (run.py:31-34)
    if (!live.empty()) {
        SM_TRY(ctx, hipMemcpyAsync(io_d, io_h, nr * sizeof(SmplifyRowIO), hipMemcpyHostToDevice, st));
        rc_launch_smplify_end_rows(io_d, n_rows, B.T_max, body, 100.0f, B.ign, st);
        SM_TRY(ctx, hipMemcpyAsync(B.pin + p_r1, B.dev + r1_begin, r_bytes, hipMemcpyDeviceToHost, st));
    }
    SM_TRY(ctx, hipStreamSynchronize(st));
    SM_TRY(ctx, hipGetLastError());
    const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (const char* e = std::getenv("RC_SMPLIFY_TRACE"); e && *e == '1') {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[rc_smplify_run_batch] rows %d live %zu: setup %.2f ms, optimisers %.2f ms (%d rounds, %.2f ms inside the executor, closure kernels %.2f ms), results %.2f ms\n",
                     n_rows, live.size(), ms(t_begin, t_opt), ms(t_opt, t_res), B.rounds, B.round_ms, B.device_ms, host_ms - ms(t_begin, t_res));
    }
    for (int r = 0; r < n_rows; ++r) {
        infos[r].host_ms = host_ms;                                       // of the whole batch
        infos[r].device_ms = B.device_ms;
    }
    for (int r : live) {
        RowBuf& b = B.row[r];
        for (int64_t t = 0; t < b.T; ++t) update_host[r][t] = frame_mean(b.h_res1 + t * 33) < frame_mean(b.h_res0 + t * 33) ? 1 : 0;
        infos[r].status = 1;
        infos[r].n_iter = res[r].n_iter;
        infos[r].n_eval = res[r].n_eval;
        infos[r].first_loss = res[r].first_loss;
        infos[r].final_loss = res[r].loss;
        infos[r].reserved = B.rounds;
    }
    return RC_OK;
}

int rc_lbfgs_minimize(rc_objective_fn objective, void* user, int64_t n, double* x, double lr, int32_t max_iter, int32_t max_eval,
                      int32_t history, double tol_grad, double tol_change, int32_t* n_iter, int32_t* n_eval, double* losses,
                      int64_t cap) {
    if (!objective || !x || n <= 0 || max_iter < 1 || max_eval < 1 || history < 1) return RC_ERR_INVALID;
    using L = rc::Lbfgs<double>;
    L::Vec xv(x, x + n);
    L::Objective fn = [&](const L::Vec& p, L::Vec& g) { return objective(user, p.data(), g.data(), n); };
    L::Options o;
    o.lr = lr; o.max_iter = max_iter; o.max_eval = max_eval; o.history_size = history;
    o.tolerance_grad = tol_grad; o.tolerance_change = tol_change;
    const L::Result r = L::minimize(fn, xv, o);
    std::memcpy(x, xv.data(), (size_t)n * sizeof(double));
    if (n_iter) *n_iter = r.n_iter;
    if (n_eval) *n_eval = r.n_eval;
    if (losses) {
        for (int64_t i = 0; i < cap; ++i) losses[i] = i < (int64_t)r.losses.size() ? r.losses[(size_t)i] : 0.0;
    }
    return RC_OK;
}

}  // extern "C"
