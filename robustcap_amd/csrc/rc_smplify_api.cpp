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

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

struct SmplifyState {
    float *means = nullptr, *prec = nullptr, *lognll = nullptr;     // device
    bool have_prior = false;
    int64_t cap = 0;                                                  // frames the buffers below hold
    float *x = nullptr, *grad = nullptr;                              // [cap*75] flat [aa | tran]
    float *ref3d = nullptr, *imu_aa = nullptr, *mj = nullptr, *proj = nullptr, *joint = nullptr;
    float *terms = nullptr;                                           // [3*cap] frame | imu | smooth losses
    int* argmin = nullptr;
    float *res0 = nullptr, *res1 = nullptr, *Kd = nullptr;            // residuals [cap,33] before / after, K on the device
    float *h_x = nullptr, *h_grad = nullptr, *h_terms = nullptr, *h_res = nullptr;   // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double device_ms = 0.0;
    const float* ref3d_override = nullptr;                            // rc_smplify_set_ref3d: caller-owned [T,33,3], next run only
};

namespace {

#define SM_TRY(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return rc_ctx_fail(ctx, RC_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

void free_work(SmplifyState* s) {
    for (float** p : {&s->x, &s->grad, &s->ref3d, &s->imu_aa, &s->mj, &s->proj, &s->joint, &s->terms, &s->res0, &s->res1})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (s->argmin) { (void)hipFree(s->argmin); s->argmin = nullptr; }
    for (float** p : {&s->h_x, &s->h_grad, &s->h_terms, &s->h_res})
        if (*p) { (void)hipHostFree(*p); *p = nullptr; }
    s->cap = 0;
}

int state_of(rc_ctx* ctx, SmplifyState** out) {
    SmplifyState*& s = rc_ctx_smplify(ctx);
    if (!s) {
        s = new SmplifyState();
        SM_TRY(ctx, hipMalloc((void**)&s->means, 8 * 69 * sizeof(float)));
        SM_TRY(ctx, hipMalloc((void**)&s->prec, 8 * 69 * 69 * sizeof(float)));
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
    A.means = s->means; A.prec = s->prec; A.lognll = s->lognll;
    A.mj = s->mj; A.proj = s->proj;
    A.frame_loss = s->terms; A.imu_loss = s->terms + T; A.smooth_loss = s->terms + 2 * T;
    A.argmin = s->argmin;
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

}  // namespace

void rc_smplify_free(SmplifyState* s) {
    if (!s) return;
    free_work(s);
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
    SM_TRY(ctx, hipMemcpy(s->prec, prec, 8 * 69 * 69 * sizeof(float), hipMemcpyHostToDevice));
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
