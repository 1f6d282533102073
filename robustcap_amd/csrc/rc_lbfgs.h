// L-BFGS with a strong-Wolfe line search, host side.
//
// The reference optimises with torch.optim.LBFGS(max_iter=20, lr=step_size, line_search_fn='strong_wolfe')
// (net/smplify/temporal_smplify.py:141-147); torch is a third-party dependency (torch/optim/lbfgs.py, 2.x), so the
// algorithm is restated here from its published behaviour: minFunc-style two-loop recursion with a single scaling
// H_diag = y.s / y.y, curvature pairs kept only when y.s > 1e-10, first trial step min(1, 1/|g|_1) * lr and lr
// afterwards, bracketing + zoom line search with safeguarded cubic interpolation (Wolfe c1 = 1e-4, c2 = 0.9), and the
// termination tests in torch's order (max_iter, max_eval, |g|_inf, |t d|_inf, |f - f_prev|).
// `Real` is the arithmetic type of every scalar of the search (float for the smplify path, whose parameters and
// losses are fp32 in the reference; double for rc_lbfgs_minimize, which tests/ pin against torch in float64).
// Reductions accumulate in double and round once.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

namespace rc {

template <class Real>
struct Lbfgs {
    using Vec = std::vector<Real>;
    using Objective = std::function<Real(const Vec& x, Vec& grad)>;   // loss at x; fills grad (same length)

    struct Options {
        Real lr = 1;
        int max_iter = 20;
        int max_eval = 25;                // torch: max_iter * 5 // 4
        Real tolerance_grad = Real(1e-7);
        Real tolerance_change = Real(1e-9);
        int history_size = 100;
    };
    struct Result {
        int n_iter = 0, n_eval = 0;
        Real first_loss = 0, loss = 0;
        std::vector<Real> losses;          // every objective value in evaluation order
    };

    static Real dot(const Vec& a, const Vec& b) {
        double s = 0.0;
        for (size_t i = 0; i < a.size(); ++i) s += (double)a[i] * (double)b[i];
        return (Real)s;
    }
    static Real abs_max(const Vec& a) {
        Real m = 0;
        for (Real v : a) m = std::max(m, (Real)std::fabs(v));
        return m;
    }
    static Real abs_sum(const Vec& a) {
        double s = 0.0;
        for (Real v : a) s += std::fabs((double)v);
        return (Real)s;
    }

    // minimiser of the cubic through (x1, f1, g1), (x2, f2, g2), clamped to the bounds (bisection if it has none)
    static Real cubic(Real x1, Real f1, Real g1, Real x2, Real f2, Real g2, bool bounded, Real lo, Real hi) {
        if (!bounded) { lo = std::min(x1, x2); hi = std::max(x1, x2); }
        const Real d1 = g1 + g2 - 3 * (f1 - f2) / (x1 - x2);
        const Real d2sq = d1 * d1 - g1 * g2;
        if (d2sq >= 0) {
            const Real d2 = std::sqrt(d2sq);
            const Real pos = (x1 <= x2) ? x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2))
                                        : x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2));
            return std::min(std::max(pos, lo), hi);     // NaN position falls through max/min like Python's
        }
        return (lo + hi) / 2;
    }

    struct Point { Real t, f, gtd; Vec g; };

    // returns the accepted (t, f, g) and the number of objective evaluations
    static int strong_wolfe(const Objective& fn, const Vec& x, Real& t, const Vec& d, Real& f, Vec& g, Real gtd,
                            Real tolerance_change, int max_ls, std::vector<Real>& log) {
        const Real c1 = Real(1e-4), c2 = Real(0.9);
        const Real d_norm = abs_max(d);
        Vec xt(x.size());
        auto eval = [&](Real step, Vec& grad) {
            for (size_t i = 0; i < x.size(); ++i) xt[i] = x[i] + step * d[i];
            const Real v = fn(xt, grad);
            log.push_back(v);
            return v;
        };
        Point cur{t, 0, 0, Vec(x.size())};
        cur.f = eval(t, cur.g);
        int evals = 1;
        cur.gtd = dot(cur.g, d);

        Point prev{0, f, gtd, g};
        Point br[2];
        int n_br = 0;
        bool done = false;
        int ls_iter = 0;
        while (ls_iter < max_ls) {
            if (cur.f > (f + c1 * cur.t * gtd) || (ls_iter > 1 && cur.f >= prev.f)) { br[0] = prev; br[1] = cur; n_br = 2; break; }
            if (std::fabs(cur.gtd) <= -c2 * gtd) { br[0] = cur; n_br = 1; done = true; break; }
            if (cur.gtd >= 0) { br[0] = prev; br[1] = cur; n_br = 2; break; }
            const Real min_step = cur.t + Real(0.01) * (cur.t - prev.t), max_step = cur.t * 10;
            const Real tn = cubic(prev.t, prev.f, prev.gtd, cur.t, cur.f, cur.gtd, true, min_step, max_step);
            prev = cur;
            cur.t = tn;
            cur.f = eval(tn, cur.g);
            ++evals;
            cur.gtd = dot(cur.g, d);
            ++ls_iter;
        }
        if (ls_iter == max_ls) { br[0] = Point{0, f, gtd, g}; br[1] = cur; n_br = 2; }

        bool insuf = false;
        int lo = 0, hi = 1;
        if (n_br == 2 && !(br[0].f <= br[1].f)) { lo = 1; hi = 0; }
        while (!done && ls_iter < max_ls) {
            if (std::fabs(br[1].t - br[0].t) * d_norm < tolerance_change) break;
            Real tn = cubic(br[0].t, br[0].f, br[0].gtd, br[1].t, br[1].f, br[1].gtd, false, 0, 0);
            const Real bmax = std::max(br[0].t, br[1].t), bmin = std::min(br[0].t, br[1].t);
            const Real eps = Real(0.1) * (bmax - bmin);
            if (std::min(bmax - tn, tn - bmin) < eps) {
                if (insuf || tn >= bmax || tn <= bmin) {
                    tn = (std::fabs(tn - bmax) < std::fabs(tn - bmin)) ? bmax - eps : bmin + eps;
                    insuf = false;
                } else insuf = true;
            } else insuf = false;

            cur.t = tn;
            cur.f = eval(tn, cur.g);
            ++evals;
            cur.gtd = dot(cur.g, d);
            ++ls_iter;

            if (cur.f > (f + c1 * tn * gtd) || cur.f >= br[lo].f) {
                br[hi] = cur;
                if (br[0].f <= br[1].f) { lo = 0; hi = 1; } else { lo = 1; hi = 0; }
            } else {
                if (std::fabs(cur.gtd) <= -c2 * gtd) done = true;
                else if (cur.gtd * (br[hi].t - br[lo].t) >= 0) br[hi] = br[lo];
                br[lo] = cur;
            }
        }
        t = br[lo].t;
        f = br[lo].f;
        g = br[lo].g;
        return evals;
    }

    // one optimizer.step(closure) of a fresh torch.optim.LBFGS; x is updated in place
    static Result minimize(const Objective& fn, Vec& x, const Options& o) {
        Result res;
        const size_t n = x.size();
        Vec g(n), d, prev_g;
        Real loss = fn(x, g);
        res.losses.push_back(loss);
        res.first_loss = res.loss = loss;
        int evals = 1;
        if (abs_max(g) <= o.tolerance_grad) { res.n_eval = evals; return res; }

        std::vector<Vec> old_dirs, old_stps;
        std::vector<Real> ro, al((size_t)std::max(o.history_size, 1));
        Real H_diag = 1, t = 0, prev_loss = loss;
        int n_iter = 0;
        while (n_iter < o.max_iter) {
            ++n_iter;
            if (n_iter == 1) {
                d.resize(n);
                for (size_t i = 0; i < n; ++i) d[i] = -g[i];
            } else {
                Vec y(n), s(n);
                for (size_t i = 0; i < n; ++i) { y[i] = g[i] - prev_g[i]; s[i] = d[i] * t; }
                const Real ys = dot(y, s);
                if (ys > Real(1e-10)) {
                    if ((int)old_dirs.size() == o.history_size) {
                        old_dirs.erase(old_dirs.begin());
                        old_stps.erase(old_stps.begin());
                        ro.erase(ro.begin());
                    }
                    H_diag = ys / dot(y, y);
                    old_dirs.push_back(std::move(y));
                    old_stps.push_back(std::move(s));
                    ro.push_back(Real(1) / ys);
                }
                const int m = (int)old_dirs.size();
                Vec q(n);
                for (size_t i = 0; i < n; ++i) q[i] = -g[i];
                for (int i = m - 1; i >= 0; --i) {
                    al[i] = dot(old_stps[i], q) * ro[i];
                    for (size_t k = 0; k < n; ++k) q[k] -= al[i] * old_dirs[i][k];
                }
                for (size_t k = 0; k < n; ++k) q[k] *= H_diag;
                for (int i = 0; i < m; ++i) {
                    const Real be = dot(old_dirs[i], q) * ro[i];
                    const Real c = al[i] - be;
                    for (size_t k = 0; k < n; ++k) q[k] += c * old_stps[i][k];
                }
                d = std::move(q);
            }
            prev_g = g;
            prev_loss = loss;
            t = (n_iter == 1) ? std::min(Real(1), Real(1) / abs_sum(g)) * o.lr : o.lr;
            const Real gtd = dot(g, d);
            if (gtd > -o.tolerance_change) break;

            const int ls = strong_wolfe(fn, x, t, d, loss, g, gtd, o.tolerance_change, o.max_eval - evals, res.losses);
            for (size_t i = 0; i < n; ++i) x[i] += t * d[i];
            const bool opt_cond = abs_max(g) <= o.tolerance_grad;
            evals += ls;

            if (n_iter == o.max_iter) break;
            if (evals >= o.max_eval) break;
            if (opt_cond) break;
            Real step_max = 0;
            for (size_t i = 0; i < n; ++i) step_max = std::max(step_max, (Real)std::fabs(d[i] * t));
            if (step_max <= o.tolerance_change) break;
            if (std::fabs(loss - prev_loss) < o.tolerance_change) break;
        }
        res.n_iter = n_iter;
        res.n_eval = evals;
        res.loss = loss;
        return res;
    }
};

}  // namespace rc
