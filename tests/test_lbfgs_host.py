"""The optimiser of the smplify path (csrc/rc_lbfgs.h) against torch.optim.LBFGS -- the implementation the reference
itself calls (net/smplify/temporal_smplify.py:141-147) -- in float64, evaluation by evaluation. Host code only."""
import ctypes as C

import numpy as np
import pytest
import torch

from robustcap_amd import _lib


def _torch_run(fn, x0, lr, max_iter, hist):
    x = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.LBFGS([x], lr=lr, max_iter=max_iter, history_size=hist, line_search_fn="strong_wolfe")
    losses = []

    def closure():
        opt.zero_grad()
        loss = fn(x)
        loss.backward()
        losses.append(float(loss.detach()))
        return loss
    opt.step(closure)
    return x.detach().numpy(), losses, opt.state[x]["n_iter"]


def _ours(fn, x0, lr, max_iter, hist):
    lib = _lib.load()

    def objective(user, xp, gp, n):
        xv = torch.tensor(np.ctypeslib.as_array(xp, (n,)).copy(), dtype=torch.float64, requires_grad=True)
        loss = fn(xv)
        loss.backward()
        np.ctypeslib.as_array(gp, (n,))[:] = xv.grad.numpy()
        return float(loss.detach())
    cb = _lib.OBJECTIVE_FN(objective)
    x = np.array(x0, dtype=np.float64)
    n_iter, n_eval, losses = C.c_int32(), C.c_int32(), np.zeros(256)
    rc = lib.rc_lbfgs_minimize(cb, None, len(x), x.ctypes.data_as(C.POINTER(C.c_double)), lr, max_iter, max_iter * 5 // 4, hist,
                               1e-7, 1e-9, C.byref(n_iter), C.byref(n_eval), losses.ctypes.data_as(C.POINTER(C.c_double)), 256)
    assert rc == 0
    return x, losses[:n_eval.value], n_iter.value


def rosenbrock(x):
    return (100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2).sum()


def kinked(x):          # non-smooth (|.|) + exponential terms, like the smoothness and angle-prior terms of the fitting loss
    return torch.exp(0.3 * x).sum() + 0.1 * (x ** 4).sum() + torch.abs(x - 0.3).sum() + torch.sin(3 * x).sum()


CASES = [("rosen_lr1", rosenbrock, np.linspace(-1.2, 1.0, 10), 1.0, 20, 100),
         ("rosen_lr1e-3", rosenbrock, np.linspace(-1.2, 1.0, 10), 1e-3, 20, 100),        # the reference's lr
         ("rosen_short_history", rosenbrock, np.linspace(-1.2, 1.0, 10), 1.0, 60, 4),    # history eviction
         ("kinked", kinked, np.linspace(-2, 2, 37), 0.5, 40, 100),
         ("kinked_lr1e-3", kinked, np.linspace(-2, 2, 37), 1e-3, 20, 100),
         ("at_optimum", rosenbrock, np.ones(6), 1.0, 20, 100)]                           # |g|_inf <= tolerance_grad: one evaluation


@pytest.mark.parametrize("name,fn,x0,lr,max_iter,hist", CASES, ids=[c[0] for c in CASES])
def test_matches_torch_lbfgs(name, fn, x0, lr, max_iter, hist):
    xt, lt, it = _torch_run(fn, x0, lr, max_iter, hist)
    xo, lo, io = _ours(fn, x0, lr, max_iter, hist)
    assert len(lo) == len(lt) and io == it, (len(lo), len(lt), io, it)               # same evaluation / iteration counts
    assert np.allclose(lo, lt, rtol=1e-6, atol=1e-12)                                     # same trial points
    assert np.allclose(xo, xt, rtol=1e-6, atol=1e-8)


def test_rejects_bad_arguments():
    lib = _lib.load()
    x = np.zeros(3)
    cb = _lib.OBJECTIVE_FN(lambda u, xp, gp, n: 0.0)
    assert lib.rc_lbfgs_minimize(cb, None, 0, x.ctypes.data_as(C.POINTER(C.c_double)), 1.0, 20, 25, 100, 1e-7, 1e-9, None, None, None, 0) != 0
    assert lib.rc_lbfgs_minimize(cb, None, 3, x.ctypes.data_as(C.POINTER(C.c_double)), 1.0, 0, 25, 100, 1e-7, 1e-9, None, None, None, 0) != 0
