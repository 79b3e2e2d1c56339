"""Oracle: exact-GP marginal-likelihood fit + predictive mean (CPU, float64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what the reference obtains from GPyTorch at these call sites:
  kernels.py:7-13   ScaleKernel(RBF | Matern nu=.5,1.5,2.5 | Linear) singletons
  kernels.py:15-24  GeneralModel = ConstantMean + that kernel
  gprsr.py:153-174  GaussianLikelihood, initialize(noise=1e-3, mean=mean(y)),
                    RBF only: lengthscale=0.223, outputscale=var(y)
  gprsr.py:176-186  50 x { loss = -mll(model(X), y); backward; Adam(lr=.1).step }
  gprsr.py:188-195  eval(); likelihood(model(X*)).mean
  agpr.py:75-83     same loop, 100 steps, linear kernel, default init
GPyTorch is NOT installed here; its semantics are restated:
  noise = softplus(raw)+1e-4, outputscale/lengthscale/variance = softplus(raw),
  loss = -log N(y | c 1, K + noise I) / n, dense Cholesky (n <= 800),
  float32 parameters promoted into float64 data, torch.optim.Adam defaults.
The linear-kernel branch is pinned by a known-answer test against the reference's
committed checkpoints (tests/test_oracle.py); the stationary kernels, the initial
values and the warm-start chain by output images the reference committed
(tests/test_refpin.py: RBF, Matern-1/2, -3/2, -5/2, bit for bit).
"""
import math

import numpy as np
import torch

RBF, MATERN12, MATERN32, MATERN52, LINEAR = 0, 1, 2, 3, 4
KERNEL_NAMES = {RBF: "RBF", MATERN12: "matern12", MATERN32: "matern32",
                MATERN52: "matern52", LINEAR: "linear"}
NOISE_LOWER = 1e-4  # GaussianLikelihood default GreaterThan(1e-4)


def inv_softplus(x):
    """gpytorch.utils.transforms.inv_softplus: x + log(-expm1(-x))."""
    return x + torch.log(-torch.expm1(-x))


def sq_dist(x1, x2):
    """Direct sum of squared differences, [n1, n2] (float64)."""
    d = x1[:, None, :] - x2[None, :, :]
    return (d * d).sum(-1)


def sq_dist_gpytorch(x1, x2):
    """GPyTorch-style ||a||^2 - 2ab + ||b||^2 after centring on mean(x1),
    clamp_min(0).  Only used to bound the difference to sq_dist()."""
    adj = x1.mean(-2, keepdim=True)
    a, b = x1 - adj, x2 - adj
    an = (a * a).sum(-1, keepdim=True)
    bn = (b * b).sum(-1, keepdim=True)
    res = an - 2.0 * a @ b.T + bn.T
    return res.clamp_min(0)


def base_kernel(x1, x2, ls, kernel_id):
    """Unscaled kernel matrix k(x1, x2) with lengthscale (or, for LINEAR,
    variance) `ls` -- float64 tensors."""
    if kernel_id == LINEAR:
        # LinearKernel: (x1 sqrt(v)) (x2 sqrt(v))^T
        sv = ls.sqrt()
        return (x1 * sv) @ (x2 * sv).T
    d2 = sq_dist(x1, x2)
    if kernel_id == RBF:
        return torch.exp(-0.5 * d2 / (ls * ls))
    # MaternKernel: distance = sqrt(clamp_min(d2, 1e-30)) of the scaled inputs
    r = torch.sqrt(torch.clamp_min(d2 / (ls * ls), 1e-30))
    if kernel_id == MATERN12:
        return torch.exp(-r)
    if kernel_id == MATERN32:
        s = math.sqrt(3.0) * r
        return (s + 1.0) * torch.exp(-s)
    if kernel_id == MATERN52:
        s = math.sqrt(5.0) * r
        return (s + 1.0 + 5.0 / 3.0 * r * r) * torch.exp(-s)
    raise ValueError(kernel_id)


class Hypers:
    """The four raw (unconstrained) float32 scalars Adam sees, in the order of
    model.parameters(): raw_noise, raw_constant, raw_outputscale,
    raw_lengthscale (raw_variance for LINEAR)."""

    def __init__(self, raw_noise=0.0, raw_const=0.0, raw_os=0.0, raw_ls=0.0):
        mk = lambda v: torch.tensor(float(v), dtype=torch.float32, requires_grad=True)
        self.raw_noise, self.raw_const = mk(raw_noise), mk(raw_const)
        self.raw_os, self.raw_ls = mk(raw_os), mk(raw_ls)

    def params(self):
        return [self.raw_noise, self.raw_const, self.raw_os, self.raw_ls]

    def constrained(self):
        sp = torch.nn.functional.softplus
        return (sp(self.raw_noise) + NOISE_LOWER, self.raw_const,
                sp(self.raw_os), sp(self.raw_ls))

    def as_array(self):
        return np.array([p.item() for p in self.params()], dtype=np.float32)


class SharedKernel:
    """Mirror of a kernels.py module-level singleton: raw_outputscale and
    raw_lengthscale live as long as the object and are NOT reset between
    patches unless the caller does it (gprsr.py:167-172 does, RBF only)."""

    def __init__(self, kernel_id):
        self.kernel_id = kernel_id
        self.raw_os = 0.0
        self.raw_ls = 0.0


def neg_mll(h, X, y, kernel_id):
    """-ExactMarginalLogLikelihood(likelihood, model)(model(X), y): per-datum."""
    n = X.shape[0]
    noise, const, os_, ls = h.constrained()
    K = os_.double() * base_kernel(X, X, ls.double(), kernel_id)
    K = K + noise.double() * torch.eye(n, dtype=torch.float64)
    L = torch.linalg.cholesky(K)
    diff = y - const.double()
    alpha = torch.cholesky_solve(diff[:, None], L)[:, 0]
    inv_quad = diff @ alpha
    logdet = 2.0 * torch.log(torch.diagonal(L)).sum()
    return 0.5 * (inv_quad + logdet + n * math.log(2.0 * math.pi)) / n


def init_hypers(y, kernel_id, shared=None):
    """gprsr.py:162-174: noise=1e-3, constant=mean(y.float()); RBF:
    lengthscale=0.223, outputscale=var(y.float()); other kernels keep what the
    shared singleton holds (raw 0 in a fresh process)."""
    yf = y.float()
    raw_noise = inv_softplus(torch.tensor(0.001) - torch.tensor(NOISE_LOWER, dtype=torch.float32))
    raw_const = yf.mean()
    if kernel_id == RBF:
        raw_ls = inv_softplus(torch.tensor(0.223))
        raw_os = inv_softplus(yf.var())
    else:
        raw_os = torch.tensor(shared.raw_os if shared is not None else 0.0)
        raw_ls = torch.tensor(shared.raw_ls if shared is not None else 0.0)
    return Hypers(raw_noise.item(), raw_const.item(), raw_os.item(), raw_ls.item())


def adam_fit(h, X, y, kernel_id, iters=50, lr=0.1, trace=None):
    """gprsr.py:178-186 / agpr.py:75-83."""
    opt = torch.optim.Adam(h.params(), lr=lr)
    for it in range(iters):
        opt.zero_grad()
        loss = neg_mll(h, X, y, kernel_id)
        loss.backward()
        if trace is not None:
            trace.append((loss.item(), h.as_array().copy(),
                          np.array([p.grad.item() for p in h.params()], dtype=np.float32)))
        opt.step()
    return h


@torch.no_grad()
def predictive_mean(h, X, y, Xs, kernel_id, chunk=4096):
    """eval-mode mean: c + os k(X*, X) (K + noise I)^-1 (y - c)  (gprsr.py:195)."""
    n = X.shape[0]
    noise, const, os_, ls = h.constrained()
    K = os_.double() * base_kernel(X, X, ls.double(), kernel_id)
    K = K + noise.double() * torch.eye(n, dtype=torch.float64)
    L = torch.linalg.cholesky(K)
    alpha = torch.cholesky_solve((y - const.double())[:, None], L)[:, 0]
    out = torch.empty(Xs.shape[0], dtype=torch.float64)
    for s in range(0, Xs.shape[0], chunk):
        Ks = os_.double() * base_kernel(Xs[s:s + chunk], X, ls.double(), kernel_id)
        out[s:s + chunk] = Ks @ alpha + const.double()
    return out, alpha


def fit_predict_patch(X, y, Xs, kernel_id, shared=None, iters=50, lr=0.1):
    """One pass of the body of the reference's patch loop (gprsr.py:146-197).
    `shared` (SharedKernel) reproduces the warm-start chain of the kernel
    singletons; shared=None gives the independent (fresh-process) start."""
    X = torch.as_tensor(X, dtype=torch.float64)
    y = torch.as_tensor(y, dtype=torch.float64)
    Xs = torch.as_tensor(Xs, dtype=torch.float64)
    h = init_hypers(y, kernel_id, shared)
    adam_fit(h, X, y, kernel_id, iters, lr)
    if shared is not None:
        shared.raw_os = h.raw_os.item()
        shared.raw_ls = h.raw_ls.item()
    mean, _ = predictive_mean(h, X, y, Xs, kernel_id)
    return mean.numpy(), h.as_array()


# ----------------------------------------------------------------------------
# Closed-form gradient (what the HIP path evaluates); checked against autograd
# in tests/test_oracle_gp.py.
# ----------------------------------------------------------------------------
def kernel_and_dl(d2, ls, kernel_id):
    """numpy: base kernel k and dk/d(lengthscale) from squared distances."""
    if kernel_id == RBF:
        k = np.exp(-0.5 * d2 / (ls * ls))
        return k, k * d2 / ls ** 3
    r = np.sqrt(np.maximum(d2 / (ls * ls), 1e-30))
    if kernel_id == MATERN12:
        k = np.exp(-r)
        return k, k * r / ls
    if kernel_id == MATERN32:
        s = math.sqrt(3.0) * r
        e = np.exp(-s)
        return (1.0 + s) * e, 3.0 * r * r * e / ls
    if kernel_id == MATERN52:
        s = math.sqrt(5.0) * r
        e = np.exp(-s)
        return (1.0 + s + 5.0 / 3.0 * r * r) * e, (5.0 / 3.0) * r * r * (1.0 + s) * e / ls
    raise ValueError(kernel_id)


def neg_mll_and_grad_closed_form(raw, X, y, kernel_id):
    """raw = float32[4] (noise, const, os, ls).  Returns loss and d loss/d raw
    via  d/dtheta = (1/2n) sum_ij (Kinv_ij - a_i a_j) dK_ij,  d/dc = -sum(a)/n."""
    raw = np.asarray(raw, dtype=np.float32)
    sp = lambda v: np.float32(np.log1p(np.exp(-abs(np.float32(v)))) + max(np.float32(v), np.float32(0)))
    sig = lambda v: np.float32(1.0 / (1.0 + np.exp(-np.float32(v))))
    noise = np.float64(np.float32(sp(raw[0]) + np.float32(NOISE_LOWER)))
    const = np.float64(raw[1])
    os_ = np.float64(sp(raw[2]))
    ls = np.float64(sp(raw[3]))
    n = X.shape[0]
    if kernel_id == LINEAR:
        G = X @ X.T
        kb, dl = ls * G, G
    else:
        d = X[:, None, :] - X[None, :, :]
        kb, dl = kernel_and_dl((d * d).sum(-1), ls, kernel_id)
    K = os_ * kb + noise * np.eye(n)
    L = np.linalg.cholesky(K)
    Kinv = np.linalg.inv(K)
    r = y - const
    a = Kinv @ r
    loss = 0.5 * (r @ a + 2 * np.log(np.diag(L)).sum() + n * math.log(2 * math.pi)) / n
    W = Kinv - np.outer(a, a)
    g_noise = 0.5 * np.trace(W) / n
    g_const = -a.sum() / n
    g_os = 0.5 * (W * kb).sum() / n
    g_ls = 0.5 * (W * (os_ * dl)).sum() / n
    grad = np.array([np.float32(g_noise) * sig(raw[0]), np.float32(g_const),
                     np.float32(g_os) * sig(raw[2]), np.float32(g_ls) * sig(raw[3])], dtype=np.float32)
    return loss, grad
