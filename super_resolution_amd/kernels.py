"""Kernel registry: drop-in for the reference's kernels.py without GPyTorch.

The reference's option values are kernel *objects* (module-level singletons,
kernels.py:7-13) handed to GPRSR(..., used_kernel=...) and to get_kernel_name
(kernels.py:26-47).  Here they are small descriptor objects carrying
  * the kernel family the HIP path selects (kernel_id of include/gpsr.h), and
  * the raw (pre-softplus) outputscale / lengthscale, which -- exactly as in the
    reference -- persist on the singleton between patches, phases and images;
    that is what the "chain" warm-start mode reads and writes.
Periodic / spectral-mixture / additive / product kernels exist by name (so
get_kernel_name keeps its tags) but are outside the accelerated path.
"""
import numpy as np

from . import ops


class _Base:
    kernel_id = None
    #: raw lengthscale (raw variance for LinearKernel); GPyTorch initialises raws to 0
    raw_parameter = 0.0


class RBFKernel(_Base):
    kernel_id = ops.RBF


class MaternKernel(_Base):
    def __init__(self, nu=2.5):
        if nu not in (0.5, 1.5, 2.5):
            raise ValueError("nu must be 0.5, 1.5 or 2.5")
        self.nu = nu
        self.kernel_id = {0.5: ops.MATERN12, 1.5: ops.MATERN32, 2.5: ops.MATERN52}[nu]


class LinearKernel(_Base):
    kernel_id = ops.LINEAR


class PeriodicKernel(_Base):
    pass


class SpectralMixtureKernel(_Base):
    def __init__(self, num_mixtures=4, ard_num_dims=8):
        self.num_mixtures, self.ard_num_dims = num_mixtures, ard_num_dims


class AdditiveKernel(_Base):
    def __init__(self, *kernels):
        self.kernels = list(kernels)


class ProductKernel(_Base):
    def __init__(self, *kernels):
        self.kernels = list(kernels)


class ScaleKernel:
    def __init__(self, base_kernel):
        self.base_kernel = base_kernel
        self.raw_outputscale = 0.0

    @property
    def kernel_id(self):
        return self.base_kernel.kernel_id

    def reset(self):
        """Back to the fresh-process state (all raws 0)."""
        self.raw_outputscale = 0.0
        self.base_kernel.raw_parameter = 0.0


rbf_kernel = ScaleKernel(RBFKernel())
matern52_kernel = ScaleKernel(MaternKernel(nu=2.5))
matern32_kernel = ScaleKernel(MaternKernel(nu=1.5))
exponential_kernel = ScaleKernel(MaternKernel(nu=0.5))
periodic_kernel = ScaleKernel(PeriodicKernel())
spectral_mixture_kernel = SpectralMixtureKernel(num_mixtures=4, ard_num_dims=8)
linear_kernel = ScaleKernel(LinearKernel())


def get_kernel_name(kernel):
    """File-name tag of a kernel object (reference kernels.py:26-47)."""
    if isinstance(kernel, ScaleKernel):
        return get_kernel_name(kernel.base_kernel)
    if isinstance(kernel, RBFKernel):
        return "RBF"
    if isinstance(kernel, MaternKernel):
        return "matern" + {2.5: "52", 1.5: "32", 0.5: "12"}.get(kernel.nu, str(kernel.nu))
    if isinstance(kernel, PeriodicKernel):
        return "periodic"
    if isinstance(kernel, SpectralMixtureKernel):
        return "spectral_mixture"
    if isinstance(kernel, LinearKernel):
        return "linear"
    if isinstance(kernel, AdditiveKernel):
        return "add(" + ",".join(get_kernel_name(k) for k in kernel.kernels) + ")"
    if isinstance(kernel, ProductKernel):
        return "prod(" + ",".join(get_kernel_name(k) for k in kernel.kernels) + ")"
    raise Exception(f"Unknown kernel type: {kernel}")


def kernel_id_of(kernel):
    kid = getattr(kernel, "kernel_id", None)
    if kid is None:
        raise NotImplementedError(
            f"kernel '{get_kernel_name(kernel)}' is outside the accelerated path "
            "(RBF, Matern 1/2, 3/2, 5/2 and linear are supported)")
    return kid


class GeneralModel:
    """ConstantMean + the given kernel on (train_x, train_y): the role of the
    reference's ExactGP subclass (kernels.py:15-24), as one GP on the GPU.

    model = GeneralModel(kernel, train_x, train_y); model.initialize(noise=1e-3, ...)
    model.fit(iters, lr); mean = model.predict(test_x)
    """

    def __init__(self, covar_module, train_x, train_y, likelihood=None):
        import torch
        self.covar_module = covar_module
        self.kernel_id = kernel_id_of(covar_module)
        dev = torch.device("cuda", torch.cuda.current_device())   # one process per GPU: the rank's own device
        self.train_x = torch.as_tensor(np.asarray(train_x, dtype=np.float64)).to(dev).contiguous()
        self.train_y = torch.as_tensor(np.asarray(train_y, dtype=np.float64)).to(dev).contiguous()
        self.raw = torch.tensor([[0.0, 0.0, covar_module.raw_outputscale, covar_module.base_kernel.raw_parameter]],
                                dtype=torch.float32, device=dev)
        self.alpha = None

    @staticmethod
    def _inv_softplus(v):
        v = np.float32(v)
        return float(v + np.log(-np.expm1(-v)))

    def initialize(self, noise=None, constant=None, outputscale=None, lengthscale=None, variance=None):
        r = self.raw.cpu().numpy()
        if noise is not None:
            r[0, 0] = self._inv_softplus(np.float32(noise) - np.float32(1e-4))
        if constant is not None:
            r[0, 1] = constant
        if outputscale is not None:
            r[0, 2] = self._inv_softplus(outputscale)
        if lengthscale is not None or variance is not None:
            r[0, 3] = self._inv_softplus(lengthscale if lengthscale is not None else variance)
        self.raw = __import__("torch").as_tensor(r).to(self.raw.device)
        return self

    def fit(self, iters=50, lr=0.1):
        X, y = self.train_x[None], self.train_y[None]
        self.alpha, self.flags = ops.fit(X, y, self.kernel_id, self.raw, iters, lr)
        r = self.raw[0].cpu()
        self.covar_module.raw_outputscale = float(r[2])
        self.covar_module.base_kernel.raw_parameter = float(r[3])
        return self

    def state_dict(self):
        """Keys and shapes of the reference's checkpoints."""
        import torch
        r = self.raw[0].cpu()
        pname = "raw_variance" if self.kernel_id == ops.LINEAR else "raw_lengthscale"
        return {
            "likelihood.noise_covar.raw_noise": r[0].reshape(1).clone(),
            "likelihood.noise_covar.raw_noise_constraint.lower_bound": torch.tensor(1e-4),
            "likelihood.noise_covar.raw_noise_constraint.upper_bound": torch.tensor(float("inf")),
            "mean_module.raw_constant": r[1].reshape(()).clone(),
            "covar_module.raw_outputscale": r[2].reshape(()).clone(),
            f"covar_module.base_kernel.{pname}": r[3].reshape(1, 1).clone(),
            f"covar_module.base_kernel.{pname}_constraint.lower_bound": torch.tensor(0.0),
            f"covar_module.base_kernel.{pname}_constraint.upper_bound": torch.tensor(float("inf")),
            "covar_module.raw_outputscale_constraint.lower_bound": torch.tensor(0.0),
            "covar_module.raw_outputscale_constraint.upper_bound": torch.tensor(float("inf")),
        }

    def load_state_dict(self, sd):
        import torch
        pname = "raw_variance" if self.kernel_id == ops.LINEAR else "raw_lengthscale"
        vals = [sd["likelihood.noise_covar.raw_noise"], sd["mean_module.raw_constant"],
                sd["covar_module.raw_outputscale"], sd[f"covar_module.base_kernel.{pname}"]]
        r = torch.tensor([[float(torch.as_tensor(v).reshape(-1)[0]) for v in vals]], dtype=torch.float32)
        self.raw = r.to(self.raw.device)
        self.alpha = None
        return self

    def ensure_alpha(self):
        if self.alpha is None:  # factorise at the current hyper-parameters, no Adam step
            X, y = self.train_x[None], self.train_y[None]
            self.alpha, self.flags = ops.fit(X, y, self.kernel_id, self.raw, 0, 0.1)
        return self.alpha

    def predict(self, test_x):
        import torch
        self.ensure_alpha()
        Xs = torch.as_tensor(np.asarray(test_x, dtype=np.float64)).to(self.raw.device).contiguous()
        return ops.predict(self.train_x[None], Xs[None], self.alpha, self.raw, self.kernel_id)[0].cpu().numpy()
