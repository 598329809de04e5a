import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_sessionstart(session):
    """No test session runs against a timing-variant build of the library (idee_version() names it; csrc/api.cpp)."""
    from idee_amd import _lib
    if os.path.exists(_lib.LIB_PATH):
        v = _lib.load().idee_version().decode()
        if "TIMING-VARIANT" in v:
            raise pytest.UsageError(f"{_lib.LIB_PATH} is a timing-variant build ({v}): its results are invalid; rebuild without IDEE_EXTRA_FLAGS")


def load_golden(name):
    """-> (arrays dict as torch tensors, state dict under 'sd/', grads under 'grad/')."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    arrs = {k: torch.from_numpy(np.asarray(z[k])) if z[k].dtype.kind in "fiub" else z[k] for k in z.files}
    sd = {k[3:]: v for k, v in arrs.items() if k.startswith("sd/")}
    grads = {k[5:]: v for k, v in arrs.items() if k.startswith("grad/")}
    return arrs, sd, grads


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---------------------------------------------------------------------------------------------
# Kernel variants.  libidee_hip.so looks its IDEE_* kernel-selection options up at EVERY launch (a process-wide table set
# through idee_set_option; the environment only supplies initial values), so the `-m gpu` suite runs the parity tests once
# per variant that ships: every kernel in the library is reached by a driver-run test id
# (e.g. test_swin_block_golden[swin_split-w244_shift]), not by scratch logs.
# ---------------------------------------------------------------------------------------------
KERNEL_VARIANTS = {
    "default": {},                                                   # exact-f32 MFMA everywhere (the library's default arithmetic)
    "swin_split": {"IDEE_SWIN_MATH": "split"},                       # two-piece split GEMMs in the Swin block (opt-in)
    "swin_f22": {"IDEE_SWIN_MATH": "f22"},                           # f16 pieces fwd + bwd, prescaled weights, per-tile gradient scale
    "conv_f22": {"IDEE_CONV_MATH": "f22"},                           # f16 pieces: forward convs, 16->16 data / weight gradients (scaled dy)
    "all_f22": {"IDEE_SWIN_MATH": "f22", "IDEE_CONV_MATH": "f22"},   # bench.py --math f22
    "conv_split": {"IDEE_CONV_MATH": "split"},                       # split-precision convolutions (opt-in)
    "all_split": {"IDEE_SWIN_MATH": "split", "IDEE_CONV_MATH": "split"},
    "fwd22": {"IDEE_SWIN_MATH": "fwd22", "IDEE_CONV_MATH": "fwd22"},   # split kernels in the forward only, exact backward
    "conv16_regs": {"IDEE_CONV16": "regs"},                          # register-staged 16->16 strip kernel
    "wgrad16_regs": {"IDEE_WGRAD16_REGS": "1"},                      # register-staged 3^3 weight gradient
    "conv32_generic": {"IDEE_CONV32": "generic"},                    # 32->32 on the generic kernel
    "dgrad_to16_wide": {"IDEE_DGRAD_TO16": "wide"},                  # (1,3,3) data gradient on conv_wide_kernel
    "wgrad_wide_regs": {"IDEE_WGRAD_WIDE": "regs"},                  # exact wide weight gradient, register-staged (default: LDS-DMA)
    "conv_bf16_direct": {"IDEE_CONV_BF16": "direct"},                # bf16 path: narrow-input convs without LDS staging
    "conv_bf16_lds": {"IDEE_CONV_BF16": "lds"},                      # bf16 path: 32->32 on the LDS kernel, weights from L1
}


def variants(*names):
    """Decorator: run the test once per named kernel variant (the `kernel_variant` fixture sets the switches)."""
    assert all(n in KERNEL_VARIANTS for n in names)
    return pytest.mark.parametrize("kernel_variant", names, indirect=True)


VARIANT_OPTIONS = ("IDEE_SWIN_MATH", "IDEE_CONV_MATH", "IDEE_CONV16", "IDEE_WGRAD16_REGS", "IDEE_CONV32", "IDEE_DGRAD_TO16",
                   "IDEE_CONV_BF16", "IDEE_WGRAD_WIDE")


@pytest.fixture
def set_opt():
    """set_opt(name, value): idee_set_option for the duration of the test (previous values restored at teardown)."""
    from idee_amd import _lib
    saved = {}

    def _set(name, value):
        saved.setdefault(name, _lib.get_option(name))
        _lib.set_option(name, value)
    yield _set
    for k, v in saved.items():
        _lib.set_option(k, v)


@pytest.fixture
def kernel_variant(request, set_opt):
    name = getattr(request, "param", "default")
    for k in VARIANT_OPTIONS:
        set_opt(k, "")
    for k, v in KERNEL_VARIANTS[name].items():
        set_opt(k, v)
    return name


# ---------------------------------------------------------------------------------------------
# ReLU ties.  A ReLU unit whose pre-activation lies within f32 summation-order rounding of zero may
# legitimately get the other mask bit on the GPU, which changes gradients discretely.  Instead of a
# blanket outlier budget the gradient tests LIST such units: the oracle's pre-activations are recorded,
# a test case is accepted only when none of them is a tie (seeds are walked until that holds; the count
# of units inspected is printed), and then every gradient is compared at the strict tolerance.
# ---------------------------------------------------------------------------------------------
TIE_REL = 4e-6      # |a| < TIE_REL * mean|a| of its tensor: ~50x the f32 rounding of a K<=1728 dot product


class relu_record:
    """Context manager: records every tensor the oracle feeds to F.relu (torch.nn.functional.relu)."""

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._orig, self.inputs = F, F.relu, []

        def relu(t, inplace=False):
            self.inputs.append(t.detach())
            return self._orig(t, inplace=False)
        F.relu = relu
        return self

    def __exit__(self, *exc):
        self._F.relu = self._orig
        return False

    def units(self):
        return sum(t.numel() for t in self.inputs)

    def ties(self):
        """Number of recorded pre-activations that are ties."""
        return sum(int((t.abs() < TIE_REL * float(t.abs().mean())).sum()) for t in self.inputs)


def tie_free_case(build, seeds, what=""):
    """build(seed) -> (case, relu_record): first case of `seeds` whose oracle run has no ReLU tie.
    Prints what was inspected, so the (empty) exclusion list is on record in the test log."""
    tried = []
    for seed in seeds:
        case, rec = build(seed)
        n = rec.ties()
        tried.append((seed, n))
        if n == 0:
            print(f"[relu ties] {what}: 0 ties among {rec.units()} pre-activations (seed {seed}; tried {tried})")
            return case
    raise AssertionError(f"{what}: every seed has a ReLU tie {tried}")


# For cases too large to be tie-free by seed (millions of ReLU units): run the oracle backward twice with the derivative
# at the tie units forced to 0 and to 1.  Every gradient element on which the two runs agree is compared strictly; an
# element they disagree on must lie between them (plus the tolerance).  The number of tie units and of affected elements
# is printed: that is the exclusion list.
class tie_relu:
    """Context manager: F.relu with d/dx forced to `mode` (0 or 1) wherever |x| < TIE_REL * mean|x|."""

    def __init__(self, mode):
        self.mode = float(mode)
        self.n_ties = 0
        self.n_units = 0

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._orig = F, F.relu
        outer = self

        class _R(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                tie = x.abs() < TIE_REL * float(x.abs().mean())
                outer.n_ties += int(tie.sum())
                outer.n_units += x.numel()
                ctx.save_for_backward(torch.where(tie, torch.full_like(x, outer.mode), (x > 0).to(x.dtype)))
                return x.clamp_min(0)

            @staticmethod
            def backward(ctx, g):
                return g * ctx.saved_tensors[0]

        F.relu = lambda t, inplace=False: _R.apply(t)
        return self

    def __exit__(self, *exc):
        self._F.relu = self._orig
        return False


def interval_close(name, got, g0, g1, rtol=1e-3, floor=1e-4, exact=None):
    """got within [min(g0,g1) - tol, max(g0,g1) + tol] elementwise, tol = floor * max|g| + rtol * |g|.
    Returns the number of elements on which g0 and g1 differ by more than tol (the tie-affected ones).
    exact: the same gradient from the oracle evaluated in float64, where rounding-level ties do not exist — an element also
    passes within tol of it (the two forced-derivative runs bracket the answer only while tie contributions share a sign; a
    16-element gradient that sums thousands of near-tie units of mixed sign can fall outside both)."""
    got, g0, g1 = (t.detach().float().cpu() for t in (got, g0, g1))
    assert got.shape == g0.shape == g1.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(g0.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite gradient"
    lo, hi = torch.minimum(g0, g1), torch.maximum(g0, g1)
    tol = max(1e-6, floor * float(hi.abs().max())) + rtol * torch.maximum(lo.abs(), hi.abs())
    bad = (got < lo - tol) | (got > hi + tol)
    if exact is not None:
        bad &= (got - exact.detach().float().cpu()).abs() > tol
    if bad.any():
        err = torch.maximum(lo - got, got - hi).clamp_min(0)
        i = int(err.argmax())
        raise AssertionError(f"{name}: {int(bad.sum())}/{got.numel()} outside the tie interval; worst got {float(got.flatten()[i]):.6g} "
                             f"interval [{float(lo.flatten()[i]):.6g}, {float(hi.flatten()[i]):.6g}]")
    return int(((hi - lo) > tol).sum())


def adam_params_close(name, pa, pb, grads_a, grads_b, lr, firm_rel=0.03):
    """Parameters of two runs that differ only by summation-order noise in their gradients (float atomics), after the Adam
    steps whose gradients are grads_a[k] / grads_b[k].  An element is FIRM when at every step its gradient exceeds 100 x the
    largest difference the two runs show at that step: there Adam's update lr * m_hat / (sqrt(v_hat) + eps) moves by ~1 % of
    lr under that noise, and the parameters must agree to firm_rel * lr per step.  On the other elements the SIGN of a step is
    decided by the noise (|step| = lr whatever |g|), so only the trivial bound 2 lr per step applies.  Prints the noise and
    the firm fraction: that is the comparison actually made."""
    steps = len(grads_a)
    firm = torch.ones_like(pa, dtype=torch.bool)
    noises = []
    for ga, gb in zip(grads_a, grads_b):
        noise = float((ga - gb).abs().max())
        noises.append(noise)
        firm &= (ga.abs() > 100.0 * max(noise, 1e-12)) & (gb.abs() > 100.0 * max(noise, 1e-12))
    d = (pa - pb).abs()
    nfirm = int(firm.sum())
    print(f"[{name}] gradient noise per step {['%.2e' % n for n in noises]}; firm elements {nfirm}/{firm.numel()}; "
          f"max |dp| firm {float(d[firm].max()) if nfirm else 0.0:.3e}, other {float(d[~firm].max()) if nfirm < firm.numel() else 0.0:.3e}")
    assert nfirm >= 100, f"{name}: only {nfirm} elements above the noise floor"
    assert float(d[firm].max()) <= firm_rel * lr * steps, f"{name}: firm elements differ by {float(d[firm].max()):.3e}"
    assert float(d.max()) <= 2.0 * lr * steps + 1e-7, f"{name}: an element moved by more than 2 lr per step: {float(d.max()):.3e}"
