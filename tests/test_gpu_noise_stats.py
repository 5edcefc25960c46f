"""Statistics of the sampler's in-kernel noise on >= 1e8 samples of the STEP field itself (csrc/common.h: Philox4x32-7, Box-Muller,
16-bit radius grid refined by 32 more bits below u = 2^-12, 16-bit angle; nu_t = fp16(sigma_t z)), written out by
gencomm_step_noise_fwd with the device functions the sampler kernels call. The reference draws torch.randn
(opencood/utils/MDD_utils.py:232-235, opencood/models/gencomm_modules/cond_diff.py:307).

What is claimed and tested: z is N(0,1) to the resolution of a 1.4e8-sample Kolmogorov-Smirnov test; tail masses beyond 3 / 4 /
4.5 / 5 sigma within binomial error (round 2's generator stopped at 4.85 sigma); no lag, cross-channel, cross-agent or cross-step
correlation, also of the squares (the two branches of a Box-Muller pair go to adjacent channels of one pixel); the field the
kernels add is exactly fp16(sigma_t z) and, de-quantised by a uniform jitter inside its fp16 rounding interval, passes the same
KS test. What is NOT claimed: bit parity with any torch generator."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_AG, C, H, W = 16, 64, 200, 704   # the benchmark's launch: 16 agents of 64 x 200 x 704 = 1.44e8 samples per step


def _field(seed, t, sigma, unrounded, n=N_AG):
    from gencomm_amd import _lib
    from gencomm_amd.runtime import ptr, stream_ptr
    dev = torch.device(DEV)
    row = torch.tensor([0.0, 0.0, 0.0, 0.0, sigma], device=dev)
    out = torch.empty(n, C, H, W, device=dev)
    _lib.check(_lib.lib().gencomm_step_noise_fwd(ptr(row), seed, t, ptr(out), n, C, H, W, 1 if unrounded else 0, stream_ptr(dev)),
               "gencomm_step_noise_fwd")
    torch.cuda.synchronize()
    return out


def _ks(sample):
    """two-sided Kolmogorov-Smirnov distance of a 1-D float tensor to N(0,1), in float64 on the device"""
    x = torch.sort(sample.flatten()).values.double()
    n = x.numel()
    cdf = torch.special.ndtr(x)
    i = torch.arange(1, n + 1, device=x.device, dtype=torch.float64)
    d = torch.maximum((i / n - cdf).max(), (cdf - (i - 1) / n).max())
    return float(d), n


def _tail(p_two_sided_fn, z, thr):
    n = z.numel()
    cnt = int((z.abs() > thr).sum())
    p = math.erfc(thr / math.sqrt(2.0))
    return cnt, n * p, math.sqrt(n * p * (1 - p))


def test_step_field_is_standard_normal_on_1e8_samples():
    z = _field(seed=987654321, t=7, sigma=1.0, unrounded=True)
    n = z.numel()
    assert n >= 100_000_000
    d, _ = _ks(z)
    crit = 1.95 / math.sqrt(n)   # alpha = 1e-3
    print(f"KS distance of {n:.3e} step-noise samples to N(0,1): {d:.3e} (critical value at alpha 1e-3: {crit:.3e})")
    assert d < crit
    zd = z.double()
    m1, m2 = float(zd.mean()), float((zd * zd).mean())
    m3, m4 = float((zd ** 3).mean()), float((zd ** 4).mean())
    m6 = float((zd ** 6).mean())
    print(f"moments: mean {m1:.2e}, var {m2:.6f}, skew {m3:.2e}, kurt {m4:.5f}, 6th {m6:.4f}")
    assert abs(m1) < 5 / math.sqrt(n) and abs(m2 - 1) < 5 * math.sqrt(2 / n)
    assert abs(m3) < 5 * math.sqrt(15 / n) and abs(m4 - 3) < 5 * math.sqrt(96 / n) and abs(m6 - 15) < 5 * math.sqrt(10170 / n)
    for thr in (3.0, 4.0, 4.5, 5.0):
        cnt, exp, sd = _tail(None, z, thr)
        print(f"|z| > {thr}: {cnt} samples, expected {exp:.1f} +- {sd:.1f}")
        assert abs(cnt - exp) <= 4.5 * sd + 1, (thr, cnt, exp, sd)
    amax = float(z.abs().max())
    print(f"max |z| = {amax:.3f} (the 16-bit radius grid of round 2 stopped at 4.85)")
    assert 5.0 < amax < 8.3


def test_step_field_has_no_correlations():
    z = _field(seed=55, t=3, sigma=1.0, unrounded=True)
    n = z.numel()
    bound = 5 / math.sqrt(n)

    def corr(a, b):
        return float((a.double() * b.double()).mean())

    sq = z * z - 1.0
    checks = {
        "lag 1 in x": (z[..., :-1], z[..., 1:]),
        "lag 4 in x (next Philox counter)": (z[..., :-4], z[..., 4:]),
        "lag 1 in y": (z[:, :, :-1], z[:, :, 1:]),
        "adjacent channels (cos / sin of one pair)": (z[:, 0::2], z[:, 1::2]),
        "channel pairs": (z[:, :-2], z[:, 2:]),
        "agents": (z[:-1], z[1:]),
        "squares, adjacent channels": (sq[:, 0::2], sq[:, 1::2]),
        "squares, lag 1 in x": (sq[..., :-1], sq[..., 1:]),
    }
    for name, (a, b) in checks.items():
        c = corr(a, b)
        scale = 2.0 if name.startswith("squares") else 1.0   # var(z^2 - 1) = 2
        print(f"{name}: {c:.3e} (bound {scale * bound * math.sqrt(n / a.numel()):.3e})")
        assert abs(c) < scale * bound * math.sqrt(n / a.numel()), name
    other_t = _field(seed=55, t=4, sigma=1.0, unrounded=True)
    other_seed = _field(seed=56, t=3, sigma=1.0, unrounded=True)
    assert abs(corr(z, other_t)) < bound and abs(corr(z, other_seed)) < bound
    assert abs(corr(sq, other_t * other_t - 1.0)) < 2 * bound


def test_added_field_is_the_fp16_rounding_and_is_normal_after_dequantisation():
    sigma = 0.1473   # a mid-schedule sigma_t of the T = 20 schedule
    raw = _field(seed=31337, t=11, sigma=sigma, unrounded=True)
    nu = _field(seed=31337, t=11, sigma=sigma, unrounded=False)
    assert torch.equal(nu, raw.half().float())           # nu_t = fp16(sigma_t z), nothing else
    dev_abs = (nu - raw).abs()
    normal = raw.abs() >= 2.0 ** -14
    rel = float((dev_abs[normal] / raw.abs()[normal]).max())
    print(f"max relative rounding of the step noise (fp16 normal range): {rel:.3e} (fp16: 2^-11 = {2 ** -11:.3e}); "
          f"max absolute rounding below 2^-14: {float(dev_abs[~normal].max()) if bool((~normal).any()) else 0.0:.3e} (2^-25 = {2 ** -25:.3e})")
    assert rel <= 2 ** -11 * 1.0001
    assert not bool((~normal).any()) or float(dev_abs[~normal].max()) <= 2 ** -25 * 1.0001
    # variance carried by the rounding: E[(nu - sigma z)^2] / sigma^2
    print("relative variance added by the rounding: %.3e" % float(((nu - raw).double() ** 2).mean() / sigma ** 2))
    # de-quantise: uniform jitter inside each value's fp16 rounding interval, then the same KS test
    mant, expo = torch.frexp(nu.abs().clamp_min(2.0 ** -14))
    ulp = torch.ldexp(torch.ones_like(nu), expo - 11)      # spacing of fp16 at |nu| (2^-24 in the subnormal range)
    g = torch.Generator(device=DEV).manual_seed(1)
    jit = (torch.rand(nu.shape, device=DEV, generator=g) - 0.5) * ulp
    d, n = _ks((nu + jit) / sigma)
    crit = 1.95 / math.sqrt(n)
    print(f"KS distance of the de-quantised added field: {d:.3e} (critical {crit:.3e})")
    assert d < crit


def test_q_sample_noise_is_standard_normal():
    """eps of q_sample (the product kernel itself: zero x_start, schedule row {0, 1})."""
    from gencomm_amd import _lib
    from gencomm_amd.runtime import ptr, stream_ptr
    dev = torch.device(DEV)
    row = torch.tensor([0.0, 1.0, 0.0, 0.0, 0.0], device=dev)
    feat = torch.zeros(1, C, H, W, device=dev)
    rows = torch.zeros(N_AG, dtype=torch.int32, device=dev)
    out = torch.empty(N_AG, C, H, W, device=dev)
    _lib.check(_lib.lib().gencomm_q_sample_fwd(ptr(row), ptr(feat), 1, ptr(rows), None, 777, 20, ptr(out), N_AG, C, H, W, stream_ptr(dev)),
               "gencomm_q_sample_fwd")
    torch.cuda.synchronize()
    d, n = _ks(out)
    assert d < 1.95 / math.sqrt(n), d
    for thr in (3.0, 4.0, 4.5):
        cnt, exp, sd = _tail(None, out, thr)
        assert abs(cnt - exp) <= 4.5 * sd + 1, (thr, cnt, exp, sd)
    zf = out.flatten()
    assert abs(float((zf[:-1].double() * zf[1:].double()).mean())) < 5 / math.sqrt(n)
