"""CPU: the two polynomial GELU forms of the split-bf16 edge kernels (csrc/common.h: gelu_scaled, gelu_scaled_dgrad) are
re-evaluated here from the coefficients as they stand in the source -- fp32 Horner steps, exp2, the final fused step -- and
held against the exact erf form (PNEConvLayer.py:94-95: torch.nn.GELU()).  The kernels themselves are covered on the GPU by
the parity suite; this test pins the numbers, their accuracy claim and the behaviour beyond the fitted interval."""
import os
import re

import numpy as np
from scipy.special import ndtr
from scipy.stats import norm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "se3conv3d_amd", "csrc", "common.h")).read()
K_IN = 0.84932180028801904272  # kGeluIn: the kernels are handed x' = kGeluIn * x and return kGeluOut * GELU(x), 2 * GELU'(x)
assert f"{K_IN:.20f}"[:18] in SRC


def function_body(name):
    start = SRC.index(f"float {name}(float xp)")
    return SRC[start:SRC.index("\n}\n", start)]


def horner_coefficients(body, branch):
    """The fmaf chain of one #if branch: [c7, c6, ..., c0]."""
    text = body[body.index(branch) + len(branch):]
    text = text[:text.index("#e")]  # up to the #else / #endif that closes the branch
    first = re.search(r"float p = fmaf\((-?[\d.e+-]+)f, a, (-?[\d.e+-]+)f\);", text)
    rest = re.findall(r"p = fmaf\(p, a, (-?[\d.e+-]+)f\);", text)
    return [np.float32(first.group(1)), np.float32(first.group(2))] + [np.float32(c) for c in rest]


def horner(coeffs, a):
    p = np.full_like(a, coeffs[0])
    with np.errstate(over="ignore"):  # far beyond the fitted interval P overflows to -inf, as in the kernel: exp2 -> 0
        for c in coeffs[1:]:
            p = (p.astype(np.float64) * a + np.float64(c)).astype(np.float32)  # one fused multiply-add, rounded once
    return p


def gelu_value(xs, coeffs):
    xp = (xs * K_IN).astype(np.float32)
    a = np.abs(xp)
    q2 = np.exp2(horner(coeffs, a).astype(np.float64)).astype(np.float32)
    y2 = (-(a.astype(np.float64)) * q2 + (a + xp).astype(np.float32)).astype(np.float32)  # fmaf(-a, q2, a + xp)
    return y2.astype(np.float64) / (2.0 * K_IN)


def gelu_derivative_x2(xs, coeffs, a0):
    xp = (xs * K_IN).astype(np.float32)
    a = np.abs(xp)
    e = np.exp2(horner(coeffs, a).astype(np.float64)).astype(np.float32)
    g = ((a - np.float32(a0)).astype(np.float32).astype(np.float64) * e + 1.0).astype(np.float32)
    return (np.sign(xs) * g.astype(np.float64) + 1.0).astype(np.float32).astype(np.float64)


def test_value_polynomial_matches_erf_gelu_and_saturates():
    body = function_body("gelu_scaled")
    c7 = horner_coefficients(body, "#if SE3_GELU_POLY == 7")
    assert len(c7) == 8 and c7[0] < 0  # leading coefficient negative: exp2(P) -> 0 beyond the fitted interval
    c5 = horner_coefficients(body[body.index("#if SE3_GELU_POLY == 7"):], "#else")  # the shipped form (round 6)
    assert len(c5) == 6 and c5[0] < 0
    assert "#define SE3_GELU_POLY 5" in SRC
    for coeffs, max_mid, rms_mid, max_wide in ((c7, 6e-7, 1.2e-7, 1.2e-6), (c5, 9e-7, 3.5e-7, 1.2e-6)):
        xs = np.linspace(-3.0, 3.0, 240001)
        err = np.abs(gelu_value(xs, coeffs) - xs * ndtr(xs))
        assert err.max() < max_mid and np.sqrt(np.mean(err ** 2)) < rms_mid  # the rounding level of the erf form they replaced
        xs = np.linspace(-7.0, 7.0, 280001)
        assert np.abs(gelu_value(xs, coeffs) - xs * ndtr(xs)).max() < max_wide  # half an ulp of the values near 7
        big = np.array([8.0, 20.0, 1e3, 1e6, 1e12, 1e30])
        assert np.allclose(gelu_value(big, coeffs), big, rtol=2e-7) and np.all(np.abs(gelu_value(-big, coeffs)) < 1e-12)
    # the degree-5 P falls on the whole half line: nothing it returns beyond the fitted range can grow back
    a = np.linspace(0.0, 3000.0, 3000001)
    assert np.all(np.diff(np.polyval(np.array(c5, dtype=np.float64), a)) < 0)


def test_derivative_polynomial_matches_erf_gelu_and_saturates():
    body = function_body("gelu_scaled_dgrad")
    c7 = horner_coefficients(body, "#else")
    a0 = float(re.search(r"fmaf\(a - ([\d.e+-]+)f,", body).group(1))
    assert len(c7) == 8 and c7[0] < 0
    assert abs(a0 / K_IN - 0.7517915246935645) < 1e-7  # the fixed point of the Mills ratio: 2 Phi(-a) = 2 a pdf(a)
    xs = np.linspace(-3.0, 3.0, 240001)
    exact = 2.0 * (ndtr(xs) + xs * norm.pdf(xs))
    err = np.abs(gelu_derivative_x2(xs, c7, a0) - exact)
    assert err.max() < 6e-7 and np.sqrt(np.mean(err ** 2)) < 2e-7
    xs = np.linspace(-8.0, 8.0, 320001)
    assert np.abs(gelu_derivative_x2(xs, c7, a0) - 2.0 * (ndtr(xs) + xs * norm.pdf(xs))).max() < 6e-7
    big = np.array([9.0, 20.0, 1e3, 1e6, 1e12, 1e30])
    assert np.array_equal(gelu_derivative_x2(big, c7, a0), np.full(6, 2.0))
    assert np.array_equal(gelu_derivative_x2(-big, c7, a0), np.zeros(6))
