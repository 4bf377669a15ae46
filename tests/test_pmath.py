"""Accuracy of the portable elementary functions (rt_pmath.h / oracle/rt_pmath_oracle.h) against float64 numpy.

They stand in for the sinf / cosf / asinf / acosf / atan2f / logf of whichever libm the reference's compiler ships
(material.h, vec3.h, main.cu:120-124, 376-377); being one fixed formula on both sides is what makes the HIP path and
the oracle bit-comparable, being within a few ulp of the true function is what keeps them interchangeable with a libm.
"""
import numpy as np
import pytest


def ulp_err(got, exact):
    exact = np.asarray(exact, np.float64)
    got = np.asarray(got, np.float64)
    ulp = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
    return np.abs(got - exact) / ulp


def pm(ob, rt, fn, x, y=None):
    rec = np.zeros((len(x), 3), np.float32)
    rec[:, 0] = np.full(len(x), fn, np.int32).view(np.float32)
    rec[:, 1] = x
    if y is not None:
        rec[:, 2] = y
    return ob.debug_op(rt.OP_PMATH, rec)


def test_sincos_accuracy_and_identities(ob, rt):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-7, 7, 100000), rng.uniform(-200, 200, 50000), [0.0, np.pi / 2, np.pi, 6.2831855]]).astype(np.float32)
    out = pm(ob, rt, 0, x)
    s, c = out[:, 0], out[:, 1]
    xd = x.astype(np.float64)
    big = np.abs(np.sin(xd)) > 1e-3
    assert ulp_err(s[big], np.sin(xd[big])).max() <= 2.0
    big = np.abs(np.cos(xd)) > 1e-3
    assert ulp_err(c[big], np.cos(xd[big])).max() <= 2.0
    assert np.abs(s - np.sin(xd)).max() < 2e-7 and np.abs(c - np.cos(xd)).max() < 2e-7          # absolute, near the zeros
    assert np.abs(s.astype(np.float64) ** 2 + c.astype(np.float64) ** 2 - 1).max() < 4e-7
    o2 = pm(ob, rt, 0, -x)
    assert np.array_equal(o2[:, 0], -s) and np.array_equal(o2[:, 1], c)                          # odd / even, exactly
    z = pm(ob, rt, 0, np.float32([0.0]))
    assert z[0, 0] == 0.0 and z[0, 1] == 1.0


def test_asin_acos_accuracy(ob, rt):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-1, 1, 200000), [-1.0, -0.5, 0.0, 0.5, 1.0]]).astype(np.float32)
    a = pm(ob, rt, 1, x)[:, 0]
    c = pm(ob, rt, 2, x)[:, 0]
    xd = x.astype(np.float64)
    nz = np.abs(xd) > 1e-6
    assert ulp_err(a[nz], np.arcsin(xd[nz])).max() <= 3.0
    assert ulp_err(c, np.arccos(xd)).max() <= 2.5
    assert np.isnan(pm(ob, rt, 1, np.float32([1.5]))[0, 0]) and np.isnan(pm(ob, rt, 2, np.float32([-1.5]))[0, 0])
    assert a[-1] == np.float32(np.pi / 2) and c[-5] == np.float32(np.pi) and c[-1] == 0.0


def test_atan2_accuracy_and_quadrants(ob, rt):
    rng = np.random.default_rng(2)
    y = rng.normal(size=200000).astype(np.float32)
    x = rng.normal(size=200000).astype(np.float32)
    a = pm(ob, rt, 3, y, x)[:, 0]
    exact = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    ok = np.abs(exact) > 1e-3
    assert ulp_err(a[ok], exact[ok]).max() <= 4.0
    assert np.abs(a - exact).max() < 6e-7
    ys = np.float32([0, 0, 1, -1, 1, -1, 0.0])
    xs = np.float32([1, -1, 0, 0, -1, -1, 0.0])
    q = pm(ob, rt, 3, ys, xs)[:, 0]
    assert np.allclose(q, [0, np.pi, np.pi / 2, -np.pi / 2, 3 * np.pi / 4, -3 * np.pi / 4, 0], atol=3e-7)
    assert np.isnan(pm(ob, rt, 3, np.float32([np.nan]), np.float32([1.0]))[0, 0])


def test_log_accuracy_on_the_rng_range(ob, rt):
    rng = np.random.default_rng(3)
    u = ((rng.integers(0, 1 << 24, 300000) + 1) / 16777216.0).astype(np.float32)       # curand_uniform's (0, 1]
    x = np.concatenate([u, rng.uniform(1, 1e6, 50000).astype(np.float32), np.float32([1.0, 2 ** -24, 1e-40])])
    l = pm(ob, rt, 4, x)[:, 0]
    exact = np.log(x.astype(np.float64))
    ok = np.abs(exact) > 1e-4
    assert ulp_err(l[ok], exact[ok]).max() <= 1.5
    assert np.abs(l[~ok] - exact[~ok]).max() < 1e-7
    assert l[-3] == 0.0
    sp = pm(ob, rt, 4, np.float32([0.0, -1.0]))[:, 0]
    assert sp[0] == -np.inf and np.isnan(sp[1])
