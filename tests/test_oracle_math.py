"""Accuracy of the pinned f32 elementary functions (oracle/kifs_oracle_math.c) against
float64.  The WGSL spec only bounds the builtins (log: 3 ULP outside [0.5,2], abs 2^-21
inside; sin/cos: abs 2^-11 on [-pi,pi]; acos, pow, log2 inherited); these pinned versions
must sit well inside those envelopes on the ranges the shader reaches."""
import ctypes as C
import math

import numpy as np
import pytest

F = np.float32


def ulp_err(got, want64):
    want32 = want64.astype(F)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / ulp


def call(oracle, name, xs, *extra):
    fn = getattr(oracle.lib(), name)
    return np.array([fn(float(x), *extra) for x in xs], dtype=F)


def test_logf_accuracy_and_specials(oracle):
    rng = np.random.default_rng(10)
    xs = np.concatenate([
        np.exp(rng.uniform(-80, 80, 20000)), rng.uniform(0.5, 2.0, 20000),
        rng.uniform(1e-45, 1e-38, 200),  # denormals
        [1.0, 2.0, 0.5, 1000.0, 1e6, np.finfo(F).max, np.finfo(F).tiny]]).astype(F)
    xs = xs[xs > 0]
    got = call(oracle, "kor_logf", xs)
    want = np.log(xs.astype(np.float64))
    far = (xs < 0.5) | (xs > 2.0)
    assert ulp_err(got[far], want[far]).max() < 1.0
    assert np.abs(got[~far].astype(np.float64) - want[~far]).max() < 2.0 ** -21
    L = oracle.lib()
    assert L.kor_logf(1.0) == 0.0
    assert L.kor_logf(0.0) == -math.inf and L.kor_logf(-0.0) == -math.inf
    assert L.kor_logf(math.inf) == math.inf
    assert math.isnan(L.kor_logf(-1.0)) and math.isnan(L.kor_logf(math.nan))


def test_log2_exp2_pow(oracle):
    rng = np.random.default_rng(11)
    xs = np.exp(rng.uniform(-60, 60, 20000)).astype(F)
    got = call(oracle, "kor_log2f", xs)
    want = np.log2(xs.astype(np.float64))
    assert (np.abs(got - want) <= 2.0 ** -21 + 3 * np.spacing(np.abs(want).astype(F))).all()
    es = rng.uniform(-126, 127, 20000).astype(F)
    got = call(oracle, "kor_exp2f", es)
    assert ulp_err(got, np.exp2(es.astype(np.float64))).max() < 2.0
    L = oracle.lib()
    assert L.kor_exp2f(0.0) == 1.0 and L.kor_exp2f(10.0) == 1024.0 and L.kor_exp2f(-1.0) == 0.5
    assert L.kor_exp2f(128.0) == math.inf and L.kor_exp2f(-200.0) == 0.0
    assert 0.0 < L.kor_exp2f(-140.0) < 1e-41  # denormal result
    # pow as the shader uses it: pow(norm, power) with norm in (0, 40], power in [1, 10]
    base = rng.uniform(0.05, 6.0, 5000).astype(F)
    for pw in (1.0, 2.0, 3.5, 10.0):
        got = call(oracle, "kor_powf", base, pw)
        want = np.power(base.astype(np.float64), pw)
        rel = np.abs(got - want) / want
        assert rel.max() < 6e-6, (pw, rel.max())  # (3 + 2|y log2 x|) ULP envelope of WGSL pow


def test_sin_cos(oracle):
    rng = np.random.default_rng(12)
    xs = np.concatenate([rng.uniform(-math.pi, math.pi, 20000), rng.uniform(-40, 40, 20000)]).astype(F)
    for name, ref in (("kor_sinf", np.sin), ("kor_cosf", np.cos)):
        got = call(oracle, name, xs)
        assert np.abs(got - ref(xs.astype(np.float64))).max() < 2.5e-7  # WGSL allows 2^-11
    L = oracle.lib()
    assert L.kor_sinf(0.0) == 0.0 and L.kor_cosf(0.0) == 1.0
    assert math.isnan(L.kor_sinf(math.inf)) and math.isnan(L.kor_cosf(math.nan))
    # beyond 2^20 the reduction is meaningless: defined as 0 (finite) / NaN (inf), no UB
    big = call(oracle, "kor_sinf", np.array([1e6, -1048576.0, -3e7, 1e20, 3e38], dtype=F))
    assert np.isfinite(big).all() and (np.abs(big) <= 1.0001).all() and (big[2:] == 0).all()


def test_acos(oracle):
    xs = np.concatenate([np.linspace(-1, 1, 20001), [0.5, -0.5, 0.500001, -0.500001]]).astype(F)
    got = call(oracle, "kor_acosf", xs)
    assert np.abs(got - np.arccos(xs.astype(np.float64))).max() < 4e-7
    L = oracle.lib()
    assert L.kor_acosf(1.0) == 0.0 and abs(L.kor_acosf(-1.0) - math.pi) < 3e-7
    assert math.isnan(L.kor_acosf(1.5)) and math.isnan(L.kor_acosf(math.nan))


def test_both_oracle_builds_are_identical(oracle):
    """The generic build (fmaf through libm) and the -mfma build must agree bit for bit."""
    import os
    import subprocess
    import sys
    code = ("import os,sys,hashlib; sys.path.insert(0, %r); import oracle as O;"
            "sc=O.screen_uniform(96,64); cam=O.camera_uniform(3.0,0.3,0.2);"
            "h=hashlib.sha256();"
            "[h.update(O.render(sc,cam,O.options_from_gui(fractal_group=g,primitive_shape=p,"
            "max_iterations=48),O.iters(10,4,8)).tobytes()) for g,p in ((1,0),(2,0),(0,4),(0,5))];"
            "print(O.lib()._variant, h.hexdigest())" % str(__import__('pathlib').Path(__file__).parent.parent))
    outs = []
    for env_extra in ({}, {"KIFS_ORACLE_GENERIC": "1"}):
        env = dict(os.environ, **env_extra)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, check=True,
                                   capture_output=True, text=True).stdout.split())
    assert outs[1][0] == "libkifs_oracle.so"
    assert outs[0][1] == outs[1][1]
