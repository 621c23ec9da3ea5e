"""Pinning the CPU oracle (and with it the shared leaf arithmetic) against every
known-answer vector the reference's own tests hold for the path-integrator hot path
(SURVEY.md §8c). Each test cites the reference test it restates. CPU only.
"""
import ctypes as C
import math

import numpy as np
import pytest

from mitsuba2_amd._capi import c_float_p, c_u32_p, c_i32_p, c_double_p


def fp(a):
    return a.ctypes.data_as(c_float_p)


# ---- src/libcore/tests/test_random.py:6-28 -----------------------------------------------------
def test_tea_float32_exact(oracle):
    L = oracle.L
    kat = {(1, 1): 0.5424730777740479, (1, 2): 0.5079904794692993, (1, 3): 0.4171961545944214,
           (1, 4): 0.008385419845581055, (1, 5): 0.8085528612136841, (2, 1): 0.6939879655838013,
           (3, 1): 0.6978365182876587, (4, 1): 0.4897364377975464}
    for (a, b), v in kat.items():
        assert L.orc_tea_float32(a, b, 4) == np.float32(v)


def test_tea_float64_exact(oracle):
    L = oracle.L
    kat = {(1, 1): 0.5424730799533735, (1, 2): 0.5079905082233922, (1, 3): 0.4171962610608142,
           (1, 4): 0.008385529523330604, (1, 5): 0.80855288317879, (2, 1): 0.6939880404156831,
           (3, 1): 0.6978365636630994, (4, 1): 0.48973647949223253}
    for (a, b), v in kat.items():
        assert L.orc_tea_float64(a, b, 4) == v


# ---- PCG32: enoki::PCG32 is not vendored; pinned by O'Neill's public pcg32-demo vector ------------
def test_pcg32_demo_vector(oracle):
    out = np.zeros(6, np.uint32)
    oracle.L.orc_pcg32_u32(42, 54, out.ctypes.data_as(c_u32_p), 6)
    assert [hex(x) for x in out] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_pcg32_float_is_23_bit_uniform(oracle):
    u = np.zeros(4096, np.uint32); f = np.zeros(4096, np.float32)
    oracle.L.orc_pcg32_u32(7, 11, u.ctypes.data_as(c_u32_p), 4096)
    oracle.L.orc_pcg32_f32(7, 11, fp(f), 4096)
    expect = ((u >> 9) | 0x3f800000).view(np.float32) - np.float32(1.0)
    assert np.array_equal(f, expect) and f.min() >= 0 and f.max() < 1


# src/samplers/tests/test_independent.py:28-33 — the sampler is the default-seeded PCG32 stream
def test_independent_sampler_is_default_pcg32(native, oracle):
    s = native.Sampler()
    ref = np.zeros(64, np.float32)
    oracle.L.orc_pcg32_f32(0x853c49e6748fea9b, 0xda3e39cb94b95bdb, fp(ref), 64)
    assert [s.next_1d() for _ in range(64)] == list(ref)
    # seed(k) == PCG32(initstate = base_seed + k, default stream)   (sampler.cpp:83-96)
    s2 = native.Sampler(seed=5); s2.seed(1000)
    oracle.L.orc_pcg32_f32(1005, 0xda3e39cb94b95bdb, fp(ref), 64)
    assert [s2.next_1d() for _ in range(64)] == list(ref)


def test_morton_decode(oracle):
    xy = np.zeros(2, np.uint32)
    seen = set()
    for i in range(1024):
        oracle.L.orc_morton_decode(i, xy.ctypes.data_as(c_u32_p))
        x, y = int(xy[0]), int(xy[1])
        code = 0
        for b in range(5):
            code |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
        assert code == i and x < 32 and y < 32
        seen.add((x, y))
    assert len(seen) == 1024
    oracle.L.orc_morton_decode(0b1110, xy.ctypes.data_as(c_u32_p))
    assert tuple(xy) == (2, 3)


# ---- src/librender/tests/test_fresnel.py:7-40 --------------------------------------------------------
def _fresnel(oracle, c, eta):
    out = np.zeros(4, np.float32)
    oracle.L.orc_fresnel(C.c_float(c), C.c_float(eta), fp(out))
    return out


def test_fresnel_spot_checks(oracle):
    ct_crit = -math.sqrt(1 - 1 / 1.5 ** 2)
    ok = lambda a, b: np.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert ok(_fresnel(oracle, 1, 1.5), (0.04, -1, 1.5, 1 / 1.5))
    assert ok(_fresnel(oracle, -1, 1.5), (0.04, 1, 1 / 1.5, 1.5))
    assert ok(_fresnel(oracle, 1, 1 / 1.5), (0.04, -1, 1 / 1.5, 1.5))
    assert ok(_fresnel(oracle, -1, 1 / 1.5), (0.04, 1, 1.5, 1 / 1.5))
    assert ok(_fresnel(oracle, 0, 1.5), (1, ct_crit, 1.5, 1 / 1.5))
    assert ok(_fresnel(oracle, 0, 1 / 1.5), (1, 0, 1 / 1.5, 1.5))
    c45 = math.cos(math.radians(45))
    F, ct, _, scale = _fresnel(oracle, c45, 1.5)
    assert ok(ct, -math.cos(math.radians(28.1255057020557)))
    assert ok(F, 0.5 * (0.09201336304552442 ** 2 + 0.3033370452904235 ** 2))
    assert ok((scale * math.sqrt(1 - c45 ** 2)) ** 2 + ct ** 2, 1)
    F, ct, _, _ = _fresnel(oracle, c45, 1 / 1.5)
    assert ok(F, 1) and ok(ct, 0)
    c10 = math.cos(math.radians(10))
    F, ct, _, scale = _fresnel(oracle, c10, 1 / 1.5)
    assert ok(ct, -math.cos(math.radians(15.098086605159006)))
    assert ok(F, 0.5 * (0.19046797197779405 ** 2 + 0.20949431963852014 ** 2))


def test_fresnel_index_matched_and_conductor(oracle):
    # test_fresnel.py:53-60: eta = 1 -> F = 0, cos_theta_t = -cos_theta_i
    for c in np.linspace(-1, 1, 20):
        F, ct, _, _ = _fresnel(oracle, c, 1.0)
        assert F == 0 and abs(ct + c) < 5e-7
    # test_fresnel.py:63-76: conductor == dielectric for a real IOR
    oracle.L.orc_fresnel_conductor.argtypes = [C.c_float] * 3
    for eta in (1.5, 1 / 1.5):
        for c in np.cos(np.linspace(0, math.pi / 2, 20)):
            r = _fresnel(oracle, c, eta)[0]
            r2 = oracle.L.orc_fresnel_conductor(c, eta, 0.0)
            assert abs(r - r2) < 2e-6


# ---- src/bsdfs/tests/test_dielectric.py:35-174 (TransportMode::Radiance rows) ---------------------------
def _example_dielectric(native):
    return native.BSDF("dielectric", specular_reflectance=0.3, specular_transmittance=0.6, int_ior=1.5, ext_ior=1.0)


def test_dielectric_sample(native):
    b = _example_dielectric(native)
    DeltaReflection, DeltaTransmission = 0x20, 0x40
    r = b.sample([0, 0, 1], 0.0, [0, 0])
    assert np.allclose(r["weight"], [0.3] * 3) and np.isclose(r["pdf"], 0.04) and r["eta"] == 1.0
    assert np.allclose(r["wo"], [0, 0, 1]) and r["sampled_type"] == DeltaReflection
    r = b.sample([0, 0, 1], 0.05, [0, 0])
    assert np.allclose(r["weight"], [0.6 / 1.5 ** 2] * 3) and np.isclose(r["pdf"], 1 - 0.04) and np.isclose(r["eta"], 1.5)
    assert np.allclose(r["wo"], [0, 0, -1]) and r["sampled_type"] == DeltaTransmission
    # test03_sample_reverse
    r = b.sample([0, 0, -1], 0.0, [0, 0])
    assert np.allclose(r["weight"], [0.3] * 3) and np.isclose(r["pdf"], 0.04) and np.allclose(r["wo"], [0, 0, -1])
    r = b.sample([0, 0, -1], 0.05, [0, 0])
    assert np.allclose(r["weight"], [0.6 * 1.5 ** 2] * 3) and np.isclose(r["eta"], 1 / 1.5) and np.allclose(r["wo"], [0, 0, 1])
    e, p = b.eval_pdf([0, 0, 1], [0, 0, 1])
    assert np.all(e == 0) and p == 0                      # dielectric.cpp:312-320


def test_dielectric_spot_check_80_degrees(native):
    b = _example_dielectric(native)
    a = math.radians(80)
    wi = [math.sin(a), 0, math.cos(a)]
    r = b.sample(wi, 0.0, [0, 0])
    assert np.isclose(r["pdf"], 0.387704354691473, rtol=1e-5) and np.allclose(r["wo"], [-math.sin(a), 0, math.cos(a)], atol=1e-6)
    r = b.sample(wi, 1.0, [0, 0])
    a2 = math.radians(41.03641052520335)
    assert np.isclose(r["pdf"], 1 - 0.387704354691473, rtol=1e-5)
    assert np.allclose(r["wo"], [-math.sin(a2), 0, -math.cos(a2)], atol=1e-6)
    r2 = b.sample(r["wo"], 1.0, [0, 0])
    assert np.isclose(r2["pdf"], 1 - 0.387704354691473, rtol=1e-5) and np.allclose(r2["wo"], wi, atol=1e-6)


def test_dielectric_rejects_negative_ior(native):
    with pytest.raises(RuntimeError):
        native.BSDF("dielectric", int_ior=-0.5)


# ---- src/bsdfs/tests/test_diffuse.py:16-38 ------------------------------------------------------------------
def test_diffuse_eval_pdf(native):
    b = native.BSDF("diffuse")
    assert b.flags() == 0x2                               # DiffuseReflection
    for i in range(20):
        th = i / 19.0 * (math.pi / 2)
        wo = [math.sin(th), 0, math.cos(th)]
        e, p = b.eval_pdf([0, 0, 1], wo)
        assert np.isclose(p, max(wo[2], 0) / math.pi, atol=1e-7) and np.isclose(e[0], 0.5 * max(wo[2], 0) / math.pi, atol=1e-7)
    e, p = b.eval_pdf([0, 0, -1], [0, 0, 1])
    assert np.all(e == 0) and p == 0                      # FrontSide only


def test_diffuse_sample_is_cosine_weighted(native):
    b = native.BSDF("diffuse", reflectance=(0.2, 0.4, 0.6))
    rng = np.random.default_rng(0)
    zs = []
    for _ in range(2000):
        r = b.sample([0.3, 0.1, 0.9], rng.random(), rng.random(2))
        assert np.allclose(r["weight"], [0.2, 0.4, 0.6]) and np.isclose(r["pdf"], r["wo"][2] / math.pi, rtol=1e-6)
        assert abs(np.linalg.norm(r["wo"]) - 1) < 1e-5
        zs.append(r["wo"][2])
    assert abs(np.mean(zs) - 2 / 3) < 0.02                # E[cos] = 2/3 for cosine-weighted


# ---- miw/special.h: the shared exp / log / erf / erfinv stand-ins for Enoki's (not vendored) --------------------
def test_special_functions_accuracy(oracle):
    import scipy.special as sp
    def ev(x):
        out = np.zeros((len(x), 4), np.float32)
        for i, v in enumerate(x):
            oracle.L.orc_special(C.c_float(float(v)), fp(out[i]))
        return out
    x = np.linspace(-87, 88, 4001).astype(np.float32)
    assert np.max(np.abs(ev(x)[:, 0] / np.exp(x.astype(np.float64)) - 1)) < 2e-7                      # exp
    x = (10.0 ** np.linspace(-44, 38, 4001)).astype(np.float32); x = x[x > 0]
    lg = np.log(x.astype(np.float64))
    assert np.max(np.abs(ev(x)[:, 1] - lg) / np.maximum(np.abs(lg), 1e-3)) < 2e-7                     # log (incl. denormals)
    x = np.linspace(-6, 6, 6001).astype(np.float32)
    assert np.max(np.abs(ev(x)[:, 2] - sp.erf(x.astype(np.float64)))) < 1.2e-7                        # erf
    x = np.concatenate([np.linspace(-0.999999, 0.999999, 6001), [0.0]]).astype(np.float32)
    ref = sp.erfinv(x.astype(np.float64))
    assert np.max(np.abs(ev(x)[:, 3] - ref) / np.maximum(np.abs(ref), 1e-6)) < 6e-7                   # erfinv
    e = ev(np.array([np.inf, -np.inf, 89.0, -104.0, 0.0, 1.0, -1.0], np.float32))
    assert e[0, 0] == np.inf and e[1, 0] == 0 and e[2, 0] == np.inf and e[3, 0] == 0 and e[4, 0] == 1
    assert e[4, 1] == -np.inf and e[5, 1] == 0 and np.isnan(e[6, 1])
    assert e[0, 2] == 1 and e[1, 2] == -1 and e[5, 3] == np.inf and e[6, 3] == -np.inf


# ---- src/librender/tests/test_microfacet.py:18-207 (Beckmann rows) ------------------------------------------------
def _mfb(oracle, op, au, av, sv, wi, x):
    wi = np.asarray(wi, np.float32); x = np.asarray(x, np.float32); out = np.zeros(4, np.float32)
    oracle.L.orc_microfacet(op, 0, C.c_float(au), C.c_float(av), int(sv), fp(wi), fp(x), fp(out))
    return out


def test_beckmann_eval_pdf(oracle):
    """test02_eval_pdf_beckmann: anisotropic (0.1, 0.3) and isotropic 0.1, sample_visible = false."""
    steps = 20
    theta = np.linspace(0, math.pi, steps); phi = math.pi / 2
    v = [[math.cos(phi) * math.sin(t), math.sin(phi) * math.sin(t), math.cos(t)] for t in theta]
    ref_e = [1.06103287e+01, 8.22650051e+00, 3.57923722e+00, 6.84863329e-01, 3.26460004e-02, 1.01964230e-04, 5.87322635e-10] + [0] * 13
    ref_p = [1.06103287e+01, 8.11430168e+00, 3.38530421e+00, 6.02319300e-01, 2.57622823e-02, 6.90584930e-05, 3.21235011e-10] + [0] * 13
    ref_ie = [3.18309879e+01, 2.07673073e+00, 3.02855828e-04, 1.01591990e-11] + [0] * 16
    ref_ip = [3.18309879e+01, 2.04840684e+00, 2.86446273e-04, 8.93474877e-12] + [0] * 16
    wi = [0, 0, 1]
    assert np.allclose([_mfb(oracle, 0, 0.1, 0.3, False, wi, x)[0] for x in v], ref_e, rtol=1e-5, atol=1e-8)
    assert np.allclose([_mfb(oracle, 1, 0.1, 0.3, False, wi, x)[0] for x in v], ref_p, rtol=1e-5, atol=1e-8)
    assert np.allclose([_mfb(oracle, 0, 0.1, 0.1, False, wi, x)[0] for x in v], ref_ie, rtol=1e-5, atol=1e-8)
    assert np.allclose([_mfb(oracle, 1, 0.1, 0.1, False, wi, x)[0] for x in v], ref_ip, rtol=1e-5, atol=1e-8)
    phi = np.linspace(0, 2 * math.pi, steps); t = 0.1
    v = [[math.cos(p) * math.sin(t), math.sin(p) * math.sin(t), math.cos(t)] for p in phi]
    ref = [3.95569706, 4.34706259, 5.54415846, 7.4061389, 9.17129803, 9.62056446, 8.37803268, 6.42071199, 4.84459257, 4.05276537,
           4.05276537, 4.84459257, 6.42071199, 8.37803268, 9.62056446, 9.17129803, 7.4061389, 5.54415846, 4.34706259, 3.95569706]
    assert np.allclose([_mfb(oracle, 0, 0.1, 0.3, False, wi, x)[0] for x in v], ref, rtol=1e-5)
    assert np.allclose([_mfb(oracle, 1, 0.1, 0.3, False, wi, x)[0] for x in v], np.array(ref) * math.cos(0.1), rtol=1e-5)
    assert np.allclose([_mfb(oracle, 0, 0.1, 0.1, False, wi, x)[0] for x in v], 11.86709118, rtol=1e-5)
    assert np.allclose([_mfb(oracle, 1, 0.1, 0.1, False, wi, x)[0] for x in v], 11.86709118 * math.cos(0.1), rtol=1e-5)


def test_beckmann_smith_g1(oracle):
    """test03_smith_g1_beckmann"""
    steps = 20
    theta = np.linspace(math.pi / 3, math.pi / 2, steps); phi = math.pi / 2
    v = [[math.cos(phi) * math.sin(t), math.sin(phi) * math.sin(t), math.cos(t)] for t in theta]
    ref_a = [1.0000000e+00, 1.0000000e+00, 1.0000000e+00, 1.0000523e+00, 9.9941480e-01, 9.9757767e-01, 9.9420297e-01, 9.8884594e-01,
             9.8091525e-01, 9.6961778e-01, 9.5387781e-01, 9.3222123e-01, 9.0260512e-01, 8.6216795e-01, 8.0686140e-01, 7.3091686e-01,
             6.2609726e-01, 4.8074335e-01, 2.7883825e-01, 1.9197471e-06]
    ref_i = [1.0] * 14 + [9.9828446e-01, 9.8627287e-01, 9.5088160e-01, 8.5989666e-01, 6.2535185e-01, 5.7592310e-06]
    assert np.allclose([_mfb(oracle, 2, 0.1, 0.3, False, x, [0, 0, 1])[0] for x in v], ref_a, rtol=1e-5, atol=1e-5)
    assert np.allclose([_mfb(oracle, 2, 0.1, 0.1, False, x, [0, 0, 1])[0] for x in v], ref_i, rtol=1e-5, atol=1e-5)
    t = math.pi / 2 * 0.98; phi = np.linspace(0, 2 * math.pi, steps)
    v = [[math.cos(p) * math.sin(t), math.sin(p) * math.sin(t), math.cos(t)] for p in phi]
    ref = [0.67333597, 0.56164336, 0.42798978, 0.35298213, 0.31838724, 0.31201753, 0.33166203, 0.38421196, 0.48717275, 0.63746351,
           0.63746351, 0.48717275, 0.38421196, 0.33166203, 0.31201753, 0.31838724, 0.35298213, 0.42798978, 0.56164336, 0.67333597]
    assert np.allclose([_mfb(oracle, 2, 0.1, 0.3, False, x, [0, 0, 1])[0] for x in v], ref, rtol=2e-5, atol=1e-5)
    assert np.allclose([_mfb(oracle, 2, 0.1, 0.1, False, x, [0, 0, 1])[0] for x in v], 0.67333597, rtol=2e-5)


def test_beckmann_sample_table(oracle):
    """test04_sample_beckmann: anisotropic (0.1, 0.3), sample_visible = false, vs Mitsuba 0.6 data (first two rows of u2)."""
    u = np.linspace(0, 1, 6)
    u1, u2 = np.meshgrid(u, u)
    ref_m = np.array([[0, 0, 1], [4.71862517e-02, 1.23754589e-08, 9.98886108e-01], [7.12896436e-02, 1.86970155e-08, 9.97455657e-01],
                      [9.52876359e-02, 2.49909284e-08, 9.95449781e-01], [1.25854731e-01, 3.30077086e-08, 9.92048681e-01], [1, 2.6e-07, 0],
                      [0, 0, 1], [1.44650340e-02, 1.33556545e-01, 9.90935624e-01], [2.16356069e-02, 1.99762881e-01, 9.79605377e-01],
                      [2.85233315e-02, 2.63357669e-01, 9.64276493e-01], [3.68374363e-02, 3.40122312e-01, 9.39659417e-01], [1.07676744e-01, 9.94185984e-01, 0],
                      [0, 0, 1], [-3.80569659e-02, 8.29499215e-02, 9.95826781e-01], [-5.72742373e-02, 1.24836378e-01, 9.90522861e-01],
                      [-7.61397704e-02, 1.65956154e-01, 9.83189344e-01], [-9.96606201e-02, 2.17222810e-01, 9.71021116e-01], [-4.17001039e-01, 9.08905983e-01, 0]])
    ref_pdf = np.array([10.61032867, 8.51669121, 6.41503906, 4.302598, 2.17350101, 0., 10.61032867, 8.72333431, 6.77215099, 4.7335186,
                        2.55768704, 0., 10.61032867, 8.59542656, 6.55068302, 4.46557426, 2.31778312, 0.])
    for k in range(18):
        if k % 6 == 5:
            continue                                      # u1 = 1: log(0), cos_theta = 0 — direction defined only up to rounding
        r = _mfb(oracle, 3, 0.1, 0.3, False, [0, 0, 1], [u1.ravel()[k], u2.ravel()[k]])
        assert np.allclose(r[:3], ref_m[k], atol=5e-4), k
        assert np.isclose(r[3], ref_pdf[k], atol=1e-4 * max(1, ref_pdf[k])), k


def test_beckmann_visible_sampling_consistency(oracle):
    """Visible-normal Beckmann sampling (microfacet.h:359-395): unit normals in the upper hemisphere whose
    returned pdf equals D*G1*|wi.m|/cos(wi) (:300-303), and whose histogram mean matches the pdf-weighted mean."""
    rng = np.random.default_rng(3)
    for wi in ([0.6, 0.0, 0.8], [0.0, -0.95, 0.3122499], [0.0, 0.0, 1.0]):
        wi = np.asarray(wi, np.float32)
        for _ in range(200):
            r = _mfb(oracle, 3, 0.25, 0.25, True, wi, rng.random(2))
            m = r[:3]
            assert abs(np.linalg.norm(m) - 1) < 1e-5 and m[2] > 0 and np.isfinite(r).all()
            assert np.isclose(r[3], _mfb(oracle, 1, 0.25, 0.25, True, wi, m)[0], rtol=1e-5)


# ---- src/librender/tests/test_microfacet.py:209-313 (GGX rows) ---------------------------------------------------
def _mf(oracle, op, au, av, sv, wi, x):
    wi = np.asarray(wi, np.float32); x = np.asarray(x, np.float32); out = np.zeros(4, np.float32)
    oracle.L.orc_microfacet(op, 1, C.c_float(au), C.c_float(av), int(sv), fp(wi), fp(x), fp(out))
    return out


def test_ggx_smith_g1(oracle):
    steps = 20
    theta = np.linspace(math.pi / 3, math.pi / 2, steps)
    v = np.stack([np.zeros(steps) * np.sin(theta), np.sin(theta), np.cos(theta)], 1)      # phi = pi/2
    v[:, 0] = np.cos(math.pi / 2) * np.sin(theta)
    ref_aniso = [9.4031686e-01, 9.3310797e-01, 9.2485082e-01, 9.1534841e-01, 9.0435863e-01, 8.9158219e-01, 8.7664890e-01,
                 8.5909742e-01, 8.3835226e-01, 8.1369340e-01, 7.8421932e-01, 7.4880326e-01, 7.0604056e-01, 6.5419233e-01,
                 5.9112519e-01, 5.1425743e-01, 4.2051861e-01, 3.0633566e-01, 1.6765384e-01, 1.0861372e-06]
    ref_iso = [9.9261039e-01, 9.9160647e-01, 9.9042398e-01, 9.8901933e-01, 9.8733366e-01, 9.8528832e-01, 9.8277503e-01,
               9.7964239e-01, 9.7567332e-01, 9.7054905e-01, 9.6378750e-01, 9.5463598e-01, 9.4187391e-01, 9.2344058e-01,
               8.9569420e-01, 8.5189372e-01, 7.7902949e-01, 6.5144652e-01, 4.1989169e-01, 3.2584082e-06]
    got_a = [_mf(oracle, 2, 0.1, 0.3, False, v[i], [0, 0, 1])[0] for i in range(steps)]
    got_i = [_mf(oracle, 2, 0.1, 0.1, False, v[i], [0, 0, 1])[0] for i in range(steps)]
    assert np.allclose(got_a, ref_aniso, rtol=1e-5, atol=1e-5) and np.allclose(got_i, ref_iso, rtol=1e-5, atol=1e-5)
    theta = math.pi / 2 * 0.98
    phi = np.linspace(0, 2 * math.pi, steps)
    ref = [0.46130955, 0.36801264, 0.26822716, 0.21645154, 0.19341162, 0.18922243, 0.20219423, 0.23769052, 0.31108665,
           0.43013984, 0.43013984, 0.31108665, 0.23769052, 0.20219423, 0.18922243, 0.19341162, 0.21645154, 0.26822716,
           0.36801264, 0.46130955]
    got = [_mf(oracle, 2, 0.1, 0.3, False, [math.cos(p) * math.sin(theta), math.sin(p) * math.sin(theta), math.cos(theta)],
               [0, 0, 1])[0] for p in phi]
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-5)
    got = [_mf(oracle, 2, 0.1, 0.1, False, [math.cos(p) * math.sin(theta), math.sin(p) * math.sin(theta), math.cos(theta)],
               [0, 0, 1])[0] for p in phi]
    assert np.allclose(got, 0.46130955, rtol=2e-5)


def test_ggx_sample_table(oracle):
    """test05_sample_ggx: anisotropic GGX (0.1, 0.3), sample_visible = false, vs Mitsuba 0.6 data."""
    u = np.linspace(0, 1, 6)
    u1, u2 = np.meshgrid(u, u)
    ref_m = np.array([[0, 0, 1], [4.99384739e-02, 1.30972797e-08, 9.98752296e-01], [8.13788623e-02, 2.13430980e-08, 9.96683240e-01],
                      [1.21566132e-01, 3.18829443e-08, 9.92583334e-01], [1.96116075e-01, 5.14350340e-08, 9.80580688e-01],
                      [1, 2.62268316e-07, 0], [0, 0, 1], [1.52942007e-02, 1.41212299e-01, 9.89861190e-01],
                      [2.45656986e-02, 2.26816610e-01, 9.73627627e-01], [3.57053429e-02, 3.29669625e-01, 9.43420947e-01],
                      [5.36015145e-02, 4.94906068e-01, 8.67291689e-01], [1.07676744e-01, 9.94185984e-01, 0]])
    ref_pdf = np.array([10.61032867, 6.81609201, 3.85797882, 1.73599267, 0.45013079, 0., 10.61032867, 7.00141668, 4.13859272,
                        2.02177191, 0.65056872, 0.])
    for k in range(12):
        r = _mf(oracle, 3, 0.1, 0.3, False, [0, 0, 1], [u1.ravel()[k], u2.ravel()[k]])
        if k % 6 == 5:
            continue                                      # u1 = 1: cos_theta = 0, direction undefined up to rounding
        assert np.allclose(r[:3], ref_m[k], atol=5e-4), k
        assert np.isclose(r[3], ref_pdf[k], atol=1e-3), k


def test_ggx_eval_normalisation_and_visible_sampling(oracle):
    """D(m) cos integrates to 1; the visible-normal sampler's pdf matches D*G1*|wi.m|/cos (microfacet.h:300-303)."""
    n = 200
    th = (np.arange(n) + 0.5) / n * (math.pi / 2); ph = (np.arange(2 * n) + 0.5) / (2 * n) * 2 * math.pi
    T, P = np.meshgrid(th, ph)
    acc = 0.0
    for t, p in zip(T.ravel()[::7], P.ravel()[::7]):
        m = [math.sin(t) * math.cos(p), math.sin(t) * math.sin(p), math.cos(t)]
        acc += _mf(oracle, 0, 0.5, 0.5, True, [0, 0, 1], m)[0] * math.cos(t) * math.sin(t)
    acc *= (math.pi / 2 / n) * (2 * math.pi / (2 * n)) * 7
    assert abs(acc - 1) < 0.02
    rng = np.random.default_rng(1)
    wi = np.array([0.6, 0.0, 0.8], np.float32)
    for _ in range(200):
        r = _mf(oracle, 3, 0.3, 0.3, True, wi, rng.random(2))
        m = r[:3]
        assert abs(np.linalg.norm(m) - 1) < 1e-5 and m[2] > 0
        assert np.isclose(r[3], _mf(oracle, 1, 0.3, 0.3, True, wi, m)[0], rtol=1e-5)


# ---- src/librender/tests/test_mesh.py:257-299 -------------------------------------------------------------------------
def test_mesh_ray_intersect_triangle(oracle):
    # rectangle.obj is absent (data submodule); these two triangles of the [-1,1]^2 quad are the
    # unique ones consistent with the test's prim_uv values (0.35, 0.3) / (0.3, 0.35).
    tris = [np.array([-1, 1, 0, 1, -1, 0, -1, -1, 0], np.float32), np.array([1, -1, 0, 1, 1, 0, -1, 1, 0], np.float32)]
    out = np.zeros(4, np.float32)
    oracle.L.orc_ray_triangle(fp(tris[0]), fp(np.array([-0.3, -0.3, -10, 0, 0, 1, 0, np.inf], np.float32)), fp(out))
    assert out[0] == 1 and np.isclose(out[1], 10) and np.allclose(out[2:], [0.35, 0.3])
    oracle.L.orc_ray_triangle(fp(tris[1]), fp(np.array([0.3, 0.3, -10, 0, 0, 1, 0, np.inf], np.float32)), fp(out))
    assert out[0] == 1 and np.isclose(out[1], 10) and np.allclose(out[2:], [0.3, 0.35])
    # no backface culling, interval test (mesh.h:210-217)
    oracle.L.orc_ray_triangle(fp(tris[0]), fp(np.array([-0.3, -0.3, 10, 0, 0, -1, 0, np.inf], np.float32)), fp(out))
    assert out[0] == 1
    oracle.L.orc_ray_triangle(fp(tris[0]), fp(np.array([-0.3, -0.3, -10, 0, 0, 1, 0, 9.9], np.float32)), fp(out))
    assert out[0] == 0
    oracle.L.orc_ray_triangle(fp(tris[0]), fp(np.array([-0.3, -0.3, -10, 0, 0, 1, 10.1, np.inf], np.float32)), fp(out))
    assert out[0] == 0
    oracle.L.orc_ray_triangle(fp(tris[0]), fp(np.array([0.3, 0.3, -10, 0, 0, 1, 0, np.inf], np.float32)), fp(out))
    assert out[0] == 0


def test_surface_interaction_fields(native, oracle):
    """interaction.h:571-596 + mesh.cpp:449-545: p from barycentrics, unit normals, orthonormal frame, wi local."""
    from mitsuba2_amd import api
    v = np.array([[-1, 1, 0], [1, -1, 0], [-1, -1, 0]], np.float32)
    scene = api.Scene([api.Mesh("t", v, [[0, 1, 2]])]).build(-1)
    ok, si = oracle.ray_intersect_full(scene.desc(), [-0.3, -0.3, -10, 0, 0, 1, 0, np.inf])
    assert ok == 1 and np.isclose(si[0], 10) and np.allclose(si[1:4], [-0.3, -0.3, 0], atol=1e-6)
    n, sh_n, s, t, wi = si[4:7], si[7:10], si[10:13], si[13:16], si[16:19]
    assert np.allclose(np.abs(n), [0, 0, 1]) and np.allclose(n, sh_n)
    assert abs(np.dot(s, t)) < 1e-6 and abs(np.dot(s, n)) < 1e-6 and np.allclose(np.cross(sh_n, s), t, atol=1e-6)
    assert np.isclose(wi[2], np.dot([0, 0, -1], n)) and np.allclose(si[19:21], [0.35, 0.3])
    ok, si = oracle.ray_intersect_full(scene.desc(), [5, 5, -10, 0, 0, 1, 0, np.inf])
    assert ok == 0 and np.isinf(si[0]) and np.allclose(si[16:19], [0, 0, -1])     # scene_native.inl:34-38


# ---- src/librender/tests/test_kdtrees.py:26-59 (stairs) on the CPU ----------------------------------------------------------
def test_stairs_bvh_equals_brute_force_equals_formula(native, oracle):
    from mitsuba2_amd import scenes, api
    n_steps = 20
    v, f = scenes.stairs(n_steps)
    scene = api.Scene([api.Mesh("stairs", v, f)]).build(-1)
    n = 64; inv_n = 1.0 / (n - 1)
    xs, ys = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    o = np.stack([xs.ravel() * inv_n, ys.ravel() * inv_n, np.full(xs.size, 2.0)], 1).astype(np.float32)
    d = np.tile(np.array([0, 0, -1], np.float32), (len(o), 1))
    naive = oracle.trace(scene.desc(), o, d, 0.0, 100.0)
    for max_leaf in (1, 4, 16):
        tree = oracle.emu_trace(scene.desc(), o, d, 0.0, 100.0, max_leaf=max_leaf)
        assert np.array_equal(tree["prim"], naive["prim"]) and np.array_equal(tree["t"].view(np.uint32), naive["t"].view(np.uint32))
        assert tree["bvh"][2] <= 62
    expected = 2.0 - np.floor((ys.ravel() * inv_n) * n_steps) / n_steps
    assert np.allclose(naive["t"], expected, atol=1e-6)
    shadow = oracle.emu_trace(scene.desc(), o, d, 0.0, 100.0, any_hit=True)
    assert np.all(shadow["t"] == 0)


def test_bvh_equals_brute_force_random_soup(native, oracle):
    from mitsuba2_amd import scenes, api
    v, f = scenes.random_triangles(1500, seed=3)
    scene = api.Scene([api.Mesh("soup", v, f)]).build(-1)
    rng = np.random.default_rng(4)
    n = 3000
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50, 0] = 0; d[50:100, 1] = 0; d[100:120] = [0, 0, 1]         # axis-parallel rays
    a = oracle.trace(scene.desc(), o, d, 1e-4, np.inf); b = oracle.emu_trace(scene.desc(), o, d, 1e-4, np.inf)
    assert np.array_equal(a["prim"], b["prim"]) and (a["prim"] != 0xffffffff).mean() > 0.1
    hit = a["prim"] != 0xffffffff
    for k in ("t", "u", "v"):
        assert np.array_equal(a[k][hit].view(np.uint32), b[k][hit].view(np.uint32))
    a = oracle.trace(scene.desc(), o, d, 1e-4, 0.5, any_hit=True); b = oracle.emu_trace(scene.desc(), o, d, 1e-4, 0.5, any_hit=True)
    assert np.array_equal(a["t"], b["t"])


def test_closest_hit_tie_break_is_smallest_prim(native, oracle):
    """Two coplanar coincident triangles: the smaller global primitive id wins (this code base's definition)."""
    from mitsuba2_amd import api
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    v = np.concatenate([tri, tri, tri + [0, 0, 1]]); f = np.arange(9, dtype=np.uint32).reshape(3, 3)
    scene = api.Scene([api.Mesh("dup", v, f)]).build(-1)
    o = np.array([[0.2, 0.2, -1.0]], np.float32); d = np.array([[0, 0, 1.0]], np.float32)
    assert oracle.trace(scene.desc(), o, d)["prim"][0] == 0 and oracle.emu_trace(scene.desc(), o, d, max_leaf=1)["prim"][0] == 0


# ---- src/librender/tests/test_imageblock.py:146-220 ------------------------------------------------------------------------------
def _film_job(native, w, h, **kw):
    from mitsuba2_amd import scenes
    sensor = scenes.cornell_sensor(w, h, 1, **kw)
    return sensor, native.PathIntegrator(block_size=32).render_job(sensor)


def test_imageblock_put_gaussian_vs_reference_loop(native, oracle):
    sensor, job = _film_job(native, 12, 12)
    cfg = job.cfg
    lut = np.array(list(cfg.filter_lut))
    radius, border = 2, cfg.filter_border
    assert cfg.filter_radius == 2.0 and border == 2

    def eval_discretized(x):
        return lut[min(int(abs(x * (31 / cfg.filter_radius))), 31)]
    positions = np.array([[5, 6], [0, 1], [5, 6], [1, 11], [11, 11], [0, 1], [2, 5], [4, 1], [0, 11], [5, 4]], np.float64)
    positions += np.random.default_rng(0).uniform(0, 0.95, positions.shape)
    positions = positions.astype(np.float32)
    n = len(positions)
    values = np.concatenate([np.arange(n * 3).reshape(n, 3), np.ones((n, 2))], 1).astype(np.float32)
    ref = np.zeros((12 + 2 * border, 12 + 2 * border, 5))
    for i in range(n):
        pos = positions[i].astype(np.float64) - 0.5 + border
        lo = np.ceil(pos - radius).astype(int); hi = np.floor(pos + radius).astype(int)
        for dy in range(lo[1], hi[1] + 1):
            for dx in range(lo[0], hi[0] + 1):
                if dx < 0 or dy < 0 or dx >= ref.shape[1] or dy >= ref.shape[0]:
                    continue
                w = eval_discretized(dx - pos[0]) * eval_discretized(dy - pos[1])
                ref[dy, dx] += w * values[i]
    out = np.zeros(ref.size, np.float32)
    got = oracle.L.orc_imageblock_put(C.byref(cfg), 0, 0, 12, 12, border, fp(positions), fp(values), n, fp(out))
    assert got == ref.size
    assert np.allclose(out.reshape(ref.shape), ref, atol=1e-5)
    # the product's shared splat helper (miw/film.h), clipped to the film, must agree with the block interior
    film64 = np.zeros(12 * 12 * 5)
    oracle.L.orc_film_splat_shared(C.byref(cfg), fp(positions), fp(values), n, film64.ctypes.data_as(c_double_p))
    assert np.allclose(film64.reshape(12, 12, 5), ref[border:-border, border:-border], atol=1e-5)


def test_gaussian_filter_table(native):
    """src/rfilters/gaussian.cpp:32-47 + src/libcore/rfilter.cpp:9-20"""
    sensor, job = _film_job(native, 8, 8)
    lut = np.array(list(job.cfg.filter_lut), np.float64)
    alpha, radius = -1.0 / (2 * 0.25), 2.0
    bias = math.exp(alpha * radius * radius)
    expect = [max(0.0, math.exp(alpha * (radius * i / 31) ** 2) - bias) for i in range(31)] + [0.0]
    assert np.allclose(lut, expect, atol=1e-7) and lut[31] == 0 and job.cfg.filter_border == 2
    sensor, job = _film_job(native, 8, 8, rfilter="box")
    assert job.cfg.filter_border == 0 and abs(job.cfg.filter_radius - 0.5) < 1e-3 and list(job.cfg.filter_lut)[:31] == [1.0] * 31


# ---- src/librender/tests/test_spiral.py:49-80 -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["host", "oracle"])
def test_spiral(native, oracle, which):
    def blocks(w, h, bs=32, off=(0, 0)):
        if which == "host":
            return native.spiral(w, h, bs, off)
        n = ((w + bs - 1) // bs) * ((h + bs - 1) // bs)
        out = np.zeros((n, 5), np.int32)
        assert oracle.L.orc_spiral(w, h, off[0], off[1], bs, out.ctypes.data_as(c_i32_p), n) == n
        return out
    b = blocks(15, 12)                                    # test02_small_film
    assert len(b) == 1 and list(b[0]) == [0, 0, 15, 12, 0]
    b = blocks(318, 322)                                  # test03_normal_film
    assert len(b) == 110
    c, w = np.array([160, 160]), 32
    expected = [c, c + [w, 0], c + [w, w], c + [0, w], c + [-w, w], c + [-w, 0], c + [-w, -w], c + [0, -w], c + [w, -w],
                c + [2 * w, -w], c + [2 * w, 0], c + [2 * w, w]]
    for i, e in enumerate(expected):
        assert list(b[i][:2]) == list(e) and list(b[i][2:4]) == [32, 32] and b[i][4] == i
    # every block exactly once, ids 0..n-1, edge blocks clipped
    assert sorted(b[:, 4]) == list(range(110))
    assert len({(x, y) for x, y in b[:, :2]}) == 110
    assert b[:, 2].min() == 318 - 9 * 32 and b[:, 3].min() == 322 - 10 * 32
    b = blocks(64, 64, 32, (5, 7))                        # crop offset is added to block offsets (spiral.cpp:45)
    assert sorted(map(tuple, b[:, :2])) == [(5, 7), (5, 39), (37, 7), (37, 39)]


def test_host_and_oracle_spiral_agree(native, oracle):
    for w, h, bs in [(1920, 1080, 32), (256, 256, 32), (77, 45, 16), (33, 1, 32)]:
        a = native.spiral(w, h, bs)
        out = np.zeros_like(a)
        oracle.L.orc_spiral(w, h, 0, 0, bs, out.ctypes.data_as(c_i32_p), len(a))
        assert np.array_equal(a, out)


# ---- src/sensors/tests/test_perspective.py --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("origin", [[1.0, 0.0, 1.5], [1.0, 4.0, 1.5]])
@pytest.mark.parametrize("direction", [[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
def test_perspective_sample_ray(native, origin, direction):
    from mitsuba2_amd import api
    film = api.Film(width=512, height=256); sampler = api.Sampler()
    t = [origin[i] + direction[i] for i in range(3)]
    for fov in (34.0, 80.0):
        cam = api.Sensor(film, sampler, fov=fov, near_clip=1.0, far_clip=35.0, to_world=dict(origin=origin, target=t, up=(0, 1, 0)))
        r = cam.sample_ray(0.5, 0.5)
        assert np.allclose(r[0:3], origin) and np.allclose(r[3:6], direction, atol=1e-6)
        assert np.isclose(r[6], 1.0, rtol=1e-6) and np.isclose(r[7], 35.0, rtol=1e-5)
        # test04_fov_axis: the edge of the film along x is fov/2 away from the optical axis
        for x in (0.0, 1.0):
            r = cam.sample_ray(x, 0.5)
            ang = math.degrees(math.acos(np.clip(np.dot(r[3:6], direction), -1, 1)))
            assert abs(ang - fov / 2) < 1e-3
            assert np.isclose(r[6], 1.0 / math.cos(math.radians(fov / 2)), rtol=1e-5)   # mint = near / d.z


def test_parse_fov_and_errors(native):
    from mitsuba2_amd import api
    film = api.Film(width=512, height=256); sampler = api.Sampler()
    cam = api.Sensor(film, sampler, fov=40.0, fov_axis="y")
    assert np.isclose(cam.x_fov(), math.degrees(2 * math.atan(math.tan(math.radians(20)) * 2)), rtol=1e-6)
    cam = api.Sensor(film, sampler)                        # default focal_length 50mm, diagonal (sensor.cpp:132-147)
    diag = 2 * math.degrees(math.atan(math.sqrt(36 * 36 + 24 * 24) / 100))
    width = 2 * math.tan(math.radians(diag) / 2) / math.sqrt(1 + 1 / 4)
    assert np.isclose(cam.x_fov(), math.degrees(2 * math.atan(width / 2)), rtol=1e-5)
    with pytest.raises(RuntimeError):
        api.Sensor(film, sampler, fov=40.0, focal_length="50mm")
    with pytest.raises(RuntimeError):
        api.Sensor(film, sampler, fov=40.0, near_clip=5.0, far_clip=1.0)
    with pytest.raises(RuntimeError):
        api.Sensor(film, sampler, fov=200.0)


# ---- src/libcore/tests/test_distr_2d.py:8-47 (Hierarchical2D0 spot checks vs Mathematica) -----------------------
def _h2d(oracle, data, op, xy):
    d = np.ascontiguousarray(data, np.float32); out = np.zeros(3, np.float32)
    rc = oracle.L.orc_hier2d(fp(d), d.shape[1], d.shape[0], op, fp(np.asarray(xy, np.float32)), fp(out))
    assert rc == 0
    return out


def _bilinear_to_square(v00, v10, v01, v11, s):
    """include/mitsuba/core/warp.h:368-426 in float64 (the reference test computes its expectations with it)"""
    def lerp(a, b, t): return a + (b - a) * t
    def lin2int(v0, v1, x): return x * ((2 - x) * v0 + x * v1) / (v0 + v1) if abs(v0 - v1) > 1e-4 * (v0 + v1) else x
    r0, r1 = v00 + v10, v01 + v11
    c0, c1 = lerp(v00, v01, s[1]), lerp(v10, v11, s[1])
    pdf = lerp(c0, c1, s[0])
    return [lin2int(c0, c1, s[0]), lin2int(r0, r1, s[1])], pdf


def test_hierarchical2d_spot_checks(oracle):
    ref = np.array([[1, 2, 5], [9, 7, 2]], np.float32)            # mismatched X/Y resolution, odd number of columns
    intg = np.array([19, 16]) / 35
    a = lambda x, y: np.allclose(x, y, atol=1e-6)
    assert a(_h2d(oracle, ref, 0, [0, 0]), [0, 0, 8.0 / 35.0])
    assert a(_h2d(oracle, ref, 0, [1, 1]), [1, 1, 16.0 / 35.0])
    assert a(_h2d(oracle, ref, 0, [intg[0], 0]), [0.5, 0, 16.0 / 35.0])
    s, pdf = _bilinear_to_square(1, 2, 9, 7, [0.4, 0.3])
    s[0] *= intg[0]; pdf *= 8.0 / 35.0
    assert a(_h2d(oracle, ref, 0, s), [0.2, 0.3, pdf]) and a(_h2d(oracle, ref, 1, [0.2, 0.3])[0], pdf)
    s, pdf = _bilinear_to_square(2, 5, 7, 2, [0.4, 0.3])
    s[0] = s[0] * intg[1] + intg[0]; pdf *= 8.0 / 35.0
    assert a(_h2d(oracle, ref, 0, s), [0.7, 0.3, pdf]) and a(_h2d(oracle, ref, 1, [0.7, 0.3])[0], pdf)


def test_hierarchical2d_sample_matches_its_density(oracle):
    """forward warp + eval are consistent on a random 7 x 5 grid: returned pdf == eval(position), positions in [0,1]^2,
    and the warp is a bijection of strata (histogram of warped stratified samples ~ integral of the density)."""
    rng = np.random.default_rng(5)
    data = (rng.random((5, 7)) * 10).astype(np.float32)
    n = 64
    hist = np.zeros((4, 6))
    for i in range(n):
        for j in range(n):
            r = _h2d(oracle, data, 0, [(i + 0.5) / n, (j + 0.5) / n])
            assert 0 <= r[0] <= 1 and 0 <= r[1] <= 1
            assert np.isclose(r[2], _h2d(oracle, data, 1, r[:2])[0], rtol=2e-4)
            hist[min(int(r[1] * 4), 3), min(int(r[0] * 6), 5)] += 1
    avg = (data[:-1, :-1] + data[:-1, 1:] + data[1:, :-1] + data[1:, 1:]) / 4
    assert np.allclose(hist / n ** 2, avg / avg.sum(), atol=4e-3)


def test_inverse_trig_accuracy(oracle):
    x = np.linspace(-1, 1, 4001).astype(np.float32)
    y = np.random.default_rng(2).uniform(-3, 3, 4001).astype(np.float32)
    got = oracle.eval(10, np.stack([y, x], 1))
    assert np.max(np.abs(got[:, 0] - np.arctan2(y.astype(np.float64), x.astype(np.float64)))) < 4e-7
    assert np.max(np.abs(got[:, 1] - np.arccos(x.astype(np.float64)))) < 4e-7
    assert np.max(np.abs(got[:, 2] - np.arcsin(x.astype(np.float64)))) < 3e-7
    sp = oracle.eval(10, np.array([[0.0, -1.0], [-0.0, -1.0], [0.0, 1.0], [1.0, 0.0], [-1.0, 0.0]], np.float32))[:, 0]
    assert np.allclose(sp, [np.pi, -np.pi, 0.0, np.pi / 2, -np.pi / 2], atol=1e-7)


@pytest.mark.parametrize("direction", [[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
@pytest.mark.parametrize("fov", [34.0, 80.0])
def test_perspective_fov_axis(native, direction, fov):
    """src/sensors/tests/test_perspective.py:133-163 (test04_fov_axis): film aspect 1.5; the film's edge along the chosen
    axis (its corners for `diagonal`) is fov / 2 away from the optical axis"""
    from mitsuba2_amd import api
    origin = [1.0, 0.0, 1.5]
    film = api.Film(width=768, height=512); sampler = api.Sampler()
    t = [origin[i] + direction[i] for i in range(3)]

    def check(axis, samples):
        cam = api.Sensor(film, sampler, fov=fov, fov_axis=axis, near_clip=1.0, far_clip=35.0, to_world=dict(origin=origin, target=t, up=(0, 1, 0)))
        for sx, sy in samples:
            r = cam.sample_ray(sx, sy)
            ang = math.degrees(math.acos(np.clip(np.dot(r[3:6], direction), -1, 1)))
            assert abs(ang - fov / 2) < 2e-3, (axis, sx, sy, ang)
    for axis in ("x", "larger"):
        check(axis, [(0.0, 0.5), (1.0, 0.5)])
    for axis in ("y", "smaller"):
        check(axis, [(0.5, 0.0), (0.5, 1.0)])
    check("diagonal", [(0.0, 0.0), (0.0, 1.0), (1.0, 0.0), (1.0, 1.0)])


def test_hier2d_sample_through_a_copy_of_its_top_levels(oracle):
    """envmap.h EnvTop (round 4): the device kernels read the warp's smallest levels from a copy in LDS. The two-loop form of
    hier2d_sample must give the sample of the one-array form for every number of copied levels, bit for bit (the checker poisons
    the originals of the copied levels, so a read from the wrong array cannot go unnoticed)."""
    rng = np.random.default_rng(8)
    for (h, w) in ((16, 32), (33, 20), (7, 50)):
        d = (rng.random((h, w)) ** 3 + 0.01).astype(np.float32)
        pts = rng.random((40, 2)).astype(np.float32)
        ref = []
        for xy in pts:
            out = np.zeros(3, np.float32)
            assert oracle.L.orc_hier2d(fp(d), w, h, 0, fp(np.ascontiguousarray(xy)), fp(out)) == 0
            ref.append(out.copy())
        k = 1
        while True:
            out = np.zeros(3, np.float32)
            rc = oracle.L.orc_hier2d(fp(d), w, h, 100 + k, fp(np.ascontiguousarray(pts[0])), fp(out))
            if rc == -2:
                break
            assert rc == 0
            for xy, want in zip(pts, ref):
                out = np.zeros(3, np.float32)
                assert oracle.L.orc_hier2d(fp(d), w, h, 100 + k, fp(np.ascontiguousarray(xy)), fp(out)) == 0
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (h, w, k, xy)
            k += 1
        assert k >= 4



def test_hier2d_two_levels_per_lookup_equals_the_one_level_descent(oracle):
    """envmap.h MIW_ENV_PAIRED (round 5): hier2d_sample recomputes a level's block from the four child blocks of the level below
    (the constructor's own float32 sums, distr_2d.h:445-461) and reads the bilinear patch's corners with level 1 — half the
    dependent lookups of the device's shade body. Same sample and pdf as the one-level-per-lookup descent (op 200), bit for bit:
    power-of-two maps, odd sizes whose padded blocks must read as zeros, thin maps, with and without copied top levels."""
    rng = np.random.default_rng(21)
    for (h, w) in ((16, 32), (33, 20), (7, 50), (64, 128), (5, 5), (2, 2), (3, 9), (130, 67)):
        d = (rng.random((h, w)) ** 4 + 0.003).astype(np.float32)
        d[rng.random((h, w)) < 0.1] = 0.0                          # empty texels: zero-mass blocks next to the chosen ones
        pts = np.vstack([rng.random((60, 2)), [[0.0, 0.0], [1.0, 1.0], [0.0, 1.0], [0.999999, 0.5]]]).astype(np.float32)
        k = 0
        while True:
            a = np.zeros(3, np.float32)
            rc = oracle.L.orc_hier2d(fp(d), w, h, 200 + k, fp(np.ascontiguousarray(pts[0])), fp(a))
            if rc == -2:
                break
            assert rc == 0
            for xy in pts:
                a = np.zeros(3, np.float32); b = np.zeros(3, np.float32)
                assert oracle.L.orc_hier2d(fp(d), w, h, 200 + k, fp(np.ascontiguousarray(xy)), fp(a)) == 0
                assert oracle.L.orc_hier2d(fp(d), w, h, 100 + k, fp(np.ascontiguousarray(xy)), fp(b)) == 0
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (h, w, k, xy, a, b)
            k += 1
        assert k >= 2
