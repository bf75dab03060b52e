"""CPU: the oracle's restatement of the reference's HOST math against fixtures generated from the reference's own C++
(oracle/ref/refgen.cpp -> tests/golden/*.bin).  Bar: bit-exact (float32 bit patterns)."""
import zlib

import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_fixture_files_match_manifest(oracle):
    import json, os
    man = json.load(open(os.path.join(oracle.GOLDEN_DIR, "manifest.json")))
    for name, ent in man.items():
        if isinstance(ent, dict) and "dtype" in ent:
            raw = open(os.path.join(oracle.GOLDEN_DIR, name + ".bin"), "rb").read()
            assert len(raw) == 4 * ent["count"], name
            assert zlib.crc32(raw) == ent["crc32"], name
    assert man["teapot_vertices"] == 3644                      # Objects/teapot.vdata


def test_splat4d_ctor2(oracle):          # Splat.h:132-159
    for r, g in zip(oracle.golden("splat4d_ctor2_in"), oracle.golden("splat4d_ctor2_cov")):
        assert np.array_equal(bits(oracle.splat4d_cov(r[0:4], r[4:7], float(r[7]), float(r[8]), r[9:12])), bits(g))


def test_splat4d_ctor1_two_quaternions(oracle):   # Splat.h:91-130
    for r, g in zip(oracle.golden("splat4d_ctor1_in"), oracle.golden("splat4d_ctor1_cov")):
        assert np.array_equal(bits(oracle.splat4d_cov2q(r[0:4], r[4:8], r[8:12])), bits(g))


def test_splat3d_ctor_and_quat_look_at(oracle):   # Splat.h:334-344, Scenes.h:268
    for r, g in zip(oracle.golden("splat3d_ctor_in"), oracle.golden("splat3d_ctor_cov")):
        assert np.array_equal(bits(oracle.splat3d_cov(r[0:4], r[4:7])), bits(g))
    for n, q in zip(oracle.golden("quatlookat_in"), oracle.golden("quatlookat_q")):
        assert np.array_equal(bits(oracle.quat_look_at(n)), bits(q))


def test_camera(oracle):                 # Camera.cpp:50-58
    for r, g in zip(oracle.golden("camera_in"), oracle.golden("camera_viewproj")):
        assert np.array_equal(bits(oracle.look_at(r[2:5], r[5:8])), bits(g[:16]))
        assert np.array_equal(bits(oracle.perspective(60.0, int(r[0]), int(r[1]), 0.1, float(r[8]))), bits(g[16:]))
    # the projection of the README screenshot: [1.73205, 1.73205, -1.00004, -0.200004] at aspect 1
    p = oracle.perspective(60.0, 800, 800, 0.1, 5000.0)
    np.testing.assert_allclose([p[0], p[5], p[10], p[14]], [1.73205, 1.73205, -1.00004, -0.200004], rtol=2e-6)


@pytest.mark.parametrize("k,t", [(0, 0.0), (1, 12.5), (2, 49.0)])
def test_sort_keys_of_the_key_loop(oracle, k, t):   # Scenes.h:28-36, 314-319
    cam = np.array([60, 90, 90], np.float32)
    rec = oracle.golden("linear_first1000")
    _, key = oracle.keygen(rec, t, cam)
    assert np.array_equal(bits(key), bits(oracle.golden(f"linear_keys_t{k}_first4000")[:1000]))
    rec25 = oracle.golden("linear_block25_first200")            # mu_t = 25: the time term is live
    assert np.all(rec25[:, 3] == 25.0)
    _, key25 = oracle.keygen(rec25, t, cam)
    assert np.array_equal(bits(key25), bits(oracle.golden(f"linear_keys_t{k}_block25_first200")))


def test_reference_record_layout(oracle):
    """SplatData = {vec4 pos, vec4 col, mat4 sig} (Scenes.h:22-37); record 0 of LinearMotion as observed in SURVEY.md App. C."""
    rec = oracle.golden("linear_first1000")
    np.testing.assert_allclose(rec[0, :4], [6.84037, 12.17719, -1.137015, 0.0], rtol=1e-6)
    np.testing.assert_allclose(rec[0, 4:8], [0.4073796, 0.4639345, 0.2659135, 1.0], rtol=1e-6)
    assert rec[0, 23] == np.float32(0.7213475) and rec[0, 20] == rec[0, 23] and rec[0, 11] == rec[0, 23]   # s, s*d.x with d=(1,0,0)
    sig = rec[:, 8:].reshape(-1, 4, 4)
    assert np.array_equal(sig, sig.transpose(0, 2, 1))           # symmetric
