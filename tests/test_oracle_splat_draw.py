"""CPU: the checker's vertex stage (oracle/gs4d_oracle.cpp: gs4do_preprocess_4d / _3d, restating Splat4DVertexShaderInstanced.GLSL:81-150
and Splat3DVertexShaderFull.GLSL:43-98) against the reference's own CPU statement of the same stage, Splat4D::Draw / Splat3D::Draw
(Splat.h:163-247, 355-431), as recorded in tests/golden/splat_draw_*.bin.  Bars and caveats: tests/splat_draw_cases.py."""
import json
import os
import zlib

import numpy as np
import pytest

import splat_draw_cases as sd


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_fixture_files_match_manifest(oracle):
    man = json.load(open(os.path.join(oracle.GOLDEN_DIR, "manifest_draw.json")))
    n = 0
    for name, ent in man.items():
        if isinstance(ent, dict):
            raw = open(os.path.join(oracle.GOLDEN_DIR, name + ".bin"), "rb").read()
            assert len(raw) == 4 * ent["count"] and zlib.crc32(raw) == ent["crc32"], name
            n += 1
    assert n == 16


def test_cameras_are_the_checkers(oracle):
    """the matrices Draw() used (shadow Camera -> gs4d_host_*) are bit for bit the checker's (and the reference's: test_oracle_golden.py::test_camera)"""
    for cam in sd.cameras(oracle):
        assert np.array_equal(bits(oracle.look_at(cam["pos"], cam["ori"])), bits(cam["view"]))
        assert np.array_equal(bits(oracle.perspective(60.0, cam["W"], cam["H"], 0.1, 5000.0)), bits(cam["proj"]))


@pytest.mark.parametrize("blk,fixture", sd.BLOCKS)
def test_splat4d_draw(oracle, blk, fixture):
    rec = oracle.golden(fixture)
    seen_culled = seen_faded = False
    for c, cam in enumerate(sd.cameras(oracle)):
        for k, t in enumerate(oracle.golden(f"splat_draw_4d_b{blk}_times")):
            ref = oracle.golden(f"splat_draw_4d_b{blk}_cam{c}_t{k}")
            pr = oracle.preprocess(oracle.MODE_4D, rec, cam["view"], cam["proj"], cam["W"], cam["H"], t=float(t))
            got = sd.from_record(pr["cx"], pr["cy"], pr["a0x"], pr["a0y"], pr["a1x"], pr["a1y"], pr["alpha"], pr["valid"], cam)
            # the checker keeps R and S themselves: they must agree with what the record encodes
            v = got["valid"]
            np.testing.assert_allclose(got["s0"][v], pr["s0"][v], rtol=2e-6)
            np.testing.assert_allclose(got["s1"][v], pr["s1"][v], rtol=2e-6)
            m = sd.check(ref, got, cam, False, f"4D block {blk} camera {c} t={t}")
            seen_culled |= bool((ref[:, 0] == 0).any())
            seen_faded |= bool(((ref[:, 16] > 1e-4) & (ref[:, 16] < 0.9)).any())
    assert seen_faded and (seen_culled or blk == 45)       # the vectors exercise the cull and a live time opacity


def test_splat3d_draw(oracle):
    din = oracle.golden("splat_draw_3d_in")
    verts = sd.verts72(din)
    culled = 0
    for c, cam in enumerate(sd.cameras(oracle)):
        ref = oracle.golden(f"splat_draw_3d_cam{c}")
        pr = oracle.preprocess(oracle.MODE_3D, verts, cam["view"], cam["proj"], cam["W"], cam["H"])
        got = sd.from_record(pr["cx"], pr["cy"], pr["a0x"], pr["a0y"], pr["a1x"], pr["a1y"], pr["alpha"], pr["valid"], cam)
        sd.check(ref, got, cam, True, f"3D camera {c}")
        vis = ref[:, 0] > 0
        assert np.array_equal(ref[vis, 13:16], din[vis, 3:6])       # uColor.rgb is the splat's colour, untouched
        culled += int((~vis).sum())
    assert culled > 100


def test_the_comparison_has_teeth(oracle):
    """an error the screenshots cannot see: the conditioning's 1/Sigma44 applied twice moves uScreenPos by more than the bar"""
    rec = oracle.golden("nonlinear_first500").copy()
    cam = sd.cameras(oracle)[0]
    t = float(oracle.golden("splat_draw_4d_b0_times")[1])
    ref = oracle.golden("splat_draw_4d_b0_cam0_t1")
    bad = rec.copy()
    bad[:, 8 + 3] /= rec[:, 8 + 15]          # iSig[0][3] / Sigma44 once more
    pr = oracle.preprocess(oracle.MODE_4D, bad, cam["view"], cam["proj"], cam["W"], cam["H"], t=t)
    got = sd.from_record(pr["cx"], pr["cy"], pr["a0x"], pr["a0y"], pr["a1x"], pr["a1y"], pr["alpha"], pr["valid"], cam)
    with pytest.raises(AssertionError):
        sd.check(ref, got, cam, False, "perturbed")
