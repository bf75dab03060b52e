"""CPU: the PRODUCT's host-side parameterisation (libgs4d.so gs4d_host_*, the Splat.h / Camera.cpp mirror) against the
fixtures generated from the reference's own C++.  Bar: bit-exact.  No GPU needed: these are CPU functions of the library."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_host_ctor_math(gs4d, oracle):
    for r, g in zip(oracle.golden("splat4d_ctor2_in"), oracle.golden("splat4d_ctor2_cov")):
        assert np.array_equal(bits(gs4d.splat4d_cov(r[0:4], r[4:7], float(r[7]), float(r[8]), r[9:12])), bits(g))
    for r, g in zip(oracle.golden("splat4d_ctor1_in"), oracle.golden("splat4d_ctor1_cov")):
        assert np.array_equal(bits(gs4d.splat4d_cov2q(r[0:4], r[4:8], r[8:12])), bits(g))
    for r, g in zip(oracle.golden("splat3d_ctor_in"), oracle.golden("splat3d_ctor_cov")):
        assert np.array_equal(bits(gs4d.splat3d_cov(r[0:4], r[4:7])), bits(g))
    for n, q in zip(oracle.golden("quatlookat_in"), oracle.golden("quatlookat_q")):
        assert np.array_equal(bits(gs4d.quat_look_at(n)), bits(q))


def test_host_camera(gs4d, oracle):
    for r, g in zip(oracle.golden("camera_in"), oracle.golden("camera_viewproj")):
        assert np.array_equal(bits(gs4d.look_at(r[2:5], r[5:8])), bits(g[:16]))
        assert np.array_equal(bits(gs4d.perspective(60.0, int(r[0]), int(r[1]), 0.1, float(r[8]))), bits(g[16:]))


def test_record_builders(gs4d, oracle):
    """Batch builders lay records out as Scenes::SplatData (Scenes.h:22-37); the static-3D embedding follows Scenes.h:2487."""
    import scenes
    pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(500)
    rec = gs4d.build_records_4d(pos4, q, scale, life, fade, vel, rgba)
    for i in (0, 17, 499):
        assert np.array_equal(bits(rec[i, 8:]), bits(oracle.splat4d_cov(q[i], scale[i], float(life[i]), float(fade[i]), vel[i])))
        assert np.array_equal(rec[i, :4], pos4[i]) and np.array_equal(rec[i, 4:8], rgba[i])
    rec3 = gs4d.build_records_3d(pos4[:, :3], q, scale, rgba)
    sig = rec3[:, 8:].reshape(-1, 4, 4)
    for i in (0, 250):
        assert np.array_equal(bits(sig[i, :3, :3].reshape(-1)), bits(oracle.splat3d_cov(q[i], scale[i])))
    assert np.all(sig[:, 3, :3] == 0) and np.all(sig[:, :3, 3] == 0) and np.all(sig[:, 3, 3] == 1) and np.all(rec3[:, 3] == 0)
    # the reference's own LinearMotion records are reproduced from their parameters: orientation from the normal, scale (4,4,1),
    # lifetime 1, fade 0.5, velocity (1,0,0)  (Scenes.h:258-279)
    tea = oracle.golden("teapot_vdata")
    ref = oracle.golden("linear_first1000")
    for i in (0, 1, 500, 999):
        qi = gs4d.quat_look_at(tea[i, 3:6])
        cov = gs4d.splat4d_cov(qi, (4.0, 4.0, 1.0), 1.0, 0.5, (1.0, 0.0, 0.0))
        assert np.array_equal(bits(cov), bits(ref[i, 8:]))


def test_scene_generators_reproduce_the_reference_ssbo(gs4d, oracle):
    """LinearMotion::init / NonLinearMotion::init (Scenes.h:258-279, 517-545, GetColor :58-68): the full SSBOs the product's
    generators build have the CRC-32 of the SSBOs built by the reference's own code (oracle/ref/refgen.cpp)."""
    import zlib
    tea = oracle.golden("teapot_vdata")
    lin = gs4d.scene_linear(tea)
    assert lin.shape == (oracle.golden("linear_full")["records"], 24)
    assert zlib.crc32(lin.tobytes()) == oracle.golden("linear_full")["crc32"]
    assert np.array_equal(bits(lin[:1000]), bits(oracle.golden("linear_first1000")))
    nl = gs4d.scene_nonlinear(tea)
    assert nl.shape == (oracle.golden("nonlinear_full")["records"], 24)
    assert zlib.crc32(nl.tobytes()) == oracle.golden("nonlinear_full")["crc32"]
    assert np.array_equal(bits(nl[45 * 3644:45 * 3644 + 200]), bits(oracle.golden("nonlinear_block45_first200")))
    # truncation used by config 5 (10^7 records of a longer sweep) is a prefix
    assert np.array_equal(gs4d.scene_nonlinear(tea, max_records=5000), nl[:5000])


def test_vdata_loader(gs4d, oracle, tmp_path):
    """VData::parse (VDataParser.h:25-58): whitespace-separated floats, 6 per vertex; missing file is reported, not fatal."""
    tea = oracle.golden("teapot_vdata")
    p = tmp_path / "t.vdata"
    with open(p, "w") as f:
        for k, row in enumerate(tea[:300]):
            f.write(" ".join(repr(float(x)) for x in row) + ("\n" if k % 3 else "   \n\n"))
    got = gs4d.parse_vdata(str(p))
    assert np.array_equal(bits(got), bits(tea[:300]))
    assert gs4d.parse_vdata(str(p), cap_vertices=10).shape == (10, 6)
    import pytest
    with pytest.raises(FileNotFoundError):
        gs4d.parse_vdata(str(tmp_path / "missing.vdata"))
