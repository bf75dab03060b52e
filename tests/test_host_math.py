"""CPU: the PRODUCT's host-side parameterisation (libgs4d.so gs4d_host_*, the Splat.h / Camera.cpp mirror) against the
fixtures generated from the reference's own C++.  Bar: bit-exact.  No GPU needed: these are CPU functions of the library."""
import numpy as np
import pytest


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


@pytest.mark.parametrize("name,blocks", [("rotation", (45,)), ("combined", (33,)), ("broken", (19,)), ("square", (23, 91))])
def test_motion_scene_generators_reproduce_the_reference_ssbo(gs4d, oracle, name, blocks):
    """RotationMotion / CombinedMotion / BrokenMotion / SquareMotion::init (Scenes.h:775-803, 1035-1068, 1965-1989, 2216-2259) with the
    class defaults: bit-for-bit the SSBOs the reference's own Splat4D / GetColor / glm code builds (oracle/ref/refgen.cpp)."""
    import zlib
    tea = oracle.golden("teapot_vdata")
    rec = getattr(gs4d, "scene_" + name)(tea)
    full = oracle.golden(name + "_full")
    assert rec.shape == (full["records"], 24)
    assert np.array_equal(bits(rec[:300]), bits(oracle.golden(name + "_first300")))
    for b in blocks:
        assert np.array_equal(bits(rec[b * 3644:b * 3644 + 200]), bits(oracle.golden(f"{name}_block{b}_first200")))
    assert zlib.crc32(rec.tobytes()) == full["crc32"]
    assert np.array_equal(getattr(gs4d, "scene_" + name)(tea, max_records=4000), rec[:4000])


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


def test_sd_loader(gs4d, oracle, tmp_path):
    """VData::parse_splat_data + ObjectDisplay::init (VDataParser.h:60-123, Scenes.h:2483-2491): the synthetic .sd fixture was read by
    the reference's own parser (oracle/ref/refgen.cpp); the records must be the same bits.  Blank lines and runs of spaces are skipped."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "synthetic.sd")
    rec = gs4d.parse_sd(path, object_scale=2.5)
    assert np.array_equal(bits(rec), bits(oracle.golden("synthetic_sd_records")))
    assert np.array_equal(gs4d.parse_sd(path, object_scale=2.5, cap_records=5), rec[:5])
    with pytest.raises(FileNotFoundError):
        gs4d.parse_sd(str(tmp_path / "missing.sd"))
    # a trailing partial splat is dropped
    txt = open(path).read().split()
    p = tmp_path / "cut.sd"
    p.write_text(" ".join(txt[:23 * 3 + 7]))
    assert np.array_equal(gs4d.parse_sd(str(p), object_scale=2.5), rec[:3])


def test_splat3d_mesh_and_2d_records(gs4d, oracle):
    """Splat3D::GetSplatMesh (Splat.h:433-447), Splat2D::CalcAndSetSigma (:576-582) and the Gaussians2D record expression
    (Scenes.h:1490-1496) against the reference's own classes / glm arithmetic (oracle/ref/refgen.cpp items 14, 15)."""
    for r, g in zip(oracle.golden("splat3d_mesh_in"), oracle.golden("splat3d_mesh_verts")):
        v = gs4d.splat3d_mesh(r[0:3], r[4:8], r[8:11], r[11:15])
        assert np.array_equal(bits(v.reshape(-1)), bits(g))
    for r, g in zip(oracle.golden("splat2d_sigma_in"), oracle.golden("splat2d_sigma_inv")):
        assert np.array_equal(bits(gs4d.splat2d_sigma_inv(r[0:2], float(r[2]), float(r[3]))), bits(g))
    for r, g in zip(oracle.golden("gaussians2d_in"), oracle.golden("gaussians2d_records")):
        assert np.array_equal(bits(gs4d.gaussians2d_record(float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4]), r[5:8])), bits(g))


def test_png_writer(gs4d, tmp_path):
    """Presentation: the PNG the library writes decodes (zlib + CRC checks, done here by hand) to the frame it was
    given, top row first."""
    import struct
    import zlib
    rng = np.random.default_rng(3)
    h, w = 37, 53
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    p = tmp_path / "f.png"
    gs4d.write_png(str(p), img)
    raw = p.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)
        chunks.append((typ, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 6, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 4 * w)
    assert np.all(rows[:, 0] == 0)
    assert np.array_equal(rows[:, 1:].reshape(h, w, 4), img[::-1])
    with pytest.raises(OSError):
        gs4d.write_png(str(tmp_path / "no_such_dir" / "f.png"), img)


def test_camera_input_model_bit_for_bit(gs4d, oracle):
    """Camera::HandleInput / HandleCamRotation (Camera.cpp:116-207) as a state machine: 96 steps of keys + cursor positions give the very
    positions, orientations and up vectors GLM gave the fixture generator (glm::rotate / normalize / cross on the reference's Camera;
    the GLFW plumbing around them is restated there, see oracle/ref/refgen.cpp section 8)."""
    steps = oracle.golden("camera_walk_in")
    want = oracle.golden("camera_walk_out")
    st = gs4d.CameraState.make(800, 800, (60, 90, 90), (0, -1, -1))
    captured_before = False
    for (keys, mx, my), row in zip(steps, want):
        recenter, hide = gs4d.camera_input(st, int(keys), mx, my)
        got = np.array(list(st.position) + list(st.orientation) + list(st.up), np.float32)
        assert np.array_equal(got.view(np.uint32), row[:9].view(np.uint32)), (keys, got, row)
        assert bool(st.capture_mouse) == bool(row[9])
        assert hide == ((int(keys) & gs4d.CAMKEY["C"]) != 0 and not captured_before)
        assert recenter == bool(st.capture_mouse) or hide
        captured_before = bool(st.capture_mouse)
    # imgui_active: nothing moves (Camera.cpp:119)
    before = bytes(st)
    gs4d.camera_input(st, 0x7FF, 10.0, 10.0, imgui_active=True)
    assert bytes(st) == before
    # SetIsViewFixedOnPoint / GetViewport / GetFocal run from the reference's own Camera.cpp in the generator
    misc = oracle.golden("camera_misc").reshape(-1)
    gs4d.camera_look_at_point(st, (1.0, 2.0, 3.0))
    got = np.array(list(st.orientation) + list(st.up), np.float32)
    assert np.array_equal(got.view(np.uint32), misc[:6].view(np.uint32))
    assert np.array_equal(gs4d.camera_viewport(800, 800).view(np.uint32), misc[6:8].view(np.uint32))
    assert np.array_equal(gs4d.camera_focal(60.0, 800, 800).view(np.uint32), misc[8:10].view(np.uint32))
