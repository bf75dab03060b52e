"""GPU: the HIP vertex stage (csrc/preprocess.hip) against the reference's own CPU statement of that stage — Splat4D::Draw / Splat3D::Draw
(Splat.h:163-247, 355-431) as recorded in tests/golden/splat_draw_*.bin.  The projected records are read back through
gs4d_debug_read_projected and turned into uScreenPos / uScale / uVec1 / uColor.a (tests/splat_draw_cases.py: bars and caveats).
The alpha column goes through the device's expf: 5e-6 relative here against 2e-6 on the CPU."""
import numpy as np
import pytest

import splat_draw_cases as sd

pytestmark = pytest.mark.gpu


def projected(ctx, gs4d, cam, draw, n):
    ctx.set_clear_color(gs4d.CLEAR_COLOR)
    ctx.clear()
    draw()
    p = ctx.debug_projected(n)
    return sd.from_record(p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4], p[:, 5], p[:, 6], p[:, 14], cam)


@pytest.mark.parametrize("blk,fixture", sd.BLOCKS)
def test_splat4d_draw(gs4d, oracle, blk, fixture):
    rec = oracle.golden(fixture)
    n = rec.shape[0]
    for c, cam in enumerate(sd.cameras(oracle)):
        ctx = gs4d.Context(cam["W"], cam["H"])
        db = ctx.buffer(rec)
        for k, t in enumerate(oracle.golden(f"splat_draw_4d_b{blk}_times")):
            def draw():
                ctx.set_uniforms(time=float(t), min_opacity=0.0, view=cam["view"], proj=cam["proj"])
                ctx.set_mode(gs4d.MODE_4D_DIRECT)
                ctx.bind(1, db)
                ctx.draw_instanced(n)
            got = projected(ctx, gs4d, cam, draw, n)
            sd.check(oracle.golden(f"splat_draw_4d_b{blk}_cam{c}_t{k}"), got, cam, False, f"4D block {blk} camera {c} t={t}", alpha_rtol=5e-6)
        ctx.close()


def test_splat3d_draw(gs4d, oracle):
    din = oracle.golden("splat_draw_3d_in")
    verts = sd.verts72(din)
    n = din.shape[0]
    for c, cam in enumerate(sd.cameras(oracle)):
        ctx = gs4d.Context(cam["W"], cam["H"])
        vb = ctx.buffer(verts)

        def draw():
            ctx.set_uniforms(time=0.0, min_opacity=0.0, view=cam["view"], proj=cam["proj"])
            ctx.set_mode(gs4d.MODE_3D_FULL)
            ctx.draw_quads(vb, n)
        got = projected(ctx, gs4d, cam, draw, n)
        sd.check(oracle.golden(f"splat_draw_3d_cam{c}"), got, cam, True, f"3D camera {c}", alpha_rtol=5e-6)
        ctx.close()
