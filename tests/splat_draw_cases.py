"""SURVEY.md §8(c) family (8): what the reference's own CPU vertex stage — Splat4D::Draw (Splat.h:163-247) and Splat3D::Draw
(Splat.h:355-431) — hands to its legacy per-splat shader, recorded by oracle/ref/refdraw_main.cpp into tests/golden/splat_draw_*.bin
(one row per Draw call: visible, uScreenPos.xy, uScale.xy, uVec1.xy, uVec2.xy, uSigma[4], uColor.rgba).

The helpers below turn a projected record — the CPU checker's or the HIP kernel's — back into those quantities and state the bars.
What is compared, and how tight:
  cull        Splat.h:230-236 / 411-415 == the shader's :108-115        exact (same records visible)
  uScreenPos  NDC centre ps.xy                                            |diff| <= 5e-7 (a handful of float roundings; measured 1.5e-7)
  uScale      {sqrt(lambda)} as a SET: the CPU helper orders large-first with a 1e-7 floor (Splat.h:43-52), the shader small-first with
              1e-6 (…Instanced.GLSL:59-66) -> the reference's values are floored at 1e-6 before the comparison; <= 1e-5 relative
  uVec1       direction of the LARGE eigenvector after the viewport division (Splat.h:224 4D, :404 3D), up to sign, to within what the
              two float32 formulas can resolve for that matrix (see check()): <= 1e-4 rad for at least 80 % of the splats of a set
  uColor.a    p(t) * colour.a (4D) / colour.a (3D)                        <= 2e-6 relative (measured: bit-equal on the CPU)
The sink of the recorded values is this build's own shadow Shader; everything between the splat's members and the sink is the
reference's and GLM's arithmetic.  An error of a factor 1/Sigma44 in the conditioning, a swapped eigenpair or a transposed Jacobian
moves these numbers by percents, not by 1e-5.
"""
import numpy as np

BLOCKS = ((0, "nonlinear_first500"), (45, "nonlinear_block45_first200"))


def cameras(oracle):
    """rows of splat_draw_cameras: W, H, position, orientation, view[16], proj[16]"""
    out = []
    for row in oracle.golden("splat_draw_cameras"):
        out.append({"W": int(row[0]), "H": int(row[1]), "pos": row[2:5], "ori": row[5:8], "view": row[8:24].copy(), "proj": row[24:40].copy()})
    return out


def verts72(din):
    """splat_draw_3d_in rows {pos3, colour4, cov9} -> the four 72-byte vertices per splat the 3D-Full path draws (Splat.h:433-447)"""
    n = din.shape[0]
    v = np.zeros((n, 4, 18), np.float32)
    corners = np.array([[0.5, 0.5], [0.5, -0.5], [-0.5, -0.5], [-0.5, 0.5]], np.float32)
    v[:, :, 0:2] = corners[None]
    v[:, :, 2:5] = din[:, None, 0:3]
    v[:, :, 5:9] = din[:, None, 3:7]
    v[:, :, 9:18] = din[:, None, 7:16]
    return v.reshape(n, 72)


def from_record(cx, cy, a0x, a0y, a1x, a1y, alpha, valid, cam):
    """NDC centre, the two scales and the unit eigenvectors back out of a projected record (f64): a_k = (e_k / s_k) / (sx, sy)"""
    W, H, P = cam["W"], cam["H"], cam["proj"].astype(np.float64)
    hw, hh = W / 2.0, H / 2.0
    sx, sy = P[0] * hw, P[5] * hh
    f = lambda a: np.asarray(a, np.float64)
    g0 = np.stack([f(a0x) * sx, f(a0y) * sy], 1)          # e0 / s0
    g1 = np.stack([f(a1x) * sx, f(a1y) * sy], 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        s0 = 1.0 / np.linalg.norm(g0, axis=1)
        s1 = 1.0 / np.linalg.norm(g1, axis=1)
        e0 = g0 * s0[:, None]
        e1 = g1 * s1[:, None]
    return {"ncx": (f(cx) - hw) / hw, "ncy": (f(cy) - hh) / hh, "s0": s0, "s1": s1, "e0": e0, "e1": e1, "alpha": f(alpha), "valid": np.asarray(valid) != 0}


def check(ref, got, cam, viewport_divides_unit_vector, what, alpha_rtol=2e-6):
    """ref: rows of a splat_draw_* fixture; got: from_record(...).  Returns the measured maxima (for the test's report)."""
    vis = ref[:, 0] > 0
    assert np.array_equal(vis, got["valid"]), f"{what}: cull differs for {np.flatnonzero(vis != got['valid'])[:8]}"
    if not vis.any():
        return {}
    r = ref[vis].astype(np.float64)
    d_ndc = max(np.abs(got["ncx"][vis] - r[:, 1]).max(), np.abs(got["ncy"][vis] - r[:, 2]).max())
    assert d_ndc <= 5e-7, f"{what}: uScreenPos off by {d_ndc}"
    lam_ref = np.sort(np.sqrt(np.maximum(r[:, 3:5] ** 2, 1e-6)), axis=1)
    lam_got = np.sort(np.stack([got["s0"][vis], got["s1"][vis]], 1), axis=1)
    d_scale = (np.abs(lam_got - lam_ref) / lam_ref).max()
    assert d_scale <= 1e-5, f"{what}: uScale off by {d_scale} relative"
    # the eigenvector of the large eigenvalue: the shader's second column (e1), the CPU helper's first (uVec1)
    big_is_1 = got["s1"][vis] >= got["s0"][vis]
    e_big = np.where(big_is_1[:, None], got["e1"][vis], got["e0"][vis])
    vp = np.array([cam["W"], cam["H"]], np.float64)
    vp /= np.linalg.norm(vp)                                # Camera.cpp:90-93
    d = e_big / vp
    d /= np.linalg.norm(d, axis=1, keepdims=True)           # the direction is what is compared (3D divides after normalising: same direction)
    v1 = r[:, 5:7] / np.linalg.norm(r[:, 5:7], axis=1, keepdims=True)
    sin = np.abs(d[:, 0] * v1[:, 1] - d[:, 1] * v1[:, 0])
    # How well each side can know that direction.  Both build the eigenvector as (offdiag, lambda - m00) — the reference with the LARGE
    # eigenvalue (Splat.h:66), the shader with the SMALL one and a quarter turn (…Instanced.GLSL:73-75) — from float32 values that carry
    # a rounding error of a few ulps of lambda_large: the angle is uncertain by that error over the length of the vector.  When the
    # off-diagonal element is tiny, the side whose eigenvalue is the diagonal element it subtracts is left with rounding noise
    # (splat 314 of block 0: (4.4e-8, 2e-10 of noise) — 0.04 rad).  The covariance is rebuilt from the decomposition under test (f64).
    l0, l1 = lam_got[:, 0] ** 2, lam_got[:, 1] ** 2
    e_small = np.where(big_is_1[:, None], got["e0"][vis], got["e1"][vis])
    m00 = l0 * e_small[:, 0] ** 2 + l1 * e_big[:, 0] ** 2
    off = l0 * e_small[:, 0] * e_small[:, 1] + l1 * e_big[:, 0] * e_big[:, 1]
    noise = 16.0 * 2.0 ** -24 * l1 * l1 / np.maximum(l1 - l0, 1e-300)     # lambda = m -+ sqrt(m^2 - p): the root amplifies the rounding of m^2 - p by m / (2 d)
    bound = noise / np.maximum(np.hypot(off, l1 - m00), 1e-300) + noise / np.maximum(np.hypot(off, l0 - m00), 1e-300) + 2e-6
    assert (sin <= bound).all(), f"{what}: uVec1 direction off by {sin[np.argmax(sin - bound)]} rad (bound {bound[np.argmax(sin - bound)]})"
    sharp = bound <= 1e-4
    assert sharp.mean() >= 0.8, f"{what}: only {sharp.mean():.2f} of the directions are well conditioned"
    worst = sin[sharp].max()
    a_ref, a_got = r[:, 16], got["alpha"][vis]
    d_alpha = (np.abs(a_got - a_ref) / np.maximum(np.abs(a_ref), 1e-30))[a_ref > 1e-30].max() if (a_ref > 1e-30).any() else 0.0
    assert d_alpha <= alpha_rtol, f"{what}: uColor.a off by {d_alpha} relative"
    assert np.abs(a_got[a_ref <= 1e-30]).max(initial=0.0) <= 1e-30
    return {"visible": int(vis.sum()), "ndc": d_ndc, "scale": d_scale, "dir": worst, "alpha": d_alpha}
