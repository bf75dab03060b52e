"""GPU parity: key generation and the radix sort, through the C ABI, against the CPU checker.

Bar: bit-exact (keys are compared as uint32 bit patterns; the permutation must equal the stable sort's).
Reference: Scenes.h:28-36, 314-319 (keys); radix_sort.hpp:258-392 (sort contract).
"""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gs4d):
    c = gs4d.Context(64, 64)
    yield c
    c.close()


def _sort_on_gpu(ctx, keys, vals):
    n = keys.size
    kb, vb = ctx.buffer(keys), ctx.buffer(vals)
    ctx.sort_pairs(kb, vb, n)
    k, v = ctx.read(kb, np.uint32, n), ctx.read(vb, np.uint32, n)
    ctx.delete(kb)
    ctx.delete(vb)
    return k, v


@pytest.mark.parametrize("n", [2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 100003, 1 << 20, (3 << 20) + 17, (5 << 20) + 3])
def test_sort_random_full_range(ctx, oracle, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    k, v = _sort_on_gpu(ctx, keys, vals)
    ek, ev = oracle.sort_pairs(keys, vals, "lsd")
    assert np.array_equal(k, ek)
    assert np.array_equal(v, ev)


@pytest.mark.parametrize("n,distinct", [(5, 2), (2049, 3), (70000, 16), (300000, 1), (1 << 20, 255)])
def test_sort_heavy_duplicates_is_stable(ctx, oracle, n, distinct):
    rng = np.random.default_rng(n + distinct)
    pool = rng.integers(0, 2 ** 32, distinct, dtype=np.uint64).astype(np.uint32)
    keys = pool[rng.integers(0, distinct, n)]
    vals = np.arange(n, dtype=np.uint32)
    k, v = _sort_on_gpu(ctx, keys, vals)
    ek, ev = oracle.sort_pairs(keys, vals, "std")      # std::stable_sort: independent of the LSD checker
    assert np.array_equal(k, ek)
    assert np.array_equal(v, ev)
    # ties keep input order
    same = k[1:] == k[:-1]
    assert np.all(v[1:][same] > v[:-1][same])


def test_sort_edge_cases(ctx, gs4d):
    # n <= 1 is a no-op (radix_sort.hpp:260)
    kb, vb = ctx.buffer(np.array([7], np.uint32)), ctx.buffer(np.array([9], np.uint32))
    ctx.sort_pairs(kb, vb, 1)
    ctx.sort_pairs(kb, vb, 0)
    assert ctx.read(kb, np.uint32, 1)[0] == 7 and ctx.read(vb, np.uint32, 1)[0] == 9
    # n larger than the buffers is refused, not a fault
    with pytest.raises(gs4d.Gs4dError):
        ctx.sort_pairs(kb, vb, 2)
    with pytest.raises(gs4d.Gs4dError):
        ctx.sort_pairs(kb, kb, 1 << 10)
    # already sorted / reverse sorted / extremes
    keys = np.array([0, 0, 1, 0xFFFFFFFF, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF], np.uint32)
    kb2, vb2 = ctx.buffer(keys), ctx.buffer(np.arange(7, dtype=np.uint32))
    ctx.sort_pairs(kb2, vb2, 7)
    assert ctx.read(kb2, np.uint32, 7).tolist() == sorted(keys.tolist())
    assert ctx.read(vb2, np.uint32, 7).tolist() == [0, 1, 2, 6, 5, 3, 4]
    # sorting a prefix leaves the tail alone
    keys = np.arange(1000, 0, -1).astype(np.uint32)
    kb3, vb3 = ctx.buffer(keys), ctx.buffer(np.arange(1000, dtype=np.uint32))
    ctx.sort_pairs(kb3, vb3, 600)
    out = ctx.read(kb3, np.uint32, 1000)
    assert np.array_equal(out[:600], np.sort(keys[:600])) and np.array_equal(out[600:], keys[600:])
    for b in (kb, vb, kb2, vb2, kb3, vb3):
        ctx.delete(b)
    ctx.delete(kb)          # double delete is tolerated (Scenes.h:220-224, 291-299)
    ctx.delete(0)


def test_keygen_matches_reference_fixture(ctx, oracle):
    """Keys for the first 1000 LinearMotion records equal the arrays the reference's key loop produced."""
    rec = oracle.golden("linear_first1000")
    db = ctx.buffer(rec)
    kb, ib = ctx.buffer(nbytes=4000), ctx.buffer(nbytes=4000)
    for k, t in enumerate([0.0, 12.5, 49.0]):
        ctx.keygen(db, t, (60.0, 90.0, 90.0), kb, ib, 1000)
        keys = ctx.read(kb, np.float32, 1000)
        gold = oracle.golden(f"linear_keys_t{k}_first4000")[:1000]
        assert np.array_equal(keys.view(np.uint32), gold.view(np.uint32))
        assert np.array_equal(ctx.read(ib, np.uint32, 1000), np.arange(1000, dtype=np.uint32))
    for b in (db, kb, ib):
        ctx.delete(b)


@pytest.mark.parametrize("n", [1, 1000, 1 << 20])
def test_keygen_and_permutation_bit_exact(ctx, gs4d, oracle, n):
    pos4, q, scale, life, fade, vel, rgba = scenes.cube_params_4d(n)
    rec = gs4d.build_records_4d(pos4, q, scale, life, fade, vel, rgba)
    cam = np.array(scenes.CAM_CUBE[0], np.float32)
    db = ctx.buffer(rec)
    kb, ib = ctx.buffer(nbytes=4 * n), ctx.buffer(nbytes=4 * n)
    for t in (0.0, 17.25):
        ctx.keygen(db, t, cam, kb, ib, n)
        keys = ctx.read(kb, np.float32, n)
        eidx, ekeys = oracle.keygen(rec, t, cam)
        assert np.array_equal(keys.view(np.uint32), ekeys.view(np.uint32))
        ctx.sort_pairs(kb, ib, n)          # scene's values buffer = sorter keys, scene's key buffer = payload (Scenes.h:327)
        perm = ctx.read(ib, np.uint32, n)
        _, eperm = oracle.sort_pairs(ekeys.view(np.uint32), eidx, "lsd")
        assert np.array_equal(perm, eperm)
        sk = ctx.read(kb, np.float32, n)
        assert np.all(sk[1:] >= sk[:-1])                               # sortedness (sort_test_check_sorted.comp.glsl intent)
        assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))   # multiset preserved
    # updating records through SubData is seen by the next keygen (SoA shadow refresh)
    if n >= 1000:
        rec2 = rec.copy()
        rec2[10:20, 0:3] += 100.0
        ctx.subdata(db, rec2[10:20], offset=10 * 96)
        ctx.keygen(db, 0.0, cam, kb, ib, n)
        _, ekeys2 = oracle.keygen(rec2, 0.0, cam)
        assert np.array_equal(ctx.read(kb, np.float32, n).view(np.uint32), ekeys2.view(np.uint32))
    for b in (db, kb, ib):
        ctx.delete(b)
