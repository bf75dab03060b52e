"""CPU: the sort contract.  The reference's three GLSL compute kernels, restated on the CPU under the radix_sort.hpp driver
(gs4do_glsl_radix_sort), produce exactly stable-argsort(uint32 key); so do the oracle's LSD port and std::stable_sort.
The reference holds no golden vectors for its sorter; its orphaned sort_test_* kernels state the intent checked here
(sortedness, equality with a reference sort, multiset preservation)."""
import numpy as np
import pytest


@pytest.mark.parametrize("n", [2, 5, 255, 256, 257, 511, 513, 1000, 2049, 5000])
@pytest.mark.parametrize("distinct", [1, 3, 17, None])
def test_glsl_kernels_equal_stable_sort(oracle, n, distinct):
    rng = np.random.default_rng(n * 31 + (distinct or 0))
    if distinct:
        pool = rng.integers(0, 2 ** 32, distinct, dtype=np.uint64).astype(np.uint32)
        keys = pool[rng.integers(0, distinct, n)]
    else:
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    kg, vg = oracle.sort_pairs(keys, vals, "glsl")
    ks, vs = oracle.sort_pairs(keys, vals, "std")
    kl, vl = oracle.sort_pairs(keys, vals, "lsd")
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ks, keys[order]) and np.array_equal(vs, vals[order])
    assert np.array_equal(kg, ks) and np.array_equal(vg, vs)
    assert np.array_equal(kl, ks) and np.array_equal(vl, vs)


def test_float_keys_sort_far_to_near(oracle):
    """Positive floats order like their bit patterns; ascending 1/dist = far -> near (SURVEY.md §8a sort contract)."""
    rng = np.random.default_rng(1)
    dist = rng.uniform(0.5, 2000.0, 10000).astype(np.float32)
    key = (np.float32(1.0) / dist).astype(np.float32)
    _, perm = oracle.sort_pairs(key.view(np.uint32), np.arange(key.size, dtype=np.uint32), "lsd")
    assert np.all(np.diff(dist[perm]) <= 0)
    # dist == 0 -> key = +inf sorts last (nearest)
    key2 = np.concatenate([key[:10], [np.float32(np.inf)]]).astype(np.float32)
    _, perm2 = oracle.sort_pairs(key2.view(np.uint32), np.arange(11, dtype=np.uint32), "lsd")
    assert perm2[-1] == 10


def test_n_le_1_is_noop(oracle):
    k, v = oracle.sort_pairs(np.array([5], np.uint32), np.array([9], np.uint32), "glsl")
    assert k[0] == 5 and v[0] == 9
    k, v = oracle.sort_pairs(np.zeros(0, np.uint32), np.zeros(0, np.uint32), "lsd")
    assert k.size == 0
