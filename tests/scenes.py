"""Deterministic synthetic inputs for the BASELINE.json configs.

Counter-based PRNG: splitmix64(seed ^ (index * 8 + stream)) -> 24-bit uniforms, so any slice of any config can be
generated independently and identically on every machine.  Only *parameters* are drawn here; the 96-byte records are
built by the host-side builders under test (product: gs4d.build_records_*; checker: oracle_lib).
"""
import numpy as np

SEED = 0x4D495335
MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & MASK
    return z ^ (z >> np.uint64(31))


def uniform(n, stream, seed=SEED, start=0):
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + n, dtype=np.uint64)
        z = splitmix64(np.uint64(seed) ^ (idx * np.uint64(64) + np.uint64(stream)))
    return ((z >> np.uint64(40)).astype(np.float64) / 16777216.0)


def normal(n, stream, seed=SEED, start=0):
    u1 = uniform(n, stream, seed, start)
    u2 = uniform(n, stream + 1, seed, start)
    return np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -25))) * np.cos(2.0 * np.pi * u2)


def cube_params(n, seed=SEED, start=0):
    """C2/C3: mu ~ U[-200,200]^3, q = normalised N(0,1)^4, scale ~ U[0.5,2]^3, rgb ~ U[0,1]^3, alpha ~ U[0.2,1]."""
    pos = np.stack([uniform(n, s, seed, start) * 400.0 - 200.0 for s in (0, 1, 2)], 1).astype(np.float32)
    q = np.stack([normal(n, s, seed, start) for s in (3, 5, 7, 9)], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = q.astype(np.float32)                                    # w,x,y,z
    scale = np.stack([uniform(n, s, seed, start) * 1.5 + 0.5 for s in (11, 12, 13)], 1).astype(np.float32)
    rgba = np.stack([uniform(n, 14, seed, start), uniform(n, 15, seed, start), uniform(n, 16, seed, start),
                     uniform(n, 17, seed, start) * 0.8 + 0.2], 1).astype(np.float32)
    return pos, q, scale, rgba


def cube_params_4d(n, seed=SEED, start=0):
    """C4: C2 distribution + mu_t ~ U[0,50], lifetime ~ U[0.5,2], fade 0.5, velocity ~ U[-5,5]^3."""
    pos, q, scale, rgba = cube_params(n, seed, start)
    mu_t = (uniform(n, 18, seed, start) * 50.0).astype(np.float32)
    life = (uniform(n, 19, seed, start) * 1.5 + 0.5).astype(np.float32)
    fade = np.full(n, 0.5, np.float32)
    vel = np.stack([uniform(n, s, seed, start) * 10.0 - 5.0 for s in (20, 21, 22)], 1).astype(np.float32)
    pos4 = np.concatenate([pos, mu_t[:, None]], 1)
    return pos4, q, scale, life, fade, vel, rgba


# cameras: (position, orientation); FOV 60, near 0.1, far 5000 (Camera.h:71-73, Application.cpp:126)
CAM_TEAPOT = ((60.0, 90.0, 90.0), (0.0, -1.0, -1.0))                                   # Scenes.h:228-229
CAM_CUBE = ((551.58, 350.43, -184.33), (-0.774978, -0.570354, 0.272222))              # README screenshot_05 camera
CAM_NONLINEAR = ((0.0, 60.0, 60.0), (0.0, -1.0, -1.0))                                 # Scenes.h:493-494
FOV, ZNEAR, ZFAR = 60.0, 0.1, 5000.0
