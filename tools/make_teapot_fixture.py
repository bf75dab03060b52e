#!/usr/bin/env python3
"""Derives tests/golden/utah_teapot_rgb100.npy from the reference's Screenshots/UtahTeapot.png (800x800 RGBA8, the only picture of
the teapot the reference holds): rows flipped to the framebuffer's bottom-up order, 8x8 box average -> 100 x 100 x 3 uint8.
Runs only where /root/reference exists (needs PIL); the derived grid (30 kB of data, no source text) is what is committed."""
import os
import sys

import numpy as np
from PIL import Image

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
im = np.asarray(Image.open(os.path.join(ref, "Screenshots", "UtahTeapot.png")).convert("RGB")).astype(np.float64)[::-1]
small = im.reshape(100, 8, 100, 8, 3).mean(axis=(1, 3))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "utah_teapot_rgb100.npy")
np.save(out, np.rint(small).astype(np.uint8))
print(out, small.shape)
