"""Sharding for multi-GPU runs, one process per GPU.  Primary mode: independent FRAMES of a camera/time sweep.
Secondary mode, for one huge frame (config 5): rows of 8x8-pixel TILES dealt round-robin (gs4d_set_tile_shard); every rank generates
keys and sorts all splats (the blend order is global), bins and composites only its own tile rows, and the bands are gathered.

The reference is single-GPU and has no counterpart; the unit that shards is the frame (Application.cpp:145-190 renders one
per loop iteration, each independent of the last).  No collective touches the data path: the only exchange is one gather of
finished frames to rank 0 per step (torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""


def frame_of(step, rank, world):
    """Global frame index rendered by `rank` at local step `step`: frames are dealt round-robin, frame k -> rank k mod world."""
    return step * world + rank


def frames_for_rank(nframes, rank, world):
    return list(range(rank, nframes, world))


def sweep_time(frame, nframes, t_max=50.0):
    """t_k = t_max * k / (nframes - 1)  (256-frame sweep over [0, 50])."""
    return t_max * frame / max(1, nframes - 1)


def gather_frames(dist, frame, gathered, dst=0):
    """One gather of this step's frames (same-shaped tensors) to rank `dst`; `gathered` is a list of world tensors on dst, else None."""
    dist.gather(frame, gathered if dist.get_rank() == dst else None, dst=dst)
    return gathered


def gather_schedule(most, every):
    """When a rank gathers during a sweep of `most` presentations into batch buffers of `every` slots, and which slots travel:
    a list of (presentations so far, batch number, first slot, one past the last slot).  A gather after every full batch — and, inside the
    LAST batch, after every max(1, every // 4) presentations: what is still on the wire when a rank's last frame has been rendered is then
    a quarter of a batch, not a whole one (a batch of 8 RGBA8 1080p frames is 66 MB: about a millisecond of one xGMI link, a fifth of an
    8-GPU sweep).  Every rank runs the same schedule (all ranks count `most` presentations), so the collective calls match."""
    every = max(1, int(every))
    piece = max(1, every // 4)
    last = (most - 1) // every if most > 0 else 0
    out, lo = [], 0
    for p in range(1, most + 1):
        b, hi = (p - 1) // every, (p - 1) % every + 1
        if p % every == 0 or p == most or (b == last and hi % piece == 0):
            out.append((p, b, lo, hi))
            lo = 0 if p % every == 0 else hi
    return out


# ---- single-frame sharding by tile rows -----------------------------------------------------------------------------------------
TILE = 8


def band_pixel_rows(rank, world, height):
    """Framebuffer rows (bottom-up, like the framebuffer) of the tile rows ty % world == rank, in band order."""
    rows = []
    tiles_y = (height + TILE - 1) // TILE
    for ty in range(rank, tiles_y, world):
        rows.extend(range(ty * TILE, min(height, (ty + 1) * TILE)))
    return rows


def band_rows_max(world, height):
    """Rows of the largest band (rank 0's): gathers need same-shaped tensors, shorter bands are padded at the end."""
    return len(band_pixel_rows(0, world, height))


def assemble_bands(bands, width, height, world):
    """bands[r]: array-like of at least len(band_pixel_rows(r)) rows x width (x channels) -> the full frame."""
    import numpy as np
    first = np.asarray(bands[0])
    out = np.zeros((height, width) + first.shape[2:], first.dtype)
    for r in range(world):
        rows = band_pixel_rows(r, world, height)
        if rows:
            out[rows] = np.asarray(bands[r])[:len(rows)]
    return out
