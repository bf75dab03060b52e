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


def default_rank0_pct(world, link_gbs=54.0, frame_mix_gbs=3000.0):
    """Rank 0 renders AND receives: world - 1 senders write their RGBA8 frames into its HBM at about `link_gbs` each while they render (8.3 MB per
    1080p frame, ~6 500 frames/s per sender), out of the ~3 TB/s its own frames' traffic mix achieves — its fair share of the frames
    is an equal share minus that fraction.  100 at world 1, 98 at 2, 87 at 8."""
    return max(10, int(round(100.0 * (1.0 - (world - 1) * link_gbs / frame_mix_gbs))))


def deal(nframes, world, rank0_pct=100):
    """owner[k] = rank that renders frame k.  rank0_pct = 100: round-robin, frame k -> rank k mod world.  Otherwise rank 0 renders
    round(nframes / world * rank0_pct / 100) frames, spread evenly over the sweep, and the other frames go round-robin to ranks 1..world-1."""
    if world <= 1:
        return [0] * nframes
    if rank0_pct == 100:
        return [k % world for k in range(nframes)]
    n0 = min(nframes, max(0, int(round(nframes / world * rank0_pct / 100.0))))
    owner, nxt = [], 0
    for k in range(nframes):
        if (k + 1) * n0 // nframes > k * n0 // nframes:
            owner.append(0)
        else:
            owner.append(1 + nxt % (world - 1))
            nxt += 1
    return owner


def frames_for_rank(nframes, rank, world, rank0_pct=100):
    return [k for k, r in enumerate(deal(nframes, world, rank0_pct)) if r == rank]


def most_frames(nframes, world, rank0_pct=100):
    """frames of the busiest rank: every rank counts this many presentations per sweep, so that all ranks make the same collective calls"""
    own = deal(nframes, world, rank0_pct)
    return max(own.count(r) for r in range(world))


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


def run_sweep(nmine, most, every, pipelined, render, pack, gather):
    """One rank's sweep: the call sequence bench.py (N > 1) and the gloo tests share.  The rank renders its `nmine` frames; presentation is
    software-pipelined as a swap chain is — frame j is queued first, then frame j - 1 (the previous image) is packed into slot (p % every) of
    batch buffer (p // every) & 1 — and the batches travel according to gather_schedule(most, every).  Every rank counts `most` presentations (a
    rank with fewer frames skips the packs it has no frame for), so all ranks make the same collective calls whatever the deal.
      render(j)                          queue this rank's j-th frame
      pack(j, frames_back, x, slot)      pack this rank's j-th frame (the image `frames_back` frames behind the current one) into batch x, slot
      gather(x, lo, hi, first)           slots [lo, hi) of batch x travel; slot lo holds presentation number `first` (this rank's first-th frame)
    Returns the list of gather calls made (for tests)."""
    schedule = {p: (b, lo, hi) for p, b, lo, hi in gather_schedule(most, every)}
    calls, presented = [], 0

    def present(j, frames_back):
        nonlocal presented
        x = (presented // every) & 1
        if j < nmine:
            pack(j, frames_back, x, presented % every)
        presented += 1
        if presented in schedule:
            _b, lo, hi = schedule[presented]
            gather(x, lo, hi, presented - (hi - lo))
            calls.append((presented, x, lo, hi))

    for j in range(most):
        rendered = j < nmine
        if rendered:
            render(j)
        if not pipelined:
            present(j, 0)
        elif j >= 1:
            present(j - 1, 1 if rendered else 0)          # no new frame was started: frame j - 1 is still the current image
    if pipelined and most >= 1:
        present(most - 1, 0)                              # the last frame of the sweep is presented inside the sweep
    return calls


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
