"""Frame sharding for multi-GPU runs: independent frames of a camera/time sweep, one process per GPU.

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
