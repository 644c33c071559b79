"""Offline replay of an image/depth/pose sequence, sharded one subsequence per GPU.

The per-frame path has a strict temporal dependency (frame t+1 fuses into the map frame t produced,
surfel_map.cpp:161), so a single sequence does not shard; independent subsequences do (SURVEY.md
§8(e)).  Each rank owns one GPU and one or more handles, replays its contiguous subsequence with
keyframe indices restarting at 0, and the final clouds are merged with one all-gather of the counts
and one all-gather of the padded clouds (RCCL over xGMI when the backend is "nccl"; the same code
runs on "gloo" with CPU tensors in the tests).  There is no collective on the per-frame path.
"""
from __future__ import annotations

SURFEL_BYTES = 44


def shard_subsequences(n_frames: int, world_size: int):
    """Contiguous split of [0, n_frames) into world_size subsequences whose lengths differ by at
    most one (4541 KITTI frames over 8 ranks -> 5 x 568 + 3 x 567)."""
    if world_size <= 0 or n_frames < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_frames, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def merge_clouds(local_cloud, group=None):
    """All-gather the per-rank surfel clouds.

    local_cloud: torch.uint8 tensor [n_r * 44] (the rank's SurfelElement array as bytes), on the
    device the process group communicates on.  Returns (merged uint8 tensor [sum n_r * 44] in rank
    order, list of per-rank counts)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = local_cloud.numel() // SURFEL_BYTES
    assert local_cloud.dtype == torch.uint8 and local_cloud.numel() == n_local * SURFEL_BYTES
    counts = torch.zeros(world, dtype=torch.int64, device=local_cloud.device)
    mine = torch.tensor([n_local], dtype=torch.int64, device=local_cloud.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_l = [int(c) for c in counts.tolist()]
    n_max = max(counts_l) if counts_l else 0
    if n_max == 0:
        return local_cloud.new_zeros(0), counts_l
    padded = local_cloud.new_zeros(n_max * SURFEL_BYTES)
    padded[: local_cloud.numel()] = local_cloud
    gathered = local_cloud.new_zeros(world * n_max * SURFEL_BYTES)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * n_max * SURFEL_BYTES: r * n_max * SURFEL_BYTES + counts_l[r] * SURFEL_BYTES]
             for r in range(world)]
    return torch.cat(parts), counts_l
