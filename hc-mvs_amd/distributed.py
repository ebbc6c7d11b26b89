"""Multi-GPU sharding of the densify path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

Depth maps are independent units given the read-only images and cameras of their source views, so reference
images are sharded over the ranks with NO collective on the estimation path; the only exchange is one all-gather
of the packed {depth, normal, conf} maps (20 B/px) before fusion, because fusing image A reads and writes every
neighbour's map (SceneDensify.cpp:3381-3449).  SURVEY.md section 8e.
"""
import torch
import torch.distributed as dist

FLOATS_PER_PIXEL = 5  # depth + normal xyz + conf


def shard_order(order, world):
    """image -> rank map: position k of the fusion order (best connected first, SceneDensify.cpp:3302) goes to
    rank k mod world, so ranks are load balanced.  Returns per-rank lists of image ids."""
    return [list(order[r::world]) for r in range(world)]


def slab_index(k, world, n_local):
    """row of order position k in the gathered [world * n_local, ...] tensor (rank-major)"""
    return (k % world) * n_local + k // world


def pack_maps(depth, normal, conf, out=None):
    """(H,W), (H,W,3), (H,W) -> flat [5*H*W] slab: depth | normal | conf"""
    hw = depth.numel()
    if out is None:
        out = torch.empty(FLOATS_PER_PIXEL * hw, dtype=torch.float32, device=depth.device)
    out[:hw] = depth.reshape(-1)
    out[hw:4 * hw] = normal.reshape(-1)
    out[4 * hw:] = conf.reshape(-1)
    return out


def unpack_maps(slab, h, w):
    hw = h * w
    return slab[:hw].view(h, w), slab[hw:4 * hw].view(h, w, 3), slab[4 * hw:].view(h, w)


def allgather_maps(local_slabs, group=None, out=None):
    """local_slabs: [n_local, 5*H*W] (the same n_local on every rank; pad with zero slabs if the images do not
    divide evenly).  Returns [world * n_local, 5*H*W], rank-major, identical on every rank.  One collective:
    with fixed equal slabs RCCL moves each rank's block directly to all peers over its xGMI links.
    out: optional preallocated result buffer (reused by callers that gather every step)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_slabs
    if out is None:
        out = torch.empty((world * local_slabs.shape[0],) + tuple(local_slabs.shape[1:]), dtype=local_slabs.dtype,
                          device=local_slabs.device)
    dist.all_gather_into_tensor(out, local_slabs.contiguous(), group=group)
    return out


def gather_scene_maps(order, my_maps, h, w, group=None, device=None):
    """my_maps: {image id: (depth, normal, conf) tensors} for the images shard_order() gave this rank.
    Returns {image id: (depth, normal, conf)} for every image of `order` on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_order(order, world)[rank]
    n_local = (len(order) + world - 1) // world
    dev = device if device is not None else (next(iter(my_maps.values()))[0].device if my_maps else torch.device("cpu"))
    slabs = torch.zeros(n_local, FLOATS_PER_PIXEL * h * w, dtype=torch.float32, device=dev)
    for j, img in enumerate(mine):
        pack_maps(*my_maps[img], out=slabs[j])
    allm = allgather_maps(slabs, group)
    return {img: unpack_maps(allm[slab_index(k, world, n_local)], h, w) for k, img in enumerate(order)}
