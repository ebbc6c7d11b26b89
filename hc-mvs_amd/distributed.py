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
    if local_slabs.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on one GPU box (several ranks on device 0): gloo gathers host tensors; staged through pinned-size copies
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local_slabs.cpu().contiguous(), group=group)
        out.copy_(host)
        return out
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


def _device_sync(dev):
    """The context enqueues on a stream of its own (hipStreamNonBlocking, hcmvs_api.cpp hcmvs_create): whatever torch or RCCL wrote into
    buffers the context is about to read -- gathered maps, copied-back slabs -- must have landed first, and the other way round."""
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def densify_scene(ctx, views, srcs, neighbors, order, init, params, group=None, device=None, batch=32, capacity=None,
                  fuse_kw=None, n_external_iters=1, postfilter=False, pf_kw=None, interleave=False):
    """The multi-rank scene path (BASELINE.json configs[3]) over the C-ABI binding, with the reference's outer
    iterations (SceneDensify.cpp:3684) and the fork's post-filters after outer iterations 1 and 2 (SceneDensify.cpp:3939-3958):

        shard_order
        for it in 0 .. n_external_iters-1:
            per-rank hcmvs_estimate_batch_device of the rank's reference images          (NO collective)
            if postfilter and it in (1, 2):
                all-gather of the packed maps -> hcmvs_postfilter_sequence over ALL images, on every rank -> the rank
                takes its own images' filtered maps back out of the gathered buffer
        all-gather of the packed maps -> hcmvs_set_depthmap_device + hcmvs_fuse on every rank (identical clouds)

    The post-filter is REPLICATED, not sharded: filtering image k is a fusion over the whole scene whose result (the depths it
    invalidates, the gaps it fills) is what image k + 1's fusion sees, so the images form one sequential chain; every rank runs that
    chain on identical gathered maps with a deterministic kernel, hence every rank holds identical filtered maps and no scatter is
    needed -- the result is independent of the number of ranks.  One all-gather per filtered outer iteration and one before the
    fusion (SURVEY.md 8e budgets exactly that).

    That batch schedule is a DEPARTURE from the reference: there image k is filtered right after its own
    estimate (EVTEstimateDepthMap queues EVTOptimizeDepthMap FIRST, SceneDensify.cpp:3914-3925), so its fusion sees the images > k as
    the previous outer iteration left them and zeroes depths in them before they are estimated again.  interleave=True is that order,
    exactly (the single-thread event order; image ids ascending): in outer iterations 1 and 2 the images are estimated ONE AT A TIME,
    each followed by its post-filter on every rank; the owner of an image broadcasts its fresh maps first.  Nothing runs in parallel
    across ranks there -- it is the exact mode, not the fast one.

    ctx:       binding.Context of this rank's device (one process per GPU)
    views:     {image id: dict(gray, K, R, C[, bgr])} of EVERY image (all the same size).  A rank uploads the gray image only of
               its own reference images and their source views; of the others it registers the camera and the colour image (what
               fusion and the post-filters read), so a 512-image scene does not sit on every GPU eight times
    srcs:      {image id: [source view ids]}            (DepthMapsData::InitViews, SceneDensify.cpp:336-397)
    neighbors: {image id: [neighbour ids]}              (DepthData::neighbors, decreasing importance)
    order:     fusion order, best connected first       (SceneDensify.cpp:3302)
    init:      {image id: (depth0 (H,W), normal0 (H,W,3), d_min, d_max)} numpy, at least for this rank's images
    params:    binding.Params of the estimate; it_external / n_external_iters are set here
    Returns the fused cloud dict of binding.Context.fuse plus `maps`: {id: (depth, normal, conf) device tensors}."""
    import copy
    import numpy as np
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    ids = list(order)
    h, w = views[ids[0]]["gray"].shape
    hw = h * w
    for i in ids:
        assert views[i]["gray"].shape == (h, w), "densify_scene: all images must have one size (equal slabs for one all-gather)"
    mine = shard_order(ids, world)[rank]
    n_local = (len(ids) + world - 1) // world
    need_gray = set(mine)
    for img in mine:
        need_gray.update(srcs[img])
    for i, v in views.items():
        if i in need_gray or v.get("bgr") is None:        # without a colour image the gradient map comes from the gray image
            ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"], bgr=v.get("bgr"))
        else:
            ctx.upload_view(i, None, v["K"], v["R"], v["C"], bgr=v["bgr"])
    slabs = torch.zeros(n_local, FLOATS_PER_PIXEL * hw, dtype=torch.float32, device=dev)
    rng = torch.zeros(n_local, 2, dtype=torch.float32, device=dev)      # (d_min, d_max) travel with the maps
    for j, img in enumerate(mine):
        d0, n0, dmin, dmax = init[img]
        slabs[j, :hw] = torch.from_numpy(np.ascontiguousarray(d0, np.float32)).reshape(-1).to(dev)
        slabs[j, hw:4 * hw] = torch.from_numpy(np.ascontiguousarray(n0, np.float32)).reshape(-1).to(dev)
        rng[j, 0], rng[j, 1] = float(dmin), float(dmax)
    allr = allgather_maps(rng, group)
    gathered = torch.empty(world * n_local, FLOATS_PER_PIXEL * hw, dtype=torch.float32, device=dev) if world > 1 else None

    def item_of(img, base):
        r = allr[slab_index(ids.index(img), world, n_local)]
        return dict(ref_id=img, src_ids=list(srcs[img]), d_min=float(r[0]), d_max=float(r[1]), d_depth=base, d_normal=base + 4 * hw,
                    d_conf=base + 16 * hw, seed_offset=img)

    def my_items(store, row_of):
        by_class = {}
        for img in mine:        # the items of a batch share a lane-layout class (up to 8 views, or 9..16)
            by_class.setdefault(8 if len(srcs[img]) <= 8 else 16, []).append(item_of(img, store[row_of(img)].data_ptr()))
        return by_class

    def register(allm):
        for k, img in enumerate(ids):
            row = slab_index(k, world, n_local)
            base = allm[row].data_ptr()
            ctx.set_depthmap_device(img, base, base + 4 * hw, base + 16 * hw, float(allr[row, 0]), float(allr[row, 1]))
            ctx.set_neighbors(img, [n for n in neighbors[img] if n in views][:31])

    p = copy.copy(params)
    p.n_external_iters = int(n_external_iters)
    _device_sync(dev)
    for it in range(int(n_external_iters)):
        p.it_external = it
        filt = postfilter and it in (1, 2)
        if filt and interleave:
            # the reference's order: estimate(k) -> post-filter(k) -> estimate(k + 1), image ids ascending.  The gathered buffer is the
            # state of the whole scene on every rank; an image is estimated in place by its owner and broadcast before its filter runs
            allm = allgather_maps(slabs, group, out=gathered)
            _device_sync(dev)
            register(allm)
            for img in sorted(ids):
                k = ids.index(img)
                row = slab_index(k, world, n_local)
                owner = k % world
                if owner == rank:
                    ctx.estimate_batch_device([item_of(img, allm[row].data_ptr())], p)
                    ctx.synchronize()
                if world > 1:
                    if allm.is_cuda and dist.get_backend(group) == "gloo":      # one-GPU rehearsal: gloo moves host tensors
                        host = allm[row].cpu()
                        dist.broadcast(host, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
                        allm[row].copy_(host)
                    else:
                        dist.broadcast(allm[row], src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
                    _device_sync(dev)
                ctx.postfilter(img, ids, **(pf_kw or {}))
            ctx.synchronize()
            if world > 1:
                for j, img in enumerate(mine):
                    slabs[j].copy_(allm[slab_index(ids.index(img), world, n_local)])
                _device_sync(dev)
            continue
        for cls, items in sorted(my_items(slabs, lambda img: mine.index(img)).items()):     # NO collective on the estimation path
            for b0 in range(0, len(items), batch):
                ctx.estimate_batch_device(items[b0:b0 + batch], p)
        ctx.synchronize()
        if filt:
            allm = allgather_maps(slabs, group, out=gathered)  # every rank: every image's current maps
            _device_sync(dev)
            register(allm)
            ctx.postfilter_sequence(sorted(ids), ids, **(pf_kw or {}))   # image after image (batch schedule, D6)
            ctx.synchronize()
            if world > 1:                                      # my images' filtered maps: out of the gathered buffer, back into my slabs
                for j, img in enumerate(mine):
                    slabs[j].copy_(allm[slab_index(ids.index(img), world, n_local)])
                _device_sync(dev)                              # the next estimate reads the slabs on the context's stream
    allm = allgather_maps(slabs, group, out=gathered)          # the exchange before fusion: 20 B/px, rank-major equal slabs
    _device_sync(dev)
    register(allm)
    maps = {img: unpack_maps(allm[slab_index(k, world, n_local)], h, w) for k, img in enumerate(ids)}
    # the cloud's buffers: as given, or counted -- a counting fusion first (no buffers), then the real one with exactly what it needs,
    # instead of half a point per pixel of the scene on every rank (66 GB of device and of host memory each at BASELINE configs[3]); a
    # fusion repeated on the maps a fusion has left makes the same decisions, so the cloud is the single pass's
    if capacity is not None:
        cap = capacity
    elif hasattr(ctx, "fuse_count"):
        kw = {k: v for k, v in (fuse_kw or {}).items() if k not in ("with_normals", "with_colors")}
        counted = ctx.fuse_count(ids, **kw)
        cap = counted[0] + 1
    else:
        cap = hw * len(ids) // 2
    cloud = ctx.fuse(ids, cap, **(fuse_kw or {}))          # replicated on every rank: identical clouds
    if capacity is None and hasattr(ctx, "fuse_count"):
        cloud["n_depths"] = counted[1]                     # the depths the fusion visited before it invalidated any (SceneDensify.cpp:3461 logs that number)
    cloud["maps"] = maps
    cloud["_keep"] = (allm, allr, slabs)                   # the registered device maps must outlive the context's use of them
    return cloud
