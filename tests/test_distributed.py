"""The N > 1 path on CPU: world_size-2 gloo processes shard the reference images, all-gather the packed maps and
fuse the gathered scene (with the oracle as the checker) -- every rank must end up with the same cloud as a
single process."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from fusion_scene import make_maps

D = importlib.import_module("hc-mvs_amd.distributed")


def test_shard_order_is_balanced_and_complete():
    order = [7, 3, 9, 1, 4, 0, 8]
    sh = D.shard_order(order, 3)
    assert sh == [[7, 1, 8], [3, 4], [9, 0]] and sorted(sum(sh, [])) == sorted(order)
    n_local = 3
    rows = [D.slab_index(k, 3, n_local) for k in range(len(order))]
    assert len(set(rows)) == len(order) and rows[:4] == [0, 3, 6, 1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        maps, order = make_maps(n_views=5, noise=0.002, outliers=0.03)
        h, w = maps[0]["depth"].shape
        mine = D.shard_order(order, world)[rank]
        # "estimate" only my reference images: the other ranks' maps are unknown to me before the exchange
        my = {i: (torch.from_numpy(maps[i]["depth"]), torch.from_numpy(maps[i]["normal"]), torch.from_numpy(maps[i]["conf"]))
              for i in mine}
        allm = D.gather_scene_maps(order, my, h, w)
        assert sorted(allm) == sorted(order)
        for i in order:
            assert np.array_equal(allm[i][0].numpy(), maps[i]["depth"]) and np.array_equal(allm[i][1].numpy(), maps[i]["normal"])
            assert np.array_equal(allm[i][2].numpy(), maps[i]["conf"])
        # fusion is replicated on every rank after the all-gather: identical clouds, identical to one process
        gathered = [dict(m, depth=allm[i][0].numpy(), normal=allm[i][1].numpy(), conf=allm[i][2].numpy()) for i, m in enumerate(maps)]
        cloud = O.fuse_depthmaps(gathered, order, 100000)
        ref = O.fuse_depthmaps(maps, order, 100000)
        assert cloud["n_points"] == ref["n_points"] and np.array_equal(cloud["xyz"], ref["xyz"])
        t = torch.tensor([cloud["n_points"]], dtype=torch.int64)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert lo.item() == hi.item() == cloud["n_points"]
        ret[rank] = cloud["n_points"]
    finally:
        dist.destroy_process_group()


def test_two_rank_allgather_and_replicated_fusion():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert len(ret) == 2 and ret[0] == ret[1] > 100


def _scene_worker(rank, world, port, ret, interleave):
    """densify_scene itself (sharding, all-gathers, the interleaved mode's broadcasts, copy-backs) on CPU tensors with the oracle
    standing in for the device context"""
    import scene_oracle as S
    binding = importlib.import_module("hc-mvs_amd.binding")
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views, srcs, neighbors, order, init = S.ring_scene(n=5, w=96, h=80, f=90.0, n_points=60)
        p = binding.Params()
        po = O.default_params()          # the oracle's defaults are the C-ABI's (DepthMap.cpp:69-143)
        for k, _ in p._fields_:
            setattr(p, k, getattr(po, k))
        p.adapthalfwin = 5; p.n_estimation_iters = 2; p.seed = 900; p.propagate_halfwin = 5; p.propagate_step = 4
        ctx = S.OracleContext()
        cloud = D.densify_scene(ctx, views, srcs, neighbors, order, init, p, device=torch.device("cpu"), n_external_iters=3, postfilter=True,
                                interleave=interleave)
        ret[rank] = (cloud["n_points"], cloud["xyz"].tobytes(), {i: cloud["maps"][i][0].numpy().tobytes() for i in order})
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_densify_scene_two_ranks_both_postfilter_schedules():
    """world-2 gloo run of densify_scene == one process == the scene-level oracle harness, for the batch schedule of the post-filters
     and for the reference's interleaved order (SceneDensify.cpp:3889-3965); the two schedules differ."""
    import scene_oracle as S
    clouds = {}
    for interleave in (False, True):
        mgr = mp.Manager()
        single = mgr.dict()
        mp.spawn(_scene_worker, args=(1, 0, single, interleave), nprocs=1, join=True)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ret = mgr.dict()
        mp.spawn(_scene_worker, args=(2, port, ret, interleave), nprocs=2, join=True)
        assert len(ret) == 2 and single[0][0] > 1000
        for r in (0, 1):
            assert ret[r][0] == single[0][0] and ret[r][1] == single[0][1]
            for i in single[0][2]:
                assert ret[r][2][i] == single[0][2][i]
        views, srcs, neighbors, order, init = S.ring_scene(n=5, w=96, h=80, f=90.0, n_points=60)
        want = S.densify(views, srcs, neighbors, order, init, n_external_iters=3, postfilter=True, interleave=interleave, seed=900, adapthalfwin=5,
                         n_estimation_iters=2, propagate_halfwin=5, propagate_step=4)
        assert want["cloud"]["n_points"] == single[0][0] and want["cloud"]["xyz"].tobytes() == single[0][1]
        clouds[interleave] = single[0][1]
    assert clouds[False] != clouds[True]
