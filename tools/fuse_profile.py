"""diagnostic (not a test): the fusion leg of bench.py alone (nine estimated 1080p maps, raster and hashed order), for
`rocprofv3 --kernel-trace --stats -- python3 tools/fuse_profile.py`"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")
dev = torch.device("cuda:0")
views = synth.make_views(bench.W, bench.H, bench.FOCAL, bench.N_SRC, seed=2)
pts = synth.sparse_points(views, 2000, seed=5)
ctx = binding.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
print(json.dumps(bench.fuse_throughput(ctx, views, pts, dev)))
ctx.close()
