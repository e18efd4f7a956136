"""Times pqt_merge_topk (the per-batch merge after the all-gather) for 2/4/8 shards of [3][QN][k] messages on one GPU."""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from common import fixture
pkg = importlib.import_module("product-quantization-tree_amd")
f = fixture("odd")
idx = f.hip_index()
qn, k = 10000, 100
dev = torch.device("cuda", 0)
for world in (2, 4, 8):
    g = torch.Generator(device=dev); g.manual_seed(1)
    d = torch.rand((world, qn, k), generator=g, device=dev).sort(dim=2).values
    pack = torch.empty((world, 3, qn, k), dtype=torch.int32, device=dev)
    pack[:, 0] = torch.randint(0, 1 << 30, (world, qn, k), generator=g, device=dev, dtype=torch.int32)
    pack[:, 1] = d.view(torch.int32)
    pack[:, 2] = torch.randint(0, 1 << 20, (world, qn, k), generator=g, device=dev, dtype=torch.int32)
    oi = torch.empty((qn, k), dtype=torch.int32, device=dev); od = torch.empty((qn, k), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        idx.merge_topk_dev(world, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oi, od, stream=st, shard_stride=3 * qn * k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        idx.merge_topk_dev(world, qn, k, pack[0, 0], pack[0, 1].view(torch.float32), pack[0, 2], oi, od, stream=st, shard_stride=3 * qn * k)
    torch.cuda.synchronize()
    print("merge of %d shards x [%d][%d]: %.4f ms" % (world, qn, k, (time.perf_counter() - t) / 20 * 1e3))
