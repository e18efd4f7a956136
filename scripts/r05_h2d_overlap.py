"""Round 5: how the 5 MB host-to-device copy of a query batch overlaps the previous batch's kernels (VERDICT r04 #3c).
Variants on the SIFT1M-shape index: (a) copy + query on one stream; (b) two slots (index / view), each copying on its own stream;
(c) two slots, the copies on ONE dedicated copy stream (event -> slot stream), issued one batch ahead."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("product-quantization-tree_amd")
w = bench.WORKLOADS["sift1m"]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
s0 = torch.cuda.Stream(dev); torch.cuda.set_stream(s0)
idx, base, meta = bench.build_index(pkg, w, 0)
idx.build_heuristic(500)
view = idx.view()
for h in (idx, view):
    h.set_option("stage_timing", 0)
qn, k = w["qn"], 100
q = bench.sift_like(qn, w["D"], 0xC0DE03, dev)
qh = torch.empty(q.shape, dtype=q.dtype, pin_memory=True); qh.copy_(q)
s1, sc = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
qd = [torch.empty_like(q), torch.empty_like(q)]
out = [(torch.empty((qn, k), dtype=torch.int32, device=dev), torch.empty((qn, k), dtype=torch.float32, device=dev), torch.empty(qn, dtype=torch.int32, device=dev)) for _ in range(2)]
hs, ss = (idx, view), (s0, s1)

def run(step, n=40, warm=6):
    for i in range(warm): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def a(i):
    with torch.cuda.stream(s0): qd[0].copy_(qh, non_blocking=True)
    idx.query_dev(qd[0], 20000, 500, k, *out[0], stream=s0.cuda_stream)
def b(i):
    s = i & 1
    with torch.cuda.stream(ss[s]): qd[s].copy_(qh, non_blocking=True)
    hs[s].query_dev(qd[s], 20000, 500, k, *out[s], stream=ss[s].cuda_stream)
ev = [torch.cuda.Event(), torch.cuda.Event()]
done = [torch.cuda.Event(), torch.cuda.Event()]
def c(i):
    s = i & 1
    with torch.cuda.stream(sc):
        sc.wait_event(done[s])           # the slot's previous batch no longer reads qd[s]
        qd[s].copy_(qh, non_blocking=True)
        ev[s].record(sc)
    ss[s].wait_event(ev[s])
    hs[s].query_dev(qd[s], 20000, 500, k, *out[s], stream=ss[s].cuda_stream)
    done[s].record(ss[s])
def d(i):  # no copy at all, two slots (the form of `value`)
    s = i & 1
    hs[s].query_dev(q, 20000, 500, k, *out[s], stream=ss[s].cuda_stream)
def f(i):  # zero copy: the traversal reads the batch from the pinned host buffer itself (device-visible host memory), two slots
    s = i & 1
    hs[s].query_dev(qh, 20000, 500, k, *out[s], stream=ss[s].cuda_stream)
def g(i):  # zero copy, one slot
    idx.query_dev(qh, 20000, 500, k, *out[0], stream=s0.cuda_stream)
def e(i):  # copies only
    s = i & 1
    with torch.cuda.stream(ss[s]): qd[s].copy_(qh, non_blocking=True)
for name, f in (("one stream: copy + query", a), ("two slots, copy on the slot's stream", b), ("two slots, copies on a dedicated stream", c), ("two slots, queries resident", d), ("zero copy (kernel reads pinned host memory), two slots", f), ("zero copy, one slot", g), ("copies only, two streams", e)):
    ms = run(f)
    print("%-58s %.4f ms per batch  %.1f M q/s" % (name, ms, qn / ms / 1e3), flush=True)

ref = [t.clone() for t in out[0]]
idx.query_dev(q, 20000, 500, k, *out[0], stream=s0.cuda_stream); torch.cuda.synchronize()
a0 = [t.clone() for t in out[0]]
idx.query_dev(qh, 20000, 500, k, *out[0], stream=s0.cuda_stream); torch.cuda.synchronize()
print("zero-copy results identical:", all(torch.equal(x, y) for x, y in zip(a0, out[0])))
