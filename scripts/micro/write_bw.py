# achieved HBM store bandwidth of plain fills / copies on this box (context for the table kernel's 80 MB of stores)
import torch, json
dev = torch.device("cuda:0")
out = {}
for mb in (80, 320, 1024, 4096):
    x = torch.empty(mb * (1 << 20) // 4, dtype=torch.float32, device=dev)
    y = torch.ones_like(x)
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: x.copy_(y)), ("read_sum", lambda: y.sum())):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out["%s_%dMB" % (name, mb)] = {"ms": round(ms, 4), "GBps_of_buffer": round(mb / 1024 / (ms / 1e3), 1)}
print(json.dumps(out, indent=1))
