"""Timing probe: the Zeng backbone forward + backward on the stacked batch (both directions, one stream) against the two
directions as two half-batches on two HIP streams (BatchNorm statistics are per direction anyway).  The two-stream form
shares the weight-gradient buffers and workspace between the streams, so its gradients are NOT valid - timing only."""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import configs
from bihome_amd.step import build_model
from bihome_amd.weights import load_synthetic

cfg = configs.get("zeng-bihome")
model = build_model(cfg)
load_synthetic(model[0], 0)
bb = model[0]
bb.train()
B = 64
x = torch.randn(2 * B, 2, 128, 128, device="cuda")
g = torch.randn(2 * B, 2, 128, 128, device="cuda")
params = [p for p in bb.parameters() if p.requires_grad]


def one():
    y = bb._forward(x, groups=2)
    y.backward(g)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        y1 = bb._forward(x[:B], groups=1)
        y1.backward(g[:B])
    with torch.cuda.stream(s2):
        y2 = bb._forward(x[B:], groups=1)
        y2.backward(g[B:])
    cur.wait_stream(s1); cur.wait_stream(s2)


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("one stream, stacked batch: %.2f ms" % bench(one))
print("two streams, half batches: %.2f ms" % bench(two))
print("one stream, stacked batch: %.2f ms" % bench(one))
