"""world_size-2 CPU (gloo) coverage of the data-parallel exchange: bucket construction over the flat
gradient buffer, hook-driven launch order, SUM semantics, and equivalence with a single process that
sees both shards (per-replica BatchNorm, loss summed over the batch - SURVEY.md 8(e))."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bihome_amd.ddp import FlatGradReducer, shard_range
from bihome_amd.net import FlatGrads


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 2, 1))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _tiny_net(0)
    fg = FlatGrads(list(net.parameters()))
    fg.attach(torch.device("cpu"))
    red = FlatGradReducer(fg, bucket_bytes=1024)          # tiny buckets -> several of them
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 2, 6, 6, generator=g)
    lo, hi = shard_range(8, rank, world)
    loss = net(x[lo:hi]).pow(2).sum()                     # batch-SUM loss like biHomE
    grads = torch.autograd.grad(loss, list(net.parameters()))
    # emulate the backward walk: last layer first, each gradient written into its flat view
    for p, gr in reversed(list(zip(net.parameters(), grads))):
        p.grad.copy_(gr)
        red.param_ready(p)
    red.allreduce()
    if rank == 0:
        torch.save({"flat": fg.flat.clone(), "launched": len(red.buckets)}, out)
    dist.destroy_process_group()


def test_bucketed_allreduce_sum_matches_single_process(tmp_path):
    port, out = _free_port(), str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # single process: the two shards go through the network separately (per-replica BN), gradients add
    net = _tiny_net(0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 2, 6, 6, generator=g)
    total = None
    for r in range(2):
        lo, hi = shard_range(8, r, 2)
        grads = torch.autograd.grad(net(x[lo:hi]).pow(2).sum(), list(net.parameters()))
        total = grads if total is None else [a + b for a, b in zip(total, grads)]
    fg = FlatGrads(list(net.parameters()))
    fg.attach(torch.device("cpu"))
    for p, gr in zip(net.parameters(), total):
        p.grad.copy_(gr)
    np.testing.assert_allclose(got["flat"].numpy(), fg.flat.numpy(), rtol=1e-5, atol=1e-6)


def test_shard_range_covers_batch():
    for gb, ws in ((512, 8), (256, 8), (10, 4), (3, 8)):
        spans = [shard_range(gb, r, ws) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_flat_grads_views_follow_channels_last_params():
    conv = torch.nn.Conv2d(3, 4, 3)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    fg = FlatGrads(list(conv.parameters()))
    fg.attach(torch.device("cpu"))
    assert conv.weight.grad.stride() == conv.weight.stride()
    conv.weight.grad[1, 2, 0, 1] = 5.0
    # kernel layout [O][kh][kw][I]: flat index of (o=1, kh=0, kw=1, i=2)
    assert fg.flat[1 * 27 + 0 * 9 + 1 * 3 + 2].item() == 5.0
    # zero_grad(set_to_none) detaches; attach() re-attaches and zeroes
    conv.zero_grad(set_to_none=True)
    fg.attach(torch.device("cpu"))
    assert conv.weight.grad is not None and fg.flat.abs().sum().item() == 0.0
    # buckets cover the buffer exactly once, last parameters first
    red = FlatGradReducer(fg, bucket_bytes=64)
    spans = sorted(red.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == fg.numel and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert red.buckets[0][1] == fg.numel


def _attach_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bihome_amd import configs
    from bihome_amd.step import attach_reducer, build_model
    from bihome_amd.weights import load_synthetic
    model = build_model(configs.get("zeng-bihome"), "cpu")
    load_synthetic(model[0], 10 + rank)                         # replicas start different
    with torch.no_grad():
        model[0].layer1[1].running_var.mul_(1.0 + rank)
    red = attach_reducer(model)                                  # product entry point: broadcast + FlatGradReducer
    fg = model[0]._runner.flat
    fg.attach(torch.device("cpu"))
    fg.flat.fill_(float(rank + 1))
    for p in reversed(fg.params):                               # the order net.run_backward finalises gradients in
        red.param_ready(p)
    launched = sum(red.launched)
    red.allreduce()
    torch.save({"w": model[0].layer3[0].upper_branch[0].weight.detach().clone(), "rv": model[0].layer1[1].running_var.clone(),
                "flat_min": fg.flat.min().item(), "flat_max": fg.flat.max().item(), "launched": launched,
                "buckets": len(red.buckets), "numel": fg.numel, "cl": model[0].layer3[0].upper_branch[0].weight.permute(0, 2, 3, 1).is_contiguous()},
               out + str(rank))
    dist.destroy_process_group()


def test_attach_reducer_broadcasts_and_buckets_the_product_model(tmp_path):
    """step.attach_reducer on the real Zeng backbone (CPU parameters, gloo): every replica ends up with rank 0's parameters
    and buffers (kernel-layout conv weights included), the 42.3 MB flat gradient splits into >= 5 buckets that all leave
    from the param_ready hooks, and the exchange is a SUM."""
    port, out = _free_port(), str(tmp_path / "a")
    mp.spawn(_attach_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    assert torch.equal(r0["w"], r1["w"]) and torch.equal(r0["rv"], r1["rv"]) and r0["cl"] and r1["cl"]
    assert r0["numel"] >= 10574178 and r0["buckets"] >= 5 and r0["launched"] == r0["buckets"]
    assert r0["flat_min"] == r0["flat_max"] == 3.0              # 1 + 2: SUM, every element exchanged exactly once


def test_bench_gpus_n_never_runs_fewer_ranks_silently():
    """`python bench.py --gpus 2` on a node without two GPUs exits non-zero before any training (it would self-spawn two
    ranks over RCCL); with a launcher that set WORLD_SIZE to something else it refuses as well (round-2 VERDICT missing #2)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BIHOME_DIST_BACKEND")}
    if torch.cuda.device_count() < 2:           # (on a multi-GPU node `--gpus 2` is a legitimate two-rank RCCL run, not a refusal)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode != 0 and "n_gpus" not in r.stdout and "refusing" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE=2" in r.stderr
