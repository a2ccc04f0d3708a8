"""Data-parallel PRODUCT path on the GPU box: two ranks share the one MI355X over the gloo backend
(RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's), each runs its shard through
step.attach_reducer + net.run_backward hooks; the all-reduced flat gradient must equal the sum of the two shards'
gradients computed by a single process (per-replica BatchNorm, batch-SUM loss), replicas must start from rank 0's
weights and stay identical after the optimizer step."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from bihome_amd import configs, synth
from bihome_amd.ddp import shard_range
from bihome_amd.weights import load_synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_two_rank_step_matches_single_process_shard_sum(tmp_path, overlap):
    from bihome_amd.step import build_model
    B = 8
    out = str(tmp_path / "r0.npz")
    env = dict(os.environ, BIHOME_DIST_BACKEND="gloo", BIHOME_OVERLAP=overlap, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_worker.py"), out, str(B)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got, got1 = dict(np.load(out)), dict(np.load(out + ".rank1.npz"))
    log = os.path.join(ROOT, "gpurun_out", "ddp_two_rank_overlap%s.log" % overlap)
    os.makedirs(os.path.dirname(log), exist_ok=True)
    # single process: each shard through the same model (weights of seed 0 = rank 0's), gradients summed
    cfg = configs.get("zeng-bihome")
    model = build_model(cfg)
    load_synthetic(model[0], 0)
    load_synthetic(model[1].auxiliary_resnet, 0)
    model.train()
    d = synth.make_pairs(B, seed=77)
    g = torch.Generator().manual_seed(5)
    ch = [torch.randint(1, 128 * 128, (B, 128), generator=g) for _ in range(2)]
    total, losses = None, []
    for rank in range(2):
        lo, hi = shard_range(B, rank, 2)
        load_synthetic(model[1].auxiliary_resnet, 0)        # (BatchNorm running statistics do not enter train-mode gradients)
        data = {k: torch.tensor(d[k][lo:hi]).cuda() for k in ("patch_1", "patch_2", "delta")}
        data["choice_12"], data["choice_21"] = ch[0][lo:hi].cuda(), ch[1][lo:hi].cuda()
        for p in model.parameters():
            p.grad = None
        loss, _, _ = model(data)
        loss.backward()
        torch.cuda.synchronize()
        flat = model[0]._runner.flat.flat.detach().cpu().numpy().copy()
        total = flat if total is None else total + flat
        losses.append(loss.item())
    num, den = float(np.sqrt(((got["flat"] - total) ** 2).sum())), float(np.sqrt((total ** 2).sum()))
    with open(log, "w") as f:
        f.write("two ranks on one MI355X (gloo), B=%d split 2x%d, BIHOME_OVERLAP=%s\n" % (B, B // 2, overlap))
        f.write("buckets %d, launched from backward hooks %d\n" % (int(got["n_buckets"]), int(got["n_hook"])))
        f.write("rank-0 loss %.6f (single-process shard 0: %.6f)\n" % (float(got["loss"]), losses[0]))
        f.write("|allreduced flat grad - single-process shard sum|_2 / |.|_2 = %.3e\n" % (num / den))
    assert abs(float(got["loss"]) - losses[0]) <= 1e-5 * abs(losses[0])
    assert num / den < 1e-4, num / den                      # fp32 atomics order only
    assert int(got["n_buckets"]) >= 4 and int(got["n_hook"]) >= int(got["n_buckets"]) - 1   # launched during backward
    # broadcast at attach: rank 1 (initialised with another seed and shifted running statistics) holds rank 0's state
    assert np.array_equal(got["w0"], got1["w0"]) and np.array_equal(got["rm0"], got1["rm0"])
    assert np.array_equal(got["w_after"], got1["w_after"])  # identical update on both replicas
    assert not np.array_equal(got["w_after"], got["w0"])
