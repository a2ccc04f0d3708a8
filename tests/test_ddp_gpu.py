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


@pytest.mark.parametrize("world,B,overlap", [(2, 8, "0"), (2, 8, "1"), (8, 64, "0")])
def test_n_rank_step_matches_single_process_shard_sum(tmp_path, world, B, overlap):
    """world = 8, B = 64: BASELINE.json configs[2]'s split (512 pairs = 8 x 64) scaled down to 8 pairs per rank - eight ranks share the one
    MI355X of the box over gloo (round-3 VERDICT item 6a: the 8-GPU RCCL run itself is the driver's; this is the functional check of
    everything but the transport: eight shards, ten buckets from the hooks, SUM, broadcast, identical replicas)."""
    from bihome_amd.step import build_model
    out = str(tmp_path / "r0.npz")
    env = dict(os.environ, BIHOME_DIST_BACKEND="gloo", BIHOME_OVERLAP=overlap, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_worker.py"), out, str(B)]
    # (the ranks normally finish in 10-20 s; twice in round 5 a run of the whole suite sat in this launch until the timeout on boxes where the
    #  same build passed before and after - a stuck multi-process start on the shared GPU is retried once on a fresh port, and the ranks'
    #  output is shown if that does not help)
    r = None
    for attempt in range(2):
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=int(os.environ.get("BIHOME_TEST_DDP_TIMEOUT", "300")))
            break
        except subprocess.TimeoutExpired as e:
            if attempt == 1:
                pytest.fail("the ranks did not finish: stdout ...%s\nstderr ...%s" % ((e.stdout or b"")[-1500:], (e.stderr or b"")[-3000:]))
    assert r.returncode == 0, r.stderr[-3000:]
    got, others = dict(np.load(out)), [dict(np.load(out + ".rank%d.npz" % k)) for k in range(1, world)]
    log = os.path.join(ROOT, "gpurun_out", "ddp_%d_rank_overlap%s.log" % (world, overlap))
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
    for rank in range(world):
        lo, hi = shard_range(B, rank, world)
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
        f.write("%d ranks on one MI355X (gloo), B=%d split %dx%d, BIHOME_OVERLAP=%s\n" % (world, B, world, B // world, overlap))
        f.write("buckets %d, launched from backward hooks %d\n" % (int(got["n_buckets"]), int(got["n_hook"])))
        f.write("rank-0 loss %.6f (single-process shard 0: %.6f)\n" % (float(got["loss"]), losses[0]))
        f.write("|allreduced flat grad - single-process shard sum|_2 / |.|_2 = %.3e\n" % (num / den))
        f.write("per-rank losses %s (single-process shards: %s)\n" % ([round(float(got["loss"]), 5)] + [round(float(o["loss"]), 5) for o in others],
                                                                        [round(v, 5) for v in losses]))
    assert abs(float(got["loss"]) - losses[0]) <= 1e-5 * abs(losses[0])
    for k, o in enumerate(others):                          # every rank ran ITS shard (the shards' losses differ)
        assert abs(float(o["loss"]) - losses[k + 1]) <= 1e-4 * abs(losses[k + 1]) + 1e-5, (k + 1, float(o["loss"]), losses[k + 1])
    assert len(set(round(v, 4) for v in losses)) == world
    assert num / den < 1e-4, num / den                      # fp32 atomics order only
    assert int(got["n_buckets"]) >= 4 and int(got["n_hook"]) >= int(got["n_buckets"]) - 1   # launched during backward
    # broadcast at attach: ranks > 0 (initialised with other seeds and shifted running statistics) hold rank 0's state
    for o in others:
        assert np.array_equal(got["w0"], o["w0"]) and np.array_equal(got["rm0"], o["rm0"])
        assert np.array_equal(got["w_after"], o["w_after"])  # identical update on every replica
    assert not np.array_equal(got["w_after"], got["w0"])
