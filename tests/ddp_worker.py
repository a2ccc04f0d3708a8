"""Worker of tests/test_ddp_gpu.py (one process per rank, launched with torch.distributed.run): one data-parallel
training step of the PRODUCT path - step.attach_reducer (parameter broadcast + FlatGradReducer), net.run_backward's
per-parameter hooks launching the bucketed all-reduce from inside the backward walk, optional wgrad side stream - on this
rank's shard; rank 0 saves the reduced flat gradient, the loss and the updated weights."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, B = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    backend = os.environ.get("BIHOME_DIST_BACKEND", "nccl")
    dist.init_process_group(backend, rank=rank, world_size=world)
    from bihome_amd import configs, synth
    from bihome_amd.ddp import shard_range
    from bihome_amd.step import attach_reducer, build_model, build_optimizer, train_step
    from bihome_amd.weights import load_synthetic
    cfg = configs.get("zeng-bihome")
    model = build_model(cfg, "cuda")
    load_synthetic(model[0], rank)                          # replicas start DIFFERENT: attach_reducer must broadcast rank 0's
    load_synthetic(model[1].auxiliary_resnet, 0)
    with torch.no_grad():
        model[0].layer1[1].running_mean.add_(float(rank))
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    red = attach_reducer(model, bucket_bytes=4 << 20)
    w0 = model[0].layer4[0].upper_branch[0].weight.detach().cpu().clone()
    rm0 = model[0].layer1[1].running_mean.detach().cpu().clone()
    d = synth.make_pairs(B, seed=77)
    g = torch.Generator().manual_seed(5)
    ch = [torch.randint(1, 128 * 128, (B, 128), generator=g) for _ in range(2)]
    lo, hi = shard_range(B, rank, world)
    data = {k: torch.tensor(d[k][lo:hi]).cuda() for k in ("patch_1", "patch_2", "delta")}
    data["choice_12"], data["choice_21"] = ch[0][lo:hi].cuda(), ch[1][lo:hi].cuda()
    # the step, with the optimizer update held back so that the reduced gradient can be saved
    model.train()
    opt.zero_grad()
    loss, _, _ = model(data)
    launched_in_backward = []
    orig = red._launch

    def spy(b):
        launched_in_backward.append(b)
        return orig(b)
    red._launch = spy
    loss.backward()
    n_hook = len(launched_in_backward)
    red.allreduce()
    torch.cuda.synchronize()
    flat = model[0]._runner.flat.flat.detach().cpu().numpy().copy()
    opt.step()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, flat=flat, loss=loss.item(), w0=w0.numpy(), rm0=rm0.numpy(), n_buckets=len(red.buckets), n_hook=n_hook,
                 w_after=model[0].layer4[0].upper_branch[0].weight.detach().cpu().numpy())
    else:
        np.savez(out + ".rank%d.npz" % rank, w0=w0.numpy(), rm0=rm0.numpy(), loss=loss.item(),
                 w_after=model[0].layer4[0].upper_branch[0].weight.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
