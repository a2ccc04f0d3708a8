"""Worker of tests/test_zhang_gpu.py::test_contentaware_two_rank_data_parallel (one process per rank, torch.distributed.run): one
data-parallel step of the ContentAware backbone + TripletHead (config zhang-orig) - step.attach_reducer gives BOTH trainable conv stacks
(the resnet and the feature extractor) a FlatGradReducer; each rank saves the two reduced flat gradient buffers and its loss."""
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
    dist.init_process_group(os.environ.get("BIHOME_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    from bihome_amd import configs, synth
    from bihome_amd.ddp import shard_range
    from bihome_amd.step import attach_reducer, build_model, build_optimizer
    from bihome_amd.weights import load_synthetic
    import copy
    cfg = copy.deepcopy(configs.get("zhang-orig"))
    trained_masks = len(sys.argv) > 3 and sys.argv[3] == "trained-masks"       # FIX_MASK False: the mask predictor is a third conv stack
    if trained_masks:
        cfg["MODEL"]["BACKBONE"]["FIX_MASK"] = False
    model = build_model(cfg, "cuda")
    load_synthetic(model[0], rank)                          # replicas start DIFFERENT: attach_reducer broadcasts rank 0's
    opt, sched = build_optimizer(model, cfg["SOLVER"])
    red = attach_reducer(model, bucket_bytes=4 << 20)
    d = synth.make_pairs(B, seed=78)
    lo, hi = shard_range(B, rank, world)
    data = {k: torch.tensor(d[k][lo:hi]).cuda() for k in ("patch_1", "patch_2", "delta")}
    model.train()
    opt.zero_grad()
    loss = model(data)[0]
    loss.backward()
    red.allreduce()
    torch.cuda.synchronize()
    extra = {"predictor": model[0].mask_predictor._runner.flat.flat.detach().cpu().numpy()} if trained_masks else {}
    np.savez(out + ".rank%d.npz" % rank, resnet=model[0]._runner.flat.flat.detach().cpu().numpy(),
             extractor=model[0].feature_extractor._runner.flat.flat.detach().cpu().numpy(), loss=loss.item(),
             n_reducers=len(getattr(red, "reducers", [red])), **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
