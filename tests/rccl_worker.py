"""Worker of tests/test_rccl_gpu.py: ONE rank on the one MI355X, process group on the "nccl" backend (= RCCL on ROCm).
RCCL loads, the communicator comes up, step.attach_reducer broadcasts the model over it (forced: a one-rank group would
skip it) and a full train_step runs with the bucketed async all-reduce launched from the backward walk on slices of the flat
gradient buffer; a second model takes the same step without a reducer.  With one rank the SUM all-reduce is the identity,
so both must agree - what the test exercises is the RCCL stream semantics that gloo cannot (async_op all-reduce enqueued on
RCCL's stream behind the producing kernels, Work.wait() ordering the optimizer behind it)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, B = sys.argv[1], int(sys.argv[2])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from bihome_amd import configs, synth
    from bihome_amd.step import attach_reducer, broadcast_model, build_model, build_optimizer, train_step
    from bihome_amd.weights import load_synthetic
    cfg = configs.get("zeng-bihome")
    d = synth.make_pairs(B, seed=31)
    g = torch.Generator().manual_seed(6)
    ch = [torch.randint(1, 128 * 128, (B, 128), generator=g).cuda() for _ in range(2)]
    res = {}
    for tag in ("rccl", "plain"):
        model = build_model(cfg, "cuda")
        load_synthetic(model[0], 0)
        load_synthetic(model[1].auxiliary_resnet, 0)
        opt, sched = build_optimizer(model, cfg["SOLVER"])
        red = None
        if tag == "rccl":
            red = attach_reducer(model, bucket_bytes=4 << 20)
            broadcast_model(model, force=True)             # every parameter / buffer through an RCCL broadcast
            launched = []
            orig = red._launch

            def spy(b, orig=orig, launched=launched):
                launched.append(b)
                return orig(b)
            red._launch = spy
        data = {k: torch.tensor(d[k]).cuda() for k in ("patch_1", "patch_2", "delta")}
        data["choice_12"], data["choice_21"] = ch[0], ch[1]
        losses = []
        for it in range(2):
            loss, _, _ = train_step(model, dict(data), opt, sched, reducer=red)
            losses.append(loss.item())
            if it == 0:
                torch.cuda.synchronize()
                res[tag + "_flat"] = model[0]._runner.flat.flat.detach().cpu().numpy().copy()
                if red is not None:
                    res["n_hook"], res["n_buckets"] = len(launched), len(red.buckets)
        torch.cuda.synchronize()
        res[tag + "_loss"] = np.array(losses)
        res[tag + "_w"] = model[0].layer4[0].upper_branch[0].weight.detach().cpu().numpy()
    t = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(t)
    res["allreduce_ok"] = float(t.sum().item())
    res["rccl_version"] = np.array(torch.cuda.nccl.version())
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
