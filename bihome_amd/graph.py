"""One training step captured in a HIP graph (hipGraph via torch.cuda.CUDAGraph).

Every launch of the step - ~285 kernels of the C ABI (ctypes calls on the capturing stream), the head's small torch ops,
fused Adam - has static shapes and allocates nothing outside torch's caching allocator, so the whole of
train.py:296-387 (zero_grad, forward, backward, [clip], optimizer step) replays as ONE hipGraphLaunch: the host cost of a
step drops from ~285 launches (~0.3 ms of gaps at the step boundary, first-order once the bf16 path shortens the
kernels) to one.  The learning-rate scheduler and the BatchNorm call counters stay on the host (per-iteration
MultiStepLR writes the device-resident lr tensor only when a milestone is crossed).

    gs = GraphedStep(model, opt, sched, example_batch)     # warm-up + capture
    loss, delta_gt, delta_hat = gs(batch)                   # copies the batch into the static inputs, replays
"""
import torch

from . import net


class GraphedStep:

    def __init__(self, model, opt, sched, example, clip=-1.0, loss_fn="biHomE", warmup=3, keys=None):
        from .step import train_step
        if not all(pg.get("capturable", False) for pg in opt.param_groups):
            raise ValueError("GraphedStep needs an optimizer built with capturable=True (step.build_optimizer(..., capturable=True))")
        self.model, self.opt, self.sched = model, opt, sched
        self.keys = list(keys) if keys is not None else [k for k, v in example.items() if torch.is_tensor(v)]
        self.static = {k: example[k].detach().clone() for k in self.keys}
        self._bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up: lazy state (runners, flat gradients, Adam state,
            for _ in range(warmup):                        # per-device kernel attributes) exists before the capture
                train_step(model, dict(self.static), opt, None, clip=clip, loss_fn=loss_fn)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        before = [getattr(m, "_bh_pending_batches", 0) for m in self._bns]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = train_step(model, dict(self.static), opt, None, clip=clip, loss_fn=loss_fn)
        # host-side BatchNorm call counters advance once per replay by what one captured step added; the capture itself
        # recorded the step without running it, so its own increment is taken back
        self._bn_inc = [getattr(m, "_bh_pending_batches", 0) - b for m, b in zip(self._bns, before)]
        for m, b in zip(self._bns, before):
            if hasattr(m, "_bh_pending_batches"):
                m._bh_pending_batches = b
        self.warmup_steps = warmup                         # optimizer steps already taken on the example batch (the capture records, it does not run)

    def __call__(self, data):
        for k in self.keys:
            self.static[k].copy_(data[k], non_blocking=True)
        self.graph.replay()
        # the replay ran no Python: the captured fused Adam and BatchNorm kernels changed weights and running statistics
        # without any of the eager forward's cache bookkeeping (version counters do not move either)
        net.invalidate_caches(self.model)
        for m, inc in zip(self._bns, self._bn_inc):
            if inc:
                m._bh_pending_batches = getattr(m, "_bh_pending_batches", 0) + inc
        if self.sched is not None:
            self.sched.step()                              # train.py:387 (per-iteration MultiStepLR)
        return self.out
