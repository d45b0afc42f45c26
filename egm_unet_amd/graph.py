"""Whole-step hipGraph capture of the training step (forward + criterion + backward + fused SGD).

The eager step issues ~1700 kernel launches through Python autograd; on MI355X their device time (tens of ms) is close
to the host time needed to issue them, so the step is captured once into a HIP graph (torch.cuda.CUDAGraph on ROCm is
hipGraph) and replayed: one host call per step, static device buffers, no Python in the loop.

    step = GraphedTrainStep(model, optimizer, example_image, example_target, loss_weight, num_classes, ignore_index=255)
    loss = step(image, target)        # copies the batch into the static buffers, replays, returns the device loss

With a GradAllReducer (data parallel) the graph holds forward+backward only; the bucketed RCCL all-reduce and the SGD
update run right after each replay on the live stream.
"""
import torch

from . import ops
from .train_utils.train_and_eval import criterion


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_image, example_target, loss_weight=None, num_classes=2, ignore_index=255,
                 reducer=None, warmup=3):
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.lw, self.nc, self.ign = loss_weight, num_classes, ignore_index
        self.x = example_image.clone()
        self.t = example_target.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        # warm-up and capture share one table namespace of their own: the pointer tables the graph re-uploads on every replay are
        # written here and never again (ops.table_namespace)
        self._ns = ops.table_namespace(("graph", id(self)))
        with torch.cuda.stream(side), self._ns:       # warm-up on a side stream (allocator, lazy kernel attributes, SGD state)
            for _ in range(max(1, warmup)):
                self.warmup_loss = self._eager().detach()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        if self.reducer is not None:
            self.reducer.hooks_enabled = False       # no collectives inside the captured graph
        with self._ns, torch.cuda.graph(self.graph):
            loss = criterion(self.model(self.x), self.t, self.lw, num_classes=self.nc, ignore_index=self.ign)
            loss.backward()
            if self.reducer is None:
                self.opt.step()
            self.loss = loss.detach()
        ops.bump_weight_generation()

    def _eager(self):
        loss = criterion(self.model(self.x), self.t, self.lw, num_classes=self.nc, ignore_index=self.ign)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.reducer is not None:
            self.opt.grad_source = self.reducer.finish()
        self.opt.step()
        return loss

    def __call__(self, image=None, target=None):
        if image is not None:
            self.x.copy_(image, non_blocking=True)
            self.t.copy_(target, non_blocking=True)
        self.graph.replay()
        if self.reducer is not None:
            self.opt.grad_source = self.reducer.reduce_now()
            self.opt.step()
        ops.bump_weight_generation()                  # packed-weight caches are stale after an in-graph update
        return self.loss
