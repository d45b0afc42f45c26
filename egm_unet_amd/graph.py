"""hipGraph capture of the training step (forward + criterion + backward + fused SGD).

The eager step issues ~1000 kernel launches through Python autograd; on MI355X their device time is close to the host time
needed to issue them, so the step is captured once into HIP graphs (torch.cuda.CUDAGraph on ROCm is hipGraph) and replayed:
static device buffers, no Python in the loop.

    step = GraphedTrainStep(model, optimizer, example_image, example_target, loss_weight, num_classes, ignore_index=255)
    loss = step(image, target)        # copies the batch into the static buffers, replays, returns the device loss

Single GPU: ONE graph holds forward, criterion, backward and the SGD update.

Data parallel (a GradAllReducer with world > 1): the step is captured as THREE graphs so that the gradient exchange overlaps the
encoder's backward (the reference has no DDP at all; BASELINE.json's north_star asks for exactly this overlap):

    graph A   forward, criterion, backward of the decoder side (out_conv, up4..up1, attn1 = bucket 0 of parallel.py), gather of
              bucket 0 into its flat buffer.  The backward stops at the encoder/decoder boundary tensors (the four skip
              connections and down4's output), whose gradients it leaves in static buffers.
    side stream: all-reduce of bucket 0 (RCCL), enqueued BEFORE graph B is launched
    graph B   backward of the encoder (down4..down1, in_conv = bucket 1) from the boundary gradients, gather of bucket 1
    side stream: all-reduce of bucket 1
    graph C   fused SGD reading the reduced buckets with grad_scale = 1/world (after the main stream has joined the side stream)

The learning rate is read by the SGD kernel from a device scalar (optimizer.lr_dev), refreshed from param_groups before every
replay, so LR schedulers keep working (a captured graph bakes by-value arguments in).  The constructor's warm-up runs
`warmup` REAL eager training steps on the example batch (allocator warm-up, lazy kernel attributes, momentum buffers); pass
restore_after_warmup=True to get weights, BatchNorm statistics and momentum back to their values from before the warm-up.
"""
import itertools
import weakref

import torch

from . import ops
from .train_utils.train_and_eval import criterion


_serial = itertools.count()


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_image, example_target, loss_weight=None, num_classes=2, ignore_index=255,
                 reducer=None, warmup=3, restore_after_warmup=False, split=None):
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.lw, self.nc, self.ign = loss_weight, num_classes, ignore_index
        self.x = example_image.clone()
        self.t = example_target.clone()
        self.trace = []                               # host-order log of the last call (tests assert the overlap structure on it)
        dev = self.x.device
        # ---- learning rate in device memory; managed here unless the caller already does (train_one_epoch)
        lrs = {float(g["lr"]) for g in optimizer.param_groups}
        if len(lrs) > 1:
            # the fused SGD launch of every group reads ONE device scalar; groups at different rates would all train at group 0's
            raise NotImplementedError("GraphedTrainStep: param_groups with different learning rates are not supported "
                                      "(one device-resident lr scalar per optimizer)")
        # whoever created optimizer.lr_dev refreshes it: this step when nobody has, or when the step that did is gone (its weakref is
        # dead) -- otherwise a second step on a long-lived optimizer would replay with a frozen learning rate
        owner = getattr(optimizer, "_lr_owner", None)
        self._own_lr = optimizer.lr_dev is None or (owner is not None and owner() is None)
        if self._own_lr:
            optimizer._lr_owner = weakref.ref(self)
            optimizer.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
            self._lr_pinned = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._lr_event = None
            self._push_lr()
        snapshot = None
        if restore_after_warmup:
            snapshot = ({k: v.clone() for k, v in model.state_dict().items()},
                        {p: (st["momentum_buffer"].clone() if st.get("momentum_buffer") is not None else None)
                         for p, st in optimizer.state.items()}, set(optimizer.state.keys()))
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        # warm-up and capture share one table namespace of their own: the pointer tables the graph re-uploads on every replay are
        # written here and never again (ops.table_namespace)
        self._tag = ("graph", next(_serial))          # never reused (id() of a dead step would hand its stale tables to a new one)
        self._ns = ops.table_namespace(self._tag)
        with torch.cuda.stream(side), self._ns:       # warm-up on a side stream (allocator, lazy kernel attributes, SGD state)
            for _ in range(max(1, warmup)):
                self.warmup_loss = self._eager().detach()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if snapshot is not None:
            with torch.no_grad():
                sd = model.state_dict()
                for k, v in snapshot[0].items():
                    sd[k].copy_(v)
                for p, st in optimizer.state.items():
                    if p in snapshot[1] and snapshot[1][p] is not None:
                        st["momentum_buffer"].copy_(snapshot[1][p])
                    elif st.get("momentum_buffer") is not None:
                        st["momentum_buffer"].zero_()  # created by the warm-up: v = 0, so the first real step gives v = g as torch does
            ops.bump_weight_generation()
        self.opt.zero_grad(set_to_none=True)
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        if self.reducer is not None:
            self.reducer.hooks_enabled = False       # no collectives inside a captured graph
        # With a process group alive, the RCCL watchdog THREAD polls the events of recent collectives (hipEventQuery every ~100 ms,
        # also of completed ones until it has retired them).  Under the default "global" capture mode any such call from any thread
        # while this thread captures is an error that aborts the process (seen on the first real RCCL run: "operation not permitted
        # when stream is capturing" from ProcessGroupNCCL::Watchdog).  Thread-local capture mode confines the check to this thread.
        import torch.distributed as _dist
        self._cap_mode = "thread_local" if (_dist.is_available() and _dist.is_initialized()) else "global"
        want_split = self.reducer is not None and (self.reducer.world > 1 if split is None else split)
        self.split = bool(want_split and hasattr(model, "ddp_boundary") and len(self.reducer.buckets) == 2)
        if self.split:
            self._capture_split()
        else:
            self.graph = torch.cuda.CUDAGraph()
            with self._ns, torch.cuda.graph(self.graph, capture_error_mode=self._cap_mode):
                loss = criterion(self.model(self.x), self.t, self.lw, num_classes=self.nc, ignore_index=self.ign)
                loss.backward(self._one)                  # the seed gradient is a resident tensor, not a fill launch per replay
                if self.reducer is None:
                    self.opt.step()
                self.loss = loss.detach()
        ops.bump_weight_generation()

    # ---- three-graph capture for the overlapped gradient exchange
    def _capture_split(self):
        red, model = self.reducer, self.model
        self.side = red.side if red.side is not None else torch.cuda.Stream(device=self.x.device)
        self.gA, self.gB, self.gC = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        b0, b1 = red.buckets
        with self._ns, torch.cuda.graph(self.gA, capture_error_mode=self._cap_mode):
            model.ddp_boundary = []                   # forward() fills it with the encoder -> decoder tensors
            loss = criterion(model(self.x), self.t, self.lw, num_classes=self.nc, ignore_index=self.ign)
            bnd = list(model.ddp_boundary)
            # the decoder half of backward: stops at the boundary tensors (their gradients land in .grad) and at bucket 0's parameters
            torch.autograd.backward(loss, self._one, inputs=list(b0) + bnd, retain_graph=True)
            red.gather_bucket(0)
            self.loss = loss.detach()
            bgrads = [b.grad for b in bnd]
        with self._ns, torch.cuda.graph(self.gB, pool=self.gA.pool(), capture_error_mode=self._cap_mode):
            torch.autograd.backward(bnd, bgrads, inputs=list(b1))
            red.gather_bucket(1)
        model.ddp_boundary = None
        for b in bnd:
            b.grad = None
        self.opt.grad_source = red.views
        with self._ns, torch.cuda.graph(self.gC, pool=self.gA.pool(), capture_error_mode=self._cap_mode):
            self.opt.step()

    def __del__(self):
        try:
            ops.drop_table_namespace(self._tag)
        except Exception:
            pass

    def _push_lr(self):
        lr = float(self.opt.param_groups[0]["lr"])
        if lr == getattr(self, "_lr_pushed", None):
            return                                    # unchanged since the last upload
        self._lr_pushed = lr
        if self._lr_event is not None:
            self._lr_event.synchronize()              # the previous upload has read the pinned scalar
        self._lr_pinned[0] = lr
        self.opt.lr_dev.copy_(self._lr_pinned, non_blocking=True)
        if self._lr_event is None:
            self._lr_event = torch.cuda.Event()
        self._lr_event.record()

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
        if self._own_lr:
            self._push_lr()
        tr = self.trace = []
        if self.split:
            red, cur = self.reducer, torch.cuda.current_stream()
            self.gA.replay(); tr.append("graph A: forward + decoder backward + gather bucket 0")
            # the trace names a collective only when one was really enqueued (a process group exists); without one the slot is
            # recorded as a bare stream hand-off so the overlap order stays visible
            sent = red.exchange_bucket(0, self.side, cur)
            tr.append("all-reduce bucket 0 enqueued on the side stream" if sent else "bucket 0 handed to the side stream (no process group: no collective)")
            self.gB.replay(); tr.append("graph B: encoder backward + gather bucket 1")
            sent = red.exchange_bucket(1, self.side, cur)
            tr.append("all-reduce bucket 1 enqueued on the side stream" if sent else "bucket 1 handed to the side stream (no process group: no collective)")
            red.join(self.side, cur); tr.append("main stream joined the side stream")
            self.gC.replay(); tr.append("graph C: SGD on the reduced buckets")
        else:
            self.graph.replay(); tr.append("graph: forward + backward" + (" + SGD" if self.reducer is None else ""))
            if self.reducer is not None:
                self.opt.grad_source = self.reducer.reduce_now()
                self.opt.step(); tr.append("exchange + SGD after the replay")
        ops.bump_weight_generation()                  # packed-weight caches are stale after an in-graph update
        return self.loss
