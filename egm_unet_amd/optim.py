"""Fused multi-tensor SGD on the HIP library: the drop-in for torch.optim.SGD(lr, momentum, weight_decay) as
constructed at train.py:115-118 of the reference (same param_groups / state_dict layout, so LambdaLR schedulers and
checkpoints work unchanged).  One kernel launch updates every parameter."""
import torch

from . import ops
from ._lib import lib, ptr, stream


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError("egm_unet_amd.optim.SGD: dampening/nesterov are not used by the reference")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.grad_scale = 1.0          # e.g. 1/world_size when gradients arrive as an all-reduced SUM
        self.grad_source = None        # optional {param: fp32 tensor view} overriding p.grad (DDP buckets)
        self.lr_dev = None             # optional device fp32 scalar read by the kernel instead of the by-value lr (a captured hipGraph
                                       # bakes by-value arguments in; a per-iteration LR schedule needs the value in memory)
        self._tables = {}              # (namespace, group index, first) -> [key, pinned host table, device table, upload event]

    @staticmethod
    def _alloc_table(cap, dev):
        return torch.empty((cap, 5), dtype=torch.int64).pin_memory(), torch.empty((cap, 5), dtype=torch.int64, device=dev)

    def _device_table(self, slot, rows, dev):
        slot = (ops._table_tag[0],) + slot            # a captured graph's tables are its own (ops.table_namespace)
        """Pointer table on the device.  Re-uploaded only when a pointer changed; the upload is an async copy from a
        pinned staging buffer, so it is legal inside hipGraph capture (and replays re-copy the same bytes)."""
        key = tuple(rows)
        hit = self._tables.get(slot)
        if hit is not None and hit[0] == key:
            return hit[2]
        n = len(rows)
        if hit is None or hit[1].shape[0] < n:
            hit = self._tables[slot] = [None, *self._alloc_table(max(n, 512), dev), None]
        # the rewrite of the pinned buffer waits for the previous upload from it (ops.upload_pinned)
        ops.upload_pinned(hit, 1, 2, 3, torch.tensor(rows, dtype=torch.int64), n)
        hit[0] = key
        return hit[2]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            for first in (True, False):           # staging buffers exist before any hipGraph capture can need them
                if (ops._table_tag[0], gi, first) not in self._tables and group["params"]:
                    cap = max(512, len(group["params"]))
                    self._tables[(ops._table_tag[0], gi, first)] = [None, *self._alloc_table(cap, group["params"][0].device), None]
            rows = {True: [], False: []}          # first-step parameters take v = g (no stale buffer read)
            keep = []
            for p in group["params"]:
                g = self.grad_source.get(p) if self.grad_source is not None else p.grad
                if g is None:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("egm_unet_amd.optim.SGD needs contiguous fp32 parameters")
                st = self.state[p]
                is_first = st.get("momentum_buffer") is None
                g = g if (g.is_contiguous() and g.dtype == torch.float32) else g.contiguous().float()
                keep.append(g)
                buf = None
                if group["momentum"] != 0:
                    if is_first:
                        st["momentum_buffer"] = torch.empty_like(p)
                    buf = st["momentum_buffer"]
                rows[is_first].append((p.data_ptr(), g.data_ptr(), buf.data_ptr() if buf is not None else 0, p.numel()))
            for first in (True, False):
                if not rows[first]:
                    continue
                ch = lib().cdll.egm_sgd_chunk()
                chunks, full = 0, []
                for r in rows[first]:                # fifth field: index of the tensor's first workgroup
                    full.append(r + (chunks,))
                    chunks += (r[3] + ch - 1) // ch
                table = self._device_table((gi, first), full, group["params"][0].device)
                lib().call("egm_sgd_multi", ptr(table), len(full), chunks, ptr(self.lr_dev), float(group["lr"]), float(group["momentum"]),
                           float(group["weight_decay"]), float(self.grad_scale), 1 if first else 0, stream())
        ops.bump_weight_generation()
        return loss
