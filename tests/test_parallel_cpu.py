"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce logic (bucket assignment in backward order, hooks,
SUM all-reduce, optimizer reading reduced buckets with 1/world scaling) gives the same update as single-process training
on the concatenated batch.  The HIP gather kernel is replaced by a torch copy here (no GPU in this container); the GPU
path of the same class is exercised by bench.py --gpus N."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.in_conv = torch.nn.Linear(6, 5)
        self.down1 = torch.nn.Linear(5, 5)
        self.up1 = torch.nn.Linear(5, 4)
        self.out_conv = torch.nn.Linear(4, 3)

    def forward(self, x):
        return self.out_conv(torch.tanh(self.up1(torch.tanh(self.down1(torch.tanh(self.in_conv(x)))))))


def _torch_gather(entries):
    for dst, src in entries:
        dst.copy_(src)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egm_unet_amd.parallel import GradAllReducer
    torch.manual_seed(0)
    net = TinyNet()
    red = GradAllReducer(net, gather_fn=_torch_gather, use_side_stream=False)
    assert len(red.buckets) == 2
    names0 = {n for n, p in net.named_parameters() if red.bucket_of[p] == 0}
    assert names0 == {"out_conv.weight", "out_conv.bias", "up1.weight", "up1.bias"}
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    x, y = x_all[rank * 4:(rank + 1) * 4], y_all[rank * 4:(rank + 1) * 4]
    for step in range(2):
        net.zero_grad(set_to_none=True)
        loss = ((net(x) - y) ** 2).sum()           # SUM over the local shard -> all-reduced SUM == full-batch gradient
        loss.backward()
        grads = red.finish()
        with torch.no_grad():
            for p in net.parameters():
                p -= 0.01 * grads[p]
    q.put((rank, {n: p.detach().numpy().tolist() for n, p in net.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process():
    world, port = 2, 29653
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = TinyNet()
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    for step in range(2):
        net.zero_grad(set_to_none=True)
        ((net(x_all) - y_all) ** 2).sum().backward()
        with torch.no_grad():
            for p in net.parameters():
                p -= 0.01 * p.grad
    for r in range(world):
        for n, p in net.named_parameters():
            assert torch.allclose(torch.tensor(results[r][n]), p, rtol=1e-5, atol=1e-6), (r, n)


def test_reducer_detects_incomplete_backward():
    sys.path.insert(0, ROOT)
    from egm_unet_amd.parallel import GradAllReducer
    net = TinyNet()
    red = GradAllReducer(net, world_size=1, gather_fn=_torch_gather, use_side_stream=False)
    with pytest.raises(RuntimeError):
        red.finish()
