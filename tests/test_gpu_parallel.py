"""GPU: the bucketed gradient exchange with the real HIP gather kernel and side stream, 2 ranks sharing cuda:0 over gloo
(the 1-GPU test box cannot host two RCCL ranks; RCCL itself is exercised by bench.py --gpus N on a multi-GPU node)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, graphed=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from egm_unet_amd import UNet
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.parallel import GradAllReducer
    from egm_unet_amd.train_utils import criterion
    torch.manual_seed(0)
    m = UNet(3, 2, base_c=8).to("cuda").train()
    red = GradAllReducer(m, world_size=world)
    assert len(red.buckets) == 2
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    opt.grad_scale = 1.0 / world
    g = torch.Generator().manual_seed(50 + rank)
    x = torch.randn(2, 3, 32, 32, generator=g).cuda()
    t = torch.randint(0, 2, (2, 32, 32), generator=g).cuda()
    lw = torch.tensor([1.0, 2.0], device="cuda")
    from egm_unet_amd.graph import GraphedTrainStep
    if graphed:
        step = GraphedTrainStep(m, opt, x, t, lw, num_classes=2, ignore_index=255, reducer=red, warmup=1)
        for _ in range(2):
            step()
    else:
        for _ in range(2):
            loss = criterion(m(x), t, lw, num_classes=2, ignore_index=255)
            opt.zero_grad()
            loss.backward()
            opt.grad_source = red.finish()
            opt.step()
    torch.cuda.synchronize()
    probe = {k: v.detach().float().cpu().numpy().tolist() for k, v in m.state_dict().items() if k in ("in_conv.0.weight", "out_conv.0.bias")}
    # the averaged gradient of rank-local batches must be identical on both ranks after the all-reduce ...
    # ... and it must BE the mean: one step from the same initial weights with both ranks' batches evaluated in this process,
    # gradients summed and scaled by 1/world, gives the weights of one distributed step (bitwise: a + b is commutative).  In the
    # graphed variant the distributed step is one replay of the three-graph step (warm-up undone by restore_after_warmup).
    torch.manual_seed(0)
    ref, dist_m = UNet(3, 2, base_c=8).to("cuda").train(), UNet(3, 2, base_c=8).to("cuda").train()
    dist_m.load_state_dict(ref.state_dict())
    red2 = GradAllReducer(dist_m, world_size=world, broadcast=False)
    opt_d = SGD(dist_m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4); opt_d.grad_scale = 1.0 / world
    if graphed:
        step2 = GraphedTrainStep(dist_m, opt_d, x, t, lw, num_classes=2, ignore_index=255, reducer=red2, warmup=1, restore_after_warmup=True)
        assert step2.split, "world 2 must take the three-graph step"
        n0 = red2.collectives_issued
        step2()
        assert red2.collectives_issued - n0 == 2, "one all-reduce per bucket per step"
        assert [s for s in step2.trace if s.startswith("all-reduce")] == ["all-reduce bucket 0 enqueued on the side stream",
                                                                          "all-reduce bucket 1 enqueued on the side stream"], step2.trace
    else:
        criterion(dist_m(x), t, lw, num_classes=2, ignore_index=255).backward()
        opt_d.grad_source = red2.finish(); opt_d.step()
    sums = None
    for r in range(world):
        gr = torch.Generator().manual_seed(50 + r)
        xr, tr = torch.randn(2, 3, 32, 32, generator=gr).cuda(), torch.randint(0, 2, (2, 32, 32), generator=gr).cuda()
        probe_m = UNet(3, 2, base_c=8).to("cuda").train()
        probe_m.load_state_dict(ref.state_dict())
        criterion(probe_m(xr), tr, lw, num_classes=2, ignore_index=255).backward()
        gs = [p.grad.clone() for p in probe_m.parameters()]
        sums = gs if sums is None else [a + b for a, b in zip(sums, gs)]
    opt_r = SGD(ref.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4); opt_r.grad_scale = 1.0 / world
    opt_r.grad_source = dict(zip(ref.parameters(), sums)); opt_r.step()
    torch.cuda.synchronize()
    mism = [n for (n, a), b in zip(ref.named_parameters(), dist_m.parameters()) if not torch.equal(a, b)]
    probe["mean_mismatch"] = mism[:4]
    q.put((rank, probe))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graphed", [False, True])
def test_two_ranks_stay_in_sync(graphed):
    world, port = 2, 29671 + int(graphed)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, graphed)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for k in res[0]:
        if k == "mean_mismatch":
            assert res[0][k] == [] and res[1][k] == [], ("the exchanged gradient is not the mean of the ranks' gradients", res[0][k])
            continue
        a, b = torch.tensor(res[0][k]), torch.tensor(res[1][k])
        assert torch.allclose(a, b, rtol=0, atol=0), k          # bitwise: same reduced gradients, same update


def test_bench_two_rank_code_path_rehearsal():
    """bench.py's N>1 path end to end (torch.distributed.run, graph capture with the reducer, exchange + SGD after each replay,
    instrumented step on every rank, max-over-ranks timing) with two ranks sharing this GPU over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EGM_BENCH_SINGLE_DEVICE="1", EGM_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128",
           "--no-cpu-baseline", "--instrument"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["value"] > 0 and line["roofline"] is not None


def test_eager_bucket_gather_sees_finished_conv_gradients():
    """The hook-driven (eager) exchange gathers bucket 0 in the middle of backward, while conv weight gradients are finished by a
    deferred multi-conv reduction: every bucket view must equal the parameter's final .grad (world size 1: gather only)."""
    sys.path.insert(0, ROOT)
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd.parallel import GradAllReducer
    from egm_unet_amd.train_utils import criterion
    torch.manual_seed(0)
    m = GRFBUNet(3, 2, base_c=8).to("cuda").train()
    red = GradAllReducer(m, world_size=1)
    assert len(red.buckets) == 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 64, 64, generator=g).cuda()
    t = torch.randint(0, 2, (2, 64, 64), generator=g).cuda()
    for p in m.parameters():
        p.grad = None
    # poison the allocator's free memory so that an unfinished gradient tensor cannot look right by accident
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(64)]
    del junk
    loss = criterion(m(x), t, torch.tensor([1.0, 2.0], device="cuda"), num_classes=2, ignore_index=255)
    loss.backward()
    views = red.finish()
    torch.cuda.synchronize()
    bad = [n for n, p in m.named_parameters() if not torch.equal(views[p], p.grad)]
    assert not bad, bad[:8]
    red.remove()


def test_reducer_paths_match_plain_training_bitwise():
    """World size 1: two SGD steps (a) without a reducer, (b) with the hook-driven eager exchange, (c) as hipGraph replays followed by
    reduce_now() must leave bit-identical weights -- the exchange only moves gradients around."""
    sys.path.insert(0, ROOT)
    import copy
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd.graph import GraphedTrainStep
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.parallel import GradAllReducer
    from egm_unet_amd.train_utils import criterion
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 64, 64, generator=g).cuda()
    t = torch.randint(0, 2, (2, 64, 64), generator=g).cuda()
    lw = torch.tensor([1.0, 2.0], device="cuda")
    torch.manual_seed(0)
    base = GRFBUNet(3, 2, base_c=8).to("cuda").train()
    sd0 = copy.deepcopy(base.state_dict())

    def run(mode):
        m = GRFBUNet(3, 2, base_c=8).to("cuda").train()
        m.load_state_dict(sd0)
        opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
        red = GradAllReducer(m, world_size=1) if mode != "plain" else None
        if mode in ("graph", "split"):
            step = GraphedTrainStep(m, opt, x, t, lw, num_classes=2, ignore_index=255, reducer=red, warmup=1,   # warm-up = step 1
                                    split=(mode == "split"))
            assert step.split == (mode == "split")
            step()
            if mode == "split":
                # the overlap structure of the data-parallel step: bucket 0's collective is enqueued (side stream) BEFORE the
                # encoder-backward graph is launched, SGD runs as its own graph after the join
                order = [s.split(":")[0] for s in step.trace]
                # (no process group in this test: the two hand-offs to the side stream carry no collective and the trace says so)
                assert order == ["graph A", "bucket 0 handed to the side stream (no process group", "graph B",
                                 "bucket 1 handed to the side stream (no process group", "main stream joined the side stream", "graph C"], step.trace
                assert red.collectives_issued == 0
        else:
            for _ in range(2):
                loss = criterion(m(x), t, lw, num_classes=2, ignore_index=255)
                opt.zero_grad()
                loss.backward()
                if red is not None:
                    opt.grad_source = red.finish()
                opt.step()
        torch.cuda.synchronize()
        if red is not None:
            red.remove()
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    plain, eager, graph, split = run("plain"), run("eager"), run("graph"), run("split")
    bad_e = [k for k in plain if not torch.equal(plain[k], eager[k])]
    bad_g = [k for k in plain if not torch.equal(plain[k], graph[k])]
    bad_s = [k for k in plain if not torch.equal(plain[k], split[k])]
    assert not bad_e, ("eager reducer", bad_e[:6])
    assert not bad_g, ("graphed reducer", bad_g[:6])
    assert not bad_s, ("three-graph step with the exchange between the graphs", bad_s[:6])


def test_three_graph_step_over_rccl_single_rank():
    """The three-graph data-parallel step with the REAL RCCL backend (init_process_group("nccl"), one rank -- all this box can host):
    the all-reduces are enqueued on the side stream between the graph replays and the weights after three steps are bit-identical to
    the single-graph step (tools/rccl_single_rank_rehearsal.py, own process: it owns the process group)."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29573", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank_rehearsal.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all-reduce bucket 0 enqueued on the side stream" in r.stdout and "tensors that differ: 0" in r.stdout
    assert "rccl all-reduces issued in 3 steps: 6" in r.stdout, r.stdout[-2000:]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher on the command line (how a driver may start it): bench.py spawns
    torch.distributed.run itself before touching the GPU and relays rank 0's JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(EGM_BENCH_SINGLE_DEVICE="1", EGM_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["value"] > 0
    assert line["config"]["allreduces_per_step"] == 2, line["config"]


def test_bench_starts_its_own_ranks_single_graph_variant():
    """The same with --dp-single-graph (one captured graph, then the exchange and the SGD launch): the N > 1 line is self-sufficient --
    world size and RCCL version in `config`, no instrumented eager step (roofline / families null without --instrument), and a
    cpu_baseline carried from the checked-in N = 1 line with its source named."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(EGM_BENCH_SINGLE_DEVICE="1", EGM_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128",
                        "--dp-single-graph"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["global_batch"] == 16 and line["value"] > 0
    assert line["config"]["launch"] == "hipGraph replay" and line["config"]["allreduces_per_step"] == 2, line["config"]
    assert line["config"]["rccl_version"], line["config"]
    assert line["roofline"] is None and line["families"] is None
    assert line["cpu_baseline"] is None or line["cpu_baseline"]["source"].startswith("n1 ("), line["cpu_baseline"]


def test_weight_with_tensor_hook_is_not_deferred():
    """A tensor hook on a conv weight makes autograd replace the gradient buffer backward() returned: the deferred slab reduction
    must not be used for it (the hooked gradient would be computed from an unfilled buffer)."""
    sys.path.insert(0, ROOT)
    from egm_unet_amd import UNet
    from egm_unet_amd.train_utils import criterion
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 32, 32, generator=g).cuda()
    t = torch.randint(0, 2, (2, 32, 32), generator=g).cuda()
    lw = torch.tensor([1.0, 2.0], device="cuda")
    torch.manual_seed(0)
    a = UNet(3, 2, base_c=8).to("cuda").train()
    torch.manual_seed(0)
    b = UNet(3, 2, base_c=8).to("cuda").train()
    hooks = [p.register_hook(lambda gr: gr * 1) for p in b.parameters() if p.dim() == 4]
    assert hooks
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(16)]
    del junk
    criterion(a(x), t, lw, num_classes=2, ignore_index=255).backward()
    criterion(b(x), t, lw, num_classes=2, ignore_index=255).backward()
    torch.cuda.synchronize()
    bad = [n for (n, p), q in zip(a.named_parameters(), b.parameters()) if not torch.equal(p.grad, q.grad)]
    assert not bad, bad[:6]
