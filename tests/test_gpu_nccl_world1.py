"""GPU: the RCCL path of slide-sharded data parallelism, executed on the one GPU there is.

torch.distributed backend "nccl" IS RCCL on ROCm; RCCL refuses two ranks on one device, so the only way to run the real
collective on a 1-GPU box is a world of ONE. That still creates the RCCL communicator for this device, launches RCCL's all-reduce
kernel on the stream the step's kernels run on, and goes through exactly the code SlideShardedDP.step executes at N = 8
(toad_amd/dp.py: accumulate -> ONE dist.all_reduce(SUM) of the flat 4.77 MB bucket -> flat Adam). A sum over one rank must leave
the bucket bit-for-bit unchanged, so the step with the collective must equal the step without it bitwise. Runs in a spawned
process: a process group is global state that must not leak into the other tests.
The reference's multi-GPU code (nn.DataParallel inside a bag, models/model_toad.py:79-81) is not reproduced."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

C = 18


def _slides(dev):
    out = []
    for i, n in enumerate((700, 2049)):
        g = torch.Generator().manual_seed(3000 + i)
        out.append(tuple(t.to(dev) for t in (torch.randn(n, 1024, generator=g), torch.tensor([float(i % 2)]),
                                              torch.tensor([(5 * i) % C]), torch.tensor([i % 2]))))
    return out


def _build():
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(7)
    m = TOAD_fc_mtl_concat(n_classes=C)
    m.relocate()
    m.train()
    return m


def worker(rank, port, ret):
    import torch.distributed as dist
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from toad_amd import launch
    from toad_amd.dp import SlideShardedDP
    launch.init_process_group("nccl", device=dev, timeout_s=120)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls["n"] += 1
        return real(t, *a, **k)
    dist.all_reduce = counting
    try:
        model = _build()
        dp = SlideShardedDP(model, {"lr": 1e-3, "weight_decay": 1e-5}, always_reduce=True)   # the construction broadcast is skipped at world 1
        slides = _slides(dev)
        for _ in range(2):
            dp.step(slides, len(slides))
        torch.cuda.synchronize()
        n_calls = calls["n"]
        # the collective on a side stream as well: RCCL work is stream-ordered, not device-synchronous
        s = torch.cuda.Stream()
        buf = dp.flat_grad.clone()
        with torch.cuda.stream(s):
            real(buf, op=dist.ReduceOp.SUM)
        s.synchronize()
        same = torch.equal(buf, dp.flat_grad)
    finally:
        dist.all_reduce = real
    ret["with"] = (dp.flat_grad.cpu().clone(), model.flat_parameters().cpu().clone(), n_calls, same)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_step_through_rccl_allreduce_at_world_1(cuda):
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(29900 + os.getpid() % 1000, ret), nprocs=1, join=True)
    g_nccl, p_nccl, n_calls, same = ret["with"]
    assert n_calls == 2, "one all-reduce per optimiser step"
    assert same, "a sum over one rank must not change the bucket"
    # the same two steps with no process group at all
    from toad_amd.dp import SlideShardedDP
    model = _build()
    dp = SlideShardedDP(model, {"lr": 1e-3, "weight_decay": 1e-5})
    slides = _slides(torch.device("cuda", 0))
    for _ in range(2):
        dp.step(slides, len(slides))
    assert torch.equal(dp.flat_grad.cpu(), g_nccl) and torch.equal(model.flat_parameters().cpu(), p_nccl)
