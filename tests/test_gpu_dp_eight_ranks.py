"""GPU: VALUES of the 8-rank slide-sharded step at BASELINE config 4's shape. Eight processes share cuda:0 (gloo carries the all-reduce: RCCL
refuses several ranks on one device; the driver's 8-GPU run takes the same code path with backend "nccl"); rank r holds slides r, r + 8, ... of
the 64 x 50,000-patch batch (shard_round_robin, bench.py --config 4), runs them through its ragged multi-slide call, and the ONE all-reduce of
the flat 4.77 MB bucket follows. The reduced gradient must equal the gradient a single process computes over the same 64 slides
("DP-reduced grad == mean of per-slide gradients"): a missing, doubled or mis-scaled slide moves it by >= 1/64 of its size.
tests/test_gpu_launch_bench.py checks the launch plumbing and the JSON line of the same configuration; this file checks the numbers.
Reference semantics: utils/core_utils_mtl_concat.py:200-234 (per-slide forward / loss / backward); the reference's only multi-GPU code, the
intra-bag nn.DataParallel of models/model_toad.py:79-81, is intentionally not reproduced."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

WORLD, SLIDES, PATCHES, C = 8, 64, 50_000, 18


def make_slide(i, dev):
    g = torch.Generator(device=dev).manual_seed(1000 + i)              # bench.py make_slide: N(0,1) bag seeded by the slide index
    return (torch.randn(PATCHES, 1024, device=dev, generator=g), torch.tensor([float((i // 2) % 2)], device=dev),
            torch.tensor([i % C], device=dev), torch.tensor([i % 2], device=dev))


def build_model():
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(1)
    m = TOAD_fc_mtl_concat(n_classes=C)
    m.relocate()
    m.train()
    return m


def landed(ids, dev):
    """The rank's bags back to back in one buffer, as bench.py --config 4 and BagPrefetcher(arena_rows=...) hold them."""
    pool = torch.empty((len(ids) * PATCHES, 1024), device=dev)
    out = []
    for k, i in enumerate(ids):
        bag, sx, lb, st = make_slide(i, dev)
        view = pool[k * PATCHES:(k + 1) * PATCHES]
        view.copy_(bag)
        out.append((view, sx, lb, st))
        del bag
    return out


def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from toad_amd.dp import SlideShardedDP, shard_round_robin
    dp = SlideShardedDP(build_model(), {"lr": 1e-4, "weight_decay": 1e-5})
    mine = shard_round_robin(SLIDES, rank, world)
    losses = dp.accumulate(landed(mine, dev), SLIDES)
    dp.reduce()
    torch.cuda.synchronize()
    if rank == 0:
        ret["grad"] = dp.flat_grad.cpu()
    ret[f"loss{rank}"] = torch.stack([l[0] for l in losses]).sum().item()
    ret[f"ids{rank}"] = mine
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_eight_rank_reduced_gradient_equals_the_single_process_gradient(cuda):
    port = 29800 + os.getpid() % 1500
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(worker, args=(WORLD, port, ret), nprocs=WORLD, join=True)
    ids = sorted(i for r in range(WORLD) for i in ret[f"ids{r}"])
    assert ids == list(range(SLIDES))                                     # every slide exactly once
    g8 = ret["grad"]
    # one process, the same 64 slides (seven ragged calls of <= 524,288 rows)
    from toad_amd.dp import SlideShardedDP
    model = build_model()
    dp = SlideShardedDP(model, {"lr": 1e-4, "weight_decay": 1e-5})
    losses = dp.accumulate(landed(list(range(SLIDES)), cuda), SLIDES)
    torch.cuda.synchronize()
    g1 = dp.flat_grad.cpu()
    loss1 = torch.stack([l[0] for l in losses]).sum().item()
    loss8 = sum(ret[f"loss{r}"] for r in range(WORLD))
    assert abs(loss8 - loss1) <= 1e-5 * max(abs(loss1), 1.0)              # mean loss over the 64 slides (the 1/64 is folded into the CE weights)
    offs, _ = model.flat_offsets()
    # The two runs concatenate different groups of slides, so their GEMM operands are scaled per 256-row block of different row ranges: results
    # agree to fp32 round-off, and a pre-activation at round-off of zero may fall on the other side of the ReLU in one of them. The ten gradients no
    # ReLU mask reaches are held to 1e-4 of their scale; the four trunk gradients (3.2 M rows x 1,024 mask bits: a few hundred such flips, each a
    # one-patch rank-one term of <= 1e-3 of the scale) to 2e-3 - two orders below the 1/64 = 1.6e-2 a mis-sharded slide would cost.
    for slot, (o, n) in offs.items():
        a, b = g8[o:o + n], g1[o:o + n]
        scale = max(b.abs().max().item(), 1e-30)
        if slot == "bc":                                                   # exactly zero by the softmax's shift invariance: round-off of dWc-sized terms
            o2, n2 = offs["wc"]
            scale = max(scale, g1[o2:o2 + n2].abs().max().item())
        err = (a - b).abs().max().item()
        tol = 2e-3 if slot in ("w1", "b1", "w2", "b2") else (5e-4 if slot in ("ba", "bb", "bc") else 1e-4)      # (ba / bb / bc: sums of cancelling terms)
        assert err <= tol * scale, (slot, err, scale)
    assert g1.abs().max().item() > 0
