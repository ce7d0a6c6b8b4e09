"""CPU, world_size 2 over gloo: the slide-sharded DP plumbing (partitioning, flat gradient bucket,
single all-reduce, mean, identical optimiser step).  Gradients per slide come from the oracle here
(a stand-in for the HIP kernels, which need a GPU); the check is the DP parity definition of
SURVEY.md §7: the reduced gradient equals the mean of the per-slide gradients."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import toad_oracle as orc
from tests.helpers import SLOT2KEY

N_SLIDES = 5
LENS = [40, 7, 33, 12, 25]


def make_slide(i):
    g = torch.Generator().manual_seed(1000 + i)
    return (torch.randn(LENS[i], 1024, generator=g), torch.tensor([float((i // 2) % 2)]),
            torch.tensor([i % 18]), torch.tensor([i % 2]))


def oracle_slide_grad(model, grads, slide, beta, scale):
    params = {k: v.detach() for k, v in model.state_dict().items()}
    bag, sex, label, site = slide
    _, loss, g = orc.fwd_bwd(params, bag, sex, label, site)
    for slot, key in SLOT2KEY.items():
        grads[slot].mul_(beta).add_(g[key] * scale)
    return torch.stack([loss * scale, loss, loss])


def worker(rank, world, port, ret, n_slides=N_SLIDES):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.dp import SlideShardedDP, shard_round_robin
    torch.manual_seed(100 + rank)                       # replicas start DIFFERENT: the broadcast must fix that
    model = TOAD_fc_mtl_concat(n_classes=18)
    dp = SlideShardedDP(model, lambda ps: torch.optim.SGD(ps, lr=0.1), slide_grad_fn=oracle_slide_grad)
    start = model.flat_parameters().clone()
    mine = shard_round_robin(n_slides, rank, world)
    dp.flat_grad.fill_(7.0)                             # stale contents: a rank without slides must still contribute ZERO
    dp.step([make_slide(i) for i in mine], n_slides)
    ret[rank] = (start, dp.flat_grad.clone(), model.flat_parameters().clone(), mine,
                 {k: v.clone() for k, v in model.state_dict().items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_world2_gradient_is_mean_of_slide_gradients():
    world, port = 2, 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, port, ret), nprocs=world, join=True)
    (s0, g0, p0, m0, sd0), (s1, g1, p1, m1, sd1) = ret[0], ret[1]
    assert sorted(m0 + m1) == list(range(N_SLIDES)) and not set(m0) & set(m1)
    assert torch.equal(s0, s1), "replicas must be identical after the construction broadcast"
    assert torch.equal(g0, g1) and torch.equal(p0, p1), "every rank holds the same reduced gradient / parameters"
    # reference: mean of per-slide oracle gradients at the broadcast parameters (rank 0's init)
    torch.manual_seed(100)
    from toad_amd import TOAD_fc_mtl_concat
    ref_model = TOAD_fc_mtl_concat(n_classes=18)
    params = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    mean = {k: torch.zeros_like(v) for k, v in params.items()}
    for i in range(N_SLIDES):
        bag, sex, label, site = make_slide(i)
        _, _, g = orc.fwd_bwd(params, bag, sex, label, site)
        for k in mean:
            mean[k] += g[k] / N_SLIDES
    for k in params:
        new = sd0[k]
        assert torch.allclose(new, params[k] - 0.1 * mean[k], atol=1e-6, rtol=1e-5), k


@pytest.mark.timeout(300)
def test_dp_rank_without_slides_contributes_zero():
    """global_slides < world (the tail of an epoch, or BASELINE config 4 on more ranks than slides): the rank that holds no slide
    zeroes its bucket, still joins the ONE all-reduce and takes the same optimiser step - nobody hangs, every replica stays identical."""
    world, n_slides, port = 3, 2, 31500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, port, ret, n_slides), nprocs=world, join=True)
    assert [ret[r][3] for r in range(world)] == [[0], [1], []]
    for r in (1, 2):
        assert torch.equal(ret[0][1], ret[r][1]) and torch.equal(ret[0][2], ret[r][2]), r
    torch.manual_seed(100)
    from toad_amd import TOAD_fc_mtl_concat
    params = {k: v.detach().clone() for k, v in TOAD_fc_mtl_concat(n_classes=18).state_dict().items()}
    mean = {k: torch.zeros_like(v) for k, v in params.items()}
    for i in range(n_slides):
        _, _, g = orc.fwd_bwd(params, *make_slide(i))
        for k in mean:
            mean[k] += g[k] / n_slides
    for k in params:
        assert torch.allclose(ret[2][4][k], params[k] - 0.1 * mean[k], atol=1e-6, rtol=1e-5), k


def test_partitioners():
    from toad_amd.dp import shard_by_length, shard_round_robin
    assert [shard_round_robin(10, r, 4) for r in range(4)] == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    lens = [100, 10, 10, 10, 50, 50, 30]
    parts = [shard_by_length(lens, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == list(range(len(lens)))
    loads = [sum(lens[i] for i in p) for p in parts]
    assert max(loads) <= 100 and min(loads) >= 80


def test_dp_constructor_rejects_batch_sizes_the_ragged_call_cannot_take():
    """batch_rows above the ragged multi-slide call's row limit (32-bit row offsets of the 1024-wide operand) or a per-slide threshold above
    batch_rows used to fail with a shape error in the MIDDLE of accumulate(), after earlier calls had already written into the gradient bucket;
    both are constructor errors now (toad_amd/dp.py)."""
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.dp import SlideShardedDP
    model = TOAD_fc_mtl_concat(n_classes=18)
    mk = lambda **kw: SlideShardedDP(model, lambda ps: torch.optim.SGD(ps, lr=0.1), slide_grad_fn=oracle_slide_grad, **kw)    # noqa: E731
    with pytest.raises(ValueError, match="batch_rows"):
        mk(batch_rows=SlideShardedDP.MAX_BATCH_ROWS + 1)
    with pytest.raises(ValueError, match="batch_rows"):
        mk(batch_rows=0)
    with pytest.raises(ValueError, match="batch_max_patches"):
        mk(batch_rows=100_000, batch_max_patches=100_001)
    dp = mk(batch_rows=SlideShardedDP.MAX_BATCH_ROWS, batch_max_patches=0)
    assert dp.batch_rows == SlideShardedDP.MAX_BATCH_ROWS and dp.batch_max_patches == 0
    assert mk().batch_rows == SlideShardedDP.BATCH_ROWS
