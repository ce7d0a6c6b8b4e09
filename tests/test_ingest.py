"""Bag ingest pipeline: order, prefetch depth, error propagation (CPU); data integrity, dtype upcast and real overlap
of host->device copies with compute (GPU)."""
import threading
import time

import pytest
import torch


def _records(n, rows=64, dtype=torch.float32):
    recs = []
    for i in range(n):
        g = torch.Generator().manual_seed(i)
        recs.append((torch.randn(rows + i, 1024, generator=g).to(dtype), i % 18, i % 2, float(i % 2)))
    return recs


def test_order_and_values_cpu(tmp_path):
    from toad_amd.ingest import BagPrefetcher
    recs = _records(7)
    # mix of sources: tensor, callable, .pt path (the reference's wire format)
    path = tmp_path / "slide_3.pt"
    torch.save(recs[3][0], path)
    mixed = list(recs)
    mixed[3] = (str(path),) + recs[3][1:]
    mixed[5] = ((lambda t=recs[5][0]: t),) + recs[5][1:]
    out = list(BagPrefetcher(mixed, "cpu", depth=3, workers=2))
    assert len(out) == 7
    for i, (bag, label, site, sex) in enumerate(out):
        assert torch.equal(bag, recs[i][0]) and int(label) == recs[i][1] and int(site) == recs[i][2] and float(sex) == recs[i][3]


def test_prefetch_runs_ahead_and_errors_propagate_cpu():
    from toad_amd.ingest import BagPrefetcher
    started = []
    lock = threading.Lock()

    def make(i):
        def load():
            with lock:
                started.append(i)
            time.sleep(0.02)
            if i == 4:
                raise RuntimeError("corrupt slide 4")
            return torch.zeros(8, 1024)
        return load

    recs = [(make(i), 0, 0, 0.0) for i in range(6)]
    it = iter(BagPrefetcher(recs, "cpu", depth=2, workers=2))
    next(it)
    time.sleep(0.1)
    assert max(started) >= 2, "loads for later slides must already be running while slide 0 is consumed"
    next(it); next(it); next(it)
    with pytest.raises(RuntimeError, match="corrupt slide 4"):
        next(it)


@pytest.mark.gpu
def test_device_bags_match_and_half_precision_is_upcast(cuda):
    from toad_amd.ingest import BagPrefetcher
    recs = _records(5, rows=300)
    out = list(BagPrefetcher(recs, cuda, depth=2))
    for i, (bag, label, site, sex) in enumerate(out):
        assert bag.is_cuda and bag.dtype == torch.float32 and torch.equal(bag.cpu(), recs[i][0])
        assert int(label) == recs[i][1] and float(sex) == recs[i][3]
    half = _records(3, rows=200, dtype=torch.float16)
    for i, (bag, *_rest) in enumerate(BagPrefetcher(half, cuda, depth=2)):
        assert bag.dtype == torch.float32 and torch.equal(bag.cpu(), half[i][0].float())


@pytest.mark.gpu
def test_copies_overlap_with_compute(cuda):
    """24 bags of 50k patches (205 MB each): consuming them through the prefetcher while the model trains must take clearly less
    than the reference's blocking-copy-then-step loop (i.e. the H2D copies hide behind the kernels). 24, not 8: a new prefetcher's
    copy stream starts with an empty allocator pool, and its first 205 MB hipMalloc (5-6 ms) is a quarter of an 8-bag run."""
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.dp import SlideShardedDP
    from toad_amd.ingest import BagPrefetcher
    torch.manual_seed(0)
    n_bags, rows = 24, 50000
    host = [torch.randn(rows, 1024).pin_memory() for _ in range(2)]
    recs = [(host[i % 2], i % 18, i % 2, 0.0) for i in range(n_bags)]
    model = TOAD_fc_mtl_concat(n_classes=18); model.relocate()
    dp = SlideShardedDP(model, "adam")

    def consume(depth):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for bag, label, site, sex in BagPrefetcher(recs, cuda, depth=depth):
            dp.step([(bag, sex, label, site)], 1)
        torch.cuda.synchronize(); return time.perf_counter() - t0

    def serial():                                # the reference's loop: blocking copy, then the step (utils/core_utils_mtl_concat.py:201-204)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for src, label, site, sex in recs:
            bag = src.to(cuda, non_blocking=False)
            dp.step([(bag, torch.tensor([sex], device=cuda), torch.tensor([label], device=cuda), torch.tensor([site], device=cuda))], 1)
        torch.cuda.synchronize(); return time.perf_counter() - t0

    consume(2); serial()                         # warm-up (allocator, kernels)
    # interleaved trials, best of three each: one slow trial on a shared box (host threads, PCIe) must not decide the test
    t_pipe, t_serial = [], []
    for _ in range(3):
        t_serial.append(serial()); t_pipe.append(consume(3))
    print(f"serial {min(t_serial)*1e3:.1f} ms  pipelined {min(t_pipe)*1e3:.1f} ms  (all: {t_serial} {t_pipe})")
    # copies are ~3.6 ms and steps ~1.4 ms per bag: fully hidden the pipeline takes ~0.77 of the serial loop; 0.92 fails a serialised one
    assert min(t_pipe) < 0.92 * min(t_serial), (t_serial, t_pipe)


@pytest.mark.gpu
def test_prefetcher_prepares_bags_on_the_copy_stream(cuda):
    """BagPrefetcher(prepare=True) yields PreparedBag objects (toad_bag_prepare_f32 on the copy stream, behind the H2D copy): a
    training step on them gives the loss and logits of the fp32 tensors (to the few ulp by which the two operand routes differ:
    tests/test_gpu_pt.py ROUTE_TOL), in record order, with copies still in flight."""
    from toad_amd import TOAD_fc_mtl_concat, ops
    from toad_amd.ingest import BagPrefetcher
    recs = _records(5, rows=700)
    torch.manual_seed(3)
    model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
    w = {k: v.detach() for k, v in model._weights().items()}

    def losses(loader):
        out = []
        for bag, label, site, sex in loader:
            g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
            loss, logits, _ = ops.mil_step(w, g, 0.0, bag, sex, label, site, want_logits=True)
            out.append((loss.cpu(), logits.cpu(), g["w2"].cpu()))
        return out
    plain = losses(BagPrefetcher(recs, cuda, depth=2))
    prep_loader = BagPrefetcher(recs, cuda, depth=3, prepare=True)
    first = next(iter(prep_loader))[0]
    assert getattr(first, "is_prepared_bag", False) and first.shape == (700, 1024)
    prep = losses(prep_loader)
    for a, b in zip(plain, prep):
        for u, v in zip(a, b):
            assert (u - v).abs().max().item() <= 5e-6 * max(u.abs().max().item(), 1e-30)
    with pytest.raises(ValueError):
        BagPrefetcher(recs, cuda, dtype=torch.float16, prepare=True)


def test_landing_arenas_make_consecutive_bags_their_own_concatenation_cpu():
    """BagPrefetcher(arena_rows=R): consecutive fp32 bags land back to back in shared buffers of R rows (what bench.py --config 4 assumes of an
    ingest buffer), values and order unchanged; bags of one buffer are recognised as their own concatenation (ops._adjacent_rows: the ragged
    multi-slide call then takes them without a copy), a bag that does not fit starts a new buffer, an over-long bag gets its own allocation,
    fp16 files are up-cast on landing."""
    from toad_amd import ops
    from toad_amd.ingest import BagPrefetcher
    lens = [300, 200, 400, 100, 1500, 64, 700]
    g = torch.Generator().manual_seed(5)
    bags = [torch.randn(n, 1024, generator=g) for n in lens]
    bags[3] = bags[3].half()
    recs = [(b, i % 18, i % 2, float(i % 2)) for i, b in enumerate(bags)]
    out = list(BagPrefetcher(recs, "cpu", depth=3, workers=2, arena_rows=1000))
    assert len(out) == len(lens)
    for i, (bag, label, site, sex) in enumerate(out):
        assert bag.dtype == torch.float32 and torch.equal(bag, bags[i].float()) and int(label) == i % 18 and int(site) == i % 2
    store = [o[0].untyped_storage().data_ptr() for o in out]
    # 300 + 200 + 400 share buffer 0 (100 more would fit, and does); 1500 > 1000 rows gets its own; 64 + 700 share the next buffer
    assert store[0] == store[1] == store[2] == store[3] and store[4] not in (store[0], store[5]) and store[5] == store[6] != store[0]
    cat = ops._adjacent_rows([o[0] for o in out[:4]])
    assert cat is not None and cat.shape == (1000, 1024) and cat.data_ptr() == out[0][0].data_ptr()
    assert torch.equal(cat, torch.cat([b.float() for b in bags[:4]], 0))
    assert ops._adjacent_rows([out[2][0], out[4][0]]) is None            # different allocations: the caller falls back to torch.cat
    with pytest.raises(ValueError):
        BagPrefetcher(recs, "cpu", dtype=torch.float16, arena_rows=1000)


@pytest.mark.gpu
def test_landed_bags_step_as_one_ragged_call_without_a_copy(cuda):
    """On the device: bags landed by BagPrefetcher(arena_rows=...) go through SlideShardedDP's ragged multi-slide call as a zero-copy view of their
    landing buffer and give bitwise the gradient bucket of the same bags held in separate allocations (which the call concatenates by a copy)."""
    from toad_amd import TOAD_fc_mtl_concat, ops
    from toad_amd.dp import SlideShardedDP
    from toad_amd.ingest import BagPrefetcher
    lens = [700, 64, 1300, 500]
    g = torch.Generator().manual_seed(9)
    recs = [(torch.randn(n, 1024, generator=g), (3 * i) % 18, i % 2, float(i % 2)) for i, n in enumerate(lens)]
    landed = [(b, sx, lb, st) for (b, lb, st, sx) in BagPrefetcher(recs, cuda, depth=3, arena_rows=4096)]
    torch.cuda.synchronize()
    view = ops._adjacent_rows([s[0] for s in landed])
    assert view is not None and view.shape == (sum(lens), 1024) and view.data_ptr() == landed[0][0].data_ptr()
    for (b, _, _, _), r in zip(landed, recs):
        assert torch.equal(b.cpu(), r[0])
    apart = [(b.clone(), sx, lb, st) for (b, sx, lb, st) in landed]
    assert ops._adjacent_rows([s[0] for s in apart]) is None
    out = []
    for slides in (landed, apart):
        torch.manual_seed(2)
        model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
        dp = SlideShardedDP(model, {"lr": 1e-3, "weight_decay": 1e-5})
        dp.accumulate(slides, len(slides), batched=True)
        out.append(dp.flat_grad.clone())
    assert torch.equal(out[0], out[1]) and out[0].abs().max().item() > 0
