"""bench.py — benchmarks of the TOAD gated-attention MIL hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5                      # headline (the driver's run)
    python bench.py --gpus N --steps K --warmup W                        # N > 1 started plainly: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W           # (what the driver runs for N > 1; same thing)
    python bench.py --config 2 | 3 | 4 | 5 [--gpus N]                    # BASELINE.json configs 2-5 (5 = bench_extract.py)
    python bench.py --dropin [--patches N]                               # the reference's own call sequence, timed

Headline (no --config). A "step" = one optimiser step of slide-sharded data parallel training: every rank runs
forward + weighted CE + backward over its own synthetic 100,000-patch x 1024-d bag - the RAW fp32 [N,1024] tensor, already
resident in HBM, measured (by the first GEMM while it converts it, DESIGN.md 5 RUN) and split into the GEMMs' operand pieces INSIDE the step like every other activation -,
ONE all-reduce of the 4.77 MB flat gradient over RCCL when N > 1, then Adam. value = slides/s over the whole job (N slides per
step / max-over-ranks step time). Weak scaling. (The reference reloads every bag every epoch, datasets/dataset_mtl_concat.py:369-373 +
utils/core_utils_mtl_concat.py:201-234, so nothing about a bag may be computed outside the timed region.) The JSON line also carries
  prepared_pipelined  the ingest pipeline's form of the same loop (toad_amd/ingest.py BagPrefetcher(prepare=True)): while step i runs,
                 bag i+1 - a raw fp32 device tensor - is converted on a side stream into the plane-tiled two-piece format the first
                 Linear and its weight gradient take by LDS-DMA (toad_bag_prepare_f32); every byte-touching pass is inside the timed
                 loop, value = steps / wall of the whole loop;
  prepared_resident  the step alone on bags prepared BEFORE the timed region (round 3's headline; an ingest-format number, not credited);
  roofline       the fused gated-attention pooling forward (the kernel BASELINE.json's metric names): algorithmic bytes
                 4*[N*(2D+L+T)+T*D+T+T*L] per launch / mean launch time measured with HIP events on the launch stream inside
                 the timed steps, vs 8 TB/s HBM;
  roofline_mfma  all eight GEMM calls of a step: 6,029,312*N algorithmic FLOP / their summed event time. The GEMMs carry
                 every fp32 operand as two fp16 pieces and every product as THREE fp16 MFMA terms (fp32-equivalent accuracy,
                 csrc/gemm_h2.inc), so the ceiling is the dense fp16 peak / 3 = 833.3 TFLOP/s fp32-equivalent (the issued
                 fp16 MFMA rate is reported next to it; round 1's six-term bf16 form had 416.7, the exact-fp32 MFMA peak is
                 157.3);
  sustained      the same step repeated for >= 8 s right after the timed region (the K = 20 default lasts ~50 ms, shorter than
                 the clock governor's settling time and than the sampling period of an external utilisation monitor);
  allreduce      the 4.77 MB flat-gradient all-reduce over RCCL on the launch stream, HIP-event timed (at N = 1 a world-1 RCCL
                 communicator is created for it: the collective path executes on the one GPU there is);
  dropin         the reference's own call sequence on the same bag: model(data, sex), torch CrossEntropyLoss x2,
                 loss.backward(), torch.optim.Adam(model.parameters()).step(), zero_grad()
                 (utils/core_utils_mtl_concat.py:206-234), which is what a reference user gets without touching the harness;
  cpu_baseline   the CPU oracle (structurally the reference's PyTorch-CPU op sequence, pinned to the reference in
                 oracle/pin_against_reference.py) + torch Adam, timed on this box's host cores at its best thread count;
  batched        the data-parallel trainer's own mode on the same bags: five slides per optimiser step through ONE ragged multi-slide call;
  ingest         (round 6) the same step fed from page-locked HOST memory through toad_amd.ingest.BagPrefetcher(depth=2) - what the reference's loader does per
                 slide (datasets/dataset_mtl_concat.py:369-373 + utils/core_utils_mtl_concat.py:201) - for fp32 and fp16 bags: slides/s, host-to-device GB/s
                 against the copy alone, copy / compute overlap. Never `value` (the contract's bags are resident).

--config 2   1 GPU, single 100k-patch bag, fused gated-attention pool FORWARD only; value = algorithmic GB/s; the oracle's
             gated_pool_fwd timed on the host cores beside it.
--config 3   1 GPU, full step (fwd + CE + bwd + Adam) on 10,000-patch bags, 52 per optimiser step = one full ragged multi-slide call (520k rows);
             roofline (pool forward launches of the batch) + roofline_mfma + cpu_baseline (>= 10 repetitions); `per_slide` (round 6) = ONE optimiser
             step per slide, the reference's own train_loop semantics (utils/core_utils_mtl_concat.py:200-234), with its own GEMM roofline fraction
             (`--slides-per-rank 1` makes that the line's `value`).
--config 4   64 slides x 50,000 patches per step, slide i on rank i mod G (shard_round_robin), one gradient all-reduce and one
             Adam step per 64 slides: STRONG scaling over G = --gpus; runs at G = 1 too; roofline + roofline_mfma + cpu_baseline
             (one 50,000-patch slide on the host cores).
Every N > 1 line carries per_rank: each rank's own step time (min / max / per rank), its pool-forward time and the all-reduce time it
measured, and scaling_efficiency_vs_per_rank_min = fastest rank's own step time / the job's step time (max over ranks, closing barrier included),
so that a scaling run can attribute lost efficiency to load imbalance, the collective or clock spread between devices without a re-run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch
import torch.distributed as dist

L0, L, D, T, C = 1024, 512, 384, 2, 18
GEMM_FLOP_PER_PATCH = 6_029_312            # 2,359,296 fwd + 3,670,016 bwd
HBM_PEAK = 8.0e12                          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F16_PEAK = 2.5e15                     # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md; 2:1-sparse figures are never used)
SPLIT_TERMS = 3                            # fp16 MFMAs per fp32-equivalent product (x*s = h+m: hh + hm + mh)
MFMA_EQ_PEAK = MFMA_F16_PEAK / SPLIT_TERMS    # 833.3 TFLOP/s fp32-equivalent ceiling of the two-piece GEMMs


def pool_fwd_bytes(n):                     # 5,128 B/patch + constants
    return 4 * (n * (2 * D + L + T) + T * D + T + T * L)


def _physical_cores() -> int:
    try:
        pairs = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def _cpu_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


POOL_TRAFFIC_FILE = "r06_pool_traffic.json"     # written by tools/pool_traffic_json.py from the PMC passes of `tools/gpu_run.sh TAG traffic`
POOL_KERNEL_SOURCES = ("toad_amd/csrc/gated_pool.hip", "toad_amd/csrc/common.h")


def pool_kernel_sha() -> str:
    """sha256 over the sources the pool kernels are compiled from: what ties a committed traffic measurement to a kernel build."""
    import hashlib
    h = hashlib.sha256()
    for rel in POOL_KERNEL_SOURCES:
        with open(os.path.join(REPO, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def measured_pool_traffic(n: int):
    """HBM bytes per launch of the fused pool forward from THIS round's committed rocprofv3 PMC passes (profiles/r06_pool_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as the gfx950 guide prescribes for 16-B/lane streaming loads). The file records the sha256
    of the pool kernels' sources at measurement time: the figure is reported only for that N and while those sources are unchanged, else None
    (JSON null) - an edit to the pool kernels silently invalidates nothing."""
    try:
        with open(os.path.join(REPO, "profiles", POOL_TRAFFIC_FILE)) as f:
            t = json.load(f)
        if int(t["patches"]) == int(n) and t.get("kernel_source_sha256") == pool_kernel_sha():
            return float(t["pool_fwd_hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


PROBE_BUDGET_S = 10.0      # wall-clock cap of the thread-count probe (the driver's lease should be GPU time, not host probing)


def _probe_threads(once, budget_s: float, min_reps: int):
    """PyTorch-CPU does not scale to every hardware thread of a big host (256 threads run ~7x slower than 64 here), and the
    best count depends on the problem size: probe 32 threads first (the winner on every box measured so far), then 16 / 64 /
    physical while the probe has spent less than PROBE_BUDGET_S, one repetition each after a warm-up; then time the FASTEST
    (median of >= min_reps) - the baseline is the best this CPU path can do on this box within a bounded sample."""
    logical = os.cpu_count() or 1
    phys = min(_physical_cores(), logical)
    # (hosts with >= 64 physical cores: 64 threads are probed before 16 - on a 128-core EPYC the 10 s probe budget used to end after {32, 16})
    order = (32, 64, 16, phys, 8) if phys >= 64 else (32, 16, 64, phys, 8)
    cands = [t for t in order if t <= max(phys, 4)]
    cands = list(dict.fromkeys(cands)) or [max(1, phys)]
    probe = {}
    t_start = time.perf_counter()
    for th in cands:
        torch.set_num_threads(th)
        once()                                   # warm-up at this thread count
        probe[th] = once()
        if time.perf_counter() - t_start > min(PROBE_BUDGET_S, budget_s * 0.5):
            break
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    times = [probe[best]]
    while len(times) < min_reps or (time.perf_counter() - t_start < budget_s and len(times) < max(7, min_reps)):
        times.append(once())
    times.sort()
    return times[len(times) // 2], best, probe, len(times), phys, logical


def cpu_baseline_step(n_patches: int, budget_s: float = 30.0, min_reps: int = 3):
    """fwd + weighted CE + bwd of the CPU oracle + torch.optim.Adam step on the host cores; bounded sample."""
    from oracle import toad_oracle as orc       # checker / reported baseline only
    params = {k: torch.nn.Parameter(v.clone()) for k, v in orc.xavier_params(C, seed=1).items()}
    opt = torch.optim.Adam(params.values(), lr=1e-4, weight_decay=1e-5)          # get_optim defaults (main_mtl_concat.py:93-96)
    x = torch.randn(n_patches, L0, generator=torch.Generator().manual_seed(1000))
    sex = torch.tensor([0.0]); label = torch.tensor([0]); site = torch.tensor([0])

    def once():
        t0 = time.perf_counter()
        with torch.no_grad():
            _, _, g = orc.fwd_bwd({k: v.detach() for k, v in params.items()}, x, sex, label, site)
        for k, p in params.items():
            p.grad = g[k]
        opt.step()
        opt.zero_grad()
        return time.perf_counter() - t0

    med, best, probe, reps, phys, logical = _probe_threads(once, budget_s, min_reps)
    return {"value": round(1.0 / med, 4), "unit": "slides/s", "cores": best, "kind": "port",
            "sample": f"median of {reps} x (fwd + weighted CE + bwd + Adam step) of one {n_patches}-patch x 1024-d bag, "
                      f"oracle/toad_oracle.py (torch CPU fp32, the reference's op sequence) on {_cpu_name()}, "
                      f"{phys} physical / {logical} logical cores; threads probed {{"
                      + ", ".join(f"{k}: {v * 1e3:.0f} ms" for k, v in probe.items()) + "}",
            "ms_per_slide": round(med * 1e3, 2)}


def cpu_baseline_pool(n_patches: int, budget_s: float = 20.0):
    """The oracle's gated_pool_fwd (tanh / sigmoid gate, scores, softmax over the bag, A @ H) on the host cores."""
    from oracle import toad_oracle as orc       # checker / reported baseline only
    g = torch.Generator().manual_seed(1000)
    p = torch.randn(n_patches, 2 * D, generator=g); h = torch.randn(n_patches, L, generator=g).relu()
    wc = torch.randn(T, D, generator=g) * 0.05; bc = torch.zeros(T)

    def once():
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.gated_pool_fwd(p[:, :D], p[:, D:], h, wc, bc)
        return time.perf_counter() - t0

    med, best, probe, reps, phys, logical = _probe_threads(once, budget_s, 5)
    return {"value": round(pool_fwd_bytes(n_patches) / med / 1e9, 2), "unit": "GB/s", "cores": best, "kind": "port",
            "sample": f"median of {reps} x oracle.gated_pool_fwd on one {n_patches}-patch bag (P [N,768], H [N,512] resident in host memory), "
                      f"{_cpu_name()}, {phys} physical / {logical} logical cores; threads probed {{"
                      + ", ".join(f"{k}: {v * 1e3:.1f} ms" for k, v in probe.items()) + "}",
            "ms_per_launch": round(med * 1e3, 3)}


BAG_DTYPE = torch.float32          # --bag-dtype fp16: bags stored in half precision (a side experiment, never the headline)
BAG_PREPARED = False               # --bag-format prepared: bags converted to the ingest format BEFORE the timed region (not the headline)


def make_slide(idx: int, n: int, dev, prepared=None):
    """SURVEY.md 8(d) synthetic inputs: N(0,1) fp32 bag seeded by the slide index, generated on the device. With `prepared` it is
    then brought into the resident ingest format (ops.prepare_bag: the two fp16 pieces of every fp32 element, plane-tiled; same
    4 bytes per element, same values to 2^-22) and the fp32 tensor is dropped - what toad_amd.ingest does once per slide."""
    g = torch.Generator(device=dev).manual_seed(1000 + idx)
    bag = torch.randn(n, L0, device=dev, generator=g).to(BAG_DTYPE)
    if (BAG_PREPARED if prepared is None else prepared) and BAG_DTYPE == torch.float32 and n >= 64:
        from toad_amd import ops
        bag = ops.prepare_bag(bag)
    return (bag, torch.tensor([float((idx // 2) % 2)], device=dev), torch.tensor([idx % C], device=dev), torch.tensor([idx % 2], device=dev))


def time_prepare(n: int, dev, reps: int = 5):
    """HIP-event time of ops.prepare_bag on one resident fp32 bag (abs-max pass + split pass): the one-off ingest cost per slide."""
    from toad_amd import ops
    x = torch.randn(n, L0, device=dev)
    ops.prepare_bag(x)
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.prepare_bag(x); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return round(t[len(t) // 2] * 1e3, 1)


def gemm_roofline(timing, rows_total):
    """All GEMM calls of the instrumented steps: algorithmic 6,029,312 FLOP per patch row x the rows those steps processed / the
    summed HIP-event time of the calls (per-slide calls: 8 per slide; the ragged multi-slide call: 8 per batch)."""
    gemm_t = sum(timing[k][1] for k in ("gemm_fwd", "gemm_dgrad", "gemm_wgrad")) * 1e-3
    calls = timing["gemm_fwd"][0] // 3                                  # library calls (3 forward GEMMs each)
    tf = GEMM_FLOP_PER_PATCH * rows_total / gemm_t
    return {"bound": "mfma",
            "kernel": "gemm_nt_h2_big_kernel x5 (+split_planes_h2, nt_fixup_h2) + gemm_tn_h2_big_kernel x3 (+slab_reduce_h2)",
            "achieved": round(tf / 1e12, 2), "peak": round(MFMA_EQ_PEAK / 1e12, 1),
            "unit": "TFLOP/s fp32-equivalent (algorithmic 2MNK; every product = 3 fp16 MFMA terms, peak = 2500/3)",
            "frac": round(tf / MFMA_EQ_PEAK, 4), "traffic": None,
            "fp16_mfma_tflops_issued": round(tf * SPLIT_TERMS / 1e12, 1),
            "frac_of_round1_six_term_ceiling_416.7": round(tf / (MFMA_F16_PEAK / 6), 4),
            "fp32_mfma_peak_for_reference": 157.3,
            # tools/ubench/mfma_power (profiles/r02av): the matrix pipe ALONE, on random fp16 operands, sustains 1735 of the nominal 2500
            # TFLOP/s (1.7 GHz; the nominal rate needs zeros) - context for `frac`, which stays against the nominal peak
            "frac_of_measured_mfma_only_rate_578": round(tf / (1735e12 / SPLIT_TERMS), 4),
            "algorithmic_flops": GEMM_FLOP_PER_PATCH * rows_total, "rows": rows_total, "library_calls": calls,
            "us_per_call": round(gemm_t / max(calls, 1) * 1e6, 1)}


def time_prepared_pipelined(dp, raw_slides, global_slides: int, steps: int, warmup: int, dev, sync):
    """The ingest pipeline's loop with every pass inside the timed region (toad_amd/ingest.py, BagPrefetcher(prepare=True)): while step i
    runs on the launch stream, bag i+1 - a raw fp32 [N,1024] device tensor, as the host-to-device copy leaves it - is measured and
    converted into the plane-tiled two-piece format on a SIDE stream (toad_bag_prepare_f32: abs-max pass + split pass, 820 MB read +
    410 MB written per 100k-patch bag), into one of two alternating buffers. Events order it: a step waits for its bag's conversion,
    a conversion waits for the step that last read its buffer. Returns seconds per step (wall of the whole loop / steps)."""
    from toad_amd import ops
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    nraw = len(raw_slides)
    bufs = [ops.prepare_bag(raw_slides[b % nraw][0][0]) for b in range(2)]          # the two ingest buffers (contents overwritten below)
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()

    def convert(i):                                   # bag i -> buffer i % 2, on the side stream
        with torch.cuda.stream(side):
            side.wait_event(free[i % 2])
            ops.prepare_bag(raw_slides[i % nraw][0][0], out=bufs[i % 2])
            ready[i % 2].record(side)

    def run(k):
        for b in range(2):
            free[b].record(main)
        convert(0)
        for i in range(k):
            convert(i + 1)                            # overlaps step i
            main.wait_event(ready[i % 2])
            _, sex, label, site = raw_slides[i % nraw][0]
            dp.step([(bufs[i % 2], sex, label, site)], global_slides)
            free[i % 2].record(main)

    run(max(warmup, 2))
    sync()
    t0 = time.perf_counter()
    run(steps)
    sync()
    return (time.perf_counter() - t0) / steps


def time_ingest(dp, n: int, dev, slides_per_leg: int = 12):
    """SURVEY.md 8(f) row 2 / the reference's loader (datasets/dataset_mtl_concat.py:369-373 `torch.load`, utils/core_utils_mtl_concat.py:201 `.to(device)`):
    bags that arrive from HOST memory. `slides_per_leg` 100k-patch bags cycle through four page-locked host buffers and toad_amd.ingest.BagPrefetcher
    (depth 2: the host-to-device copy of bag i+1 runs on its own HIP stream while step i computes) into the headline step (one optimiser step per
    slide). Two wire formats: fp32 (410 MB per bag) and fp16 (205 MB; handed to the model as fp16: toad_mil_step_x16_f32). Beside each: the copy
    alone (the link's rate on this box) and the step alone on resident bags, from which
        overlap = (t_copy + t_step - t_pipelined) / min(t_copy, t_step)      (1 = the shorter of the two is fully hidden, 0 = serial).
    Never `value`: the headline's bags are resident (bench contract); this is the rate a training run fed over PCIe sees."""
    from toad_amd.ingest import BagPrefetcher
    out = {}
    nbuf = 4
    for wire, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
        host = []
        for i in range(nbuf):
            bag, sx_, lb_, st_ = make_slide(200 + i, n, dev, prepared=False)
            hb = torch.empty((n, L0), dtype=dt, pin_memory=True)
            hb.copy_(bag.to(dt))
            host.append(hb)
            del bag
        torch.cuda.synchronize()
        nbytes = host[0].numel() * host[0].element_size()
        # the copy alone (pinned -> device, one stream)
        land = torch.empty((n, L0), dtype=dt, device=dev)
        for _ in range(2):
            land.copy_(host[0], non_blocking=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(slides_per_leg):
            land.copy_(host[i % nbuf], non_blocking=True)
        torch.cuda.synchronize(); t_copy = (time.perf_counter() - t0) / slides_per_leg
        # the step alone on the landed bag
        sx_, lb_, st_ = torch.tensor([1.0], device=dev), torch.tensor([3], device=dev), torch.tensor([1], device=dev)
        for _ in range(2):
            dp.step([(land, sx_, lb_, st_)], 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(slides_per_leg):
            dp.step([(land, sx_, lb_, st_)], 1)
        torch.cuda.synchronize(); t_step = (time.perf_counter() - t0) / slides_per_leg
        del land

        def run(k):                                           # the pipeline: BagPrefetcher(depth 2) feeding the step
            recs = [(host[i % nbuf], i % C, i % 2, float((i // 2) % 2)) for i in range(k)]
            for bag, lb, st, sx in BagPrefetcher(recs, dev, depth=2, workers=2, dtype=dt):
                dp.step([(bag, sx, lb, st)], 1)
        run(3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(slides_per_leg)
        torch.cuda.synchronize(); t_pipe = (time.perf_counter() - t0) / slides_per_leg
        ov = (t_copy + t_step - t_pipe) / max(min(t_copy, t_step), 1e-9)
        out[wire] = {"value": round(1.0 / t_pipe, 2), "unit": "slides/s", "ms_per_slide": round(t_pipe * 1e3, 3), "slides": slides_per_leg,
                     "bytes_per_bag": nbytes, "h2d_gbps_in_pipeline": round(nbytes / t_pipe / 1e9, 1),
                     "copy_alone": {"ms": round(t_copy * 1e3, 3), "gbps": round(nbytes / t_copy / 1e9, 1)},
                     "step_alone_ms": round(t_step * 1e3, 3), "overlap": round(max(0.0, min(1.0, ov)), 3),
                     "bound": "host-to-device link" if t_copy >= t_step else "compute"}
        del host
    out["what"] = ("100k-patch bags from page-locked host memory through toad_amd.ingest.BagPrefetcher(depth=2) into one optimiser step per slide "
                   "(datasets/dataset_mtl_concat.py:369-373 + utils/core_utils_mtl_concat.py:201 in the reference); fp16 = features stored as fp16, taken "
                   "by the fp16-bag kernels without an up-cast pass; copy_alone = the pinned host-to-device rate of this box; not `value` (resident bags)")
    return out


def time_dropin(n: int, steps: int, warmup: int, dev, host_reads: bool):
    """The reference's train_loop body (utils/core_utils_mtl_concat.py:201-234) on the drop-in module, resident bags.
    host_reads adds what the reference's loop reads back per slide (two loss .item(), Y_hat / site_hat for the loggers)."""
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(1)
    model = TOAD_fc_mtl_concat(dropout=False, n_classes=C)
    model.relocate()
    model.train()
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4, weight_decay=1e-5)   # get_optim, utils/utils.py:63-65
    loss_fn = torch.nn.CrossEntropyLoss()
    slides = [make_slide(i, n, dev, prepared=False) for i in range(2)]      # the reference's loop hands fp32 tensors to model(data, sex)

    def one(i):
        data, sex, label, site = slides[i % 2]
        res = model(data, sex)
        cls_loss = loss_fn(res["logits"], label)
        site_loss = loss_fn(res["site_logits"], site)
        loss = cls_loss * 0.75 + site_loss * 0.25
        if host_reads:
            _ = (int(res["Y_hat"]), int(label), int(res["site_hat"]), int(site), cls_loss.item(), site_loss.item())
        loss.backward()
        opt.step()
        opt.zero_grad()

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def time_allreduce(dp, dev, world: int, backend: str):
    """The gradient all-reduce of a step, alone: dist.all_reduce(SUM) of the flat fp32 bucket on the launch stream, HIP-event
    timed over 20 launches after 5 warm-ups. At world 1 (the driver's default run) a one-rank RCCL communicator is created
    first, so the collective path executes on hardware even when only one GPU exists; failures are reported, never raised."""
    info = {"bytes": int(dp.flat_grad.numel() * 4), "backend": "nccl (RCCL)" if backend == "nccl" else backend, "world": world}
    try:
        if not dist.is_initialized():
            from toad_amd import launch
            launch.init_process_group(backend, device=dev if backend == "nccl" else None, timeout_s=60)
            info["communicator"] = "created for this measurement (world 1)"
        buf = dp.flat_grad.clone()
        for _ in range(5):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record(); dist.all_reduce(buf, op=dist.ReduceOp.SUM); b.record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        info.update({"us": round(us[len(us) // 2], 2), "us_min": round(us[0], 2), "launches": len(us),
                     "unchanged_at_world_1": bool(world > 1 or torch.equal(buf, dp.flat_grad))})
    except Exception as e:                                   # noqa: BLE001 - a bench line is still worth printing without it
        info["error"] = f"{type(e).__name__}: {e}"[:300]
    return info


_REAL_STDOUT = None


def claim_stdout() -> None:
    """Keep stdout for THE json line alone: file descriptor 1 is pointed at stderr for the rest of the run (RCCL prints a version
    banner through C stdio when a communicator is created, libraries print notices), the original descriptor is kept for emit()."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(out: dict) -> None:
    """Write THE json line - the only thing that reaches the caller's stdout (claim_stdout)."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)                       # whatever sits in C stdio buffers goes where fd 1 points now (stderr)
    except Exception:                                        # noqa: BLE001
        pass
    sys.stdout.flush()
    line = (json.dumps(out) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.buffer.write(line); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def run_config2(args, dev):
    """Pool forward only, one resident 100k-patch bag (BASELINE config 2)."""
    from toad_amd import ops
    n = args.patches or 100_000
    g = torch.Generator(device=dev).manual_seed(1000)
    p = torch.randn(n, 2 * D, device=dev, generator=g); h = torch.randn(n, L, device=dev, generator=g).relu_()
    wc = torch.randn(T, D, device=dev, generator=g) * 0.05; bc = torch.zeros(T, device=dev)
    for _ in range(args.warmup):
        ops.gated_pool_fwd(p, D, h, wc, bc)
    torch.cuda.synchronize()
    ops.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ops.gated_pool_fwd(p, D, h, wc, bc)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    calls, tot_ms = ops.collect_timing()["pool_fwd"]
    ops.enable_timing(False)
    t = tot_ms / calls * 1e-3
    bw = pool_fwd_bytes(n) / t
    out = {"metric": "fused gated-attention pool forward, algorithmic HBM GB/s (100k-patch x 1024-d bag)", "value": round(bw / 1e9, 1),
           "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE config 2: toad_gated_pool_fwd_f32 (tanh*sigmoid gate, scores, online softmax, weighted sum; "
                                  f"models/model_toad.py:37-40,92,97-98) on one resident {n}-patch bag: P [N,768], H [N,512] -> A_raw [N,2], M [2,512]"},
           "roofline": {"bound": "hbm", "kernel": "gated_pool_fwd_kernel<2,3,4,true> + gated_pool_combine_kernel", "achieved": round(bw / 1e9, 1),
                        "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(bw / HBM_PEAK, 4), "traffic": measured_pool_traffic(n),
                        "algorithmic_bytes": pool_fwd_bytes(n), "us_per_launch": round(t * 1e6, 2)}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_pool(n)
        out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    emit(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bag-dtype", choices=["fp32", "fp16"], default="fp32",
                    help="fp16: feature bags stored in half precision (toad_mil_step_x16_f32: two MFMA terms in the first layer, no abs-max pass); "
                         "reported under its own metric name, BASELINE's configurations are fp32")
    ap.add_argument("--bag-format", choices=["fp32", "prepared"], default="fp32",
                    help="fp32 (default, the headline): the raw [N,1024] fp32 tensor, measured and split inside every step; prepared: bags "
                         "converted to the ingest format of toad_bag_prepare_f32 BEFORE the timed region (an ingest-format experiment, "
                         "reported under its own metric name)")
    ap.add_argument("--no-prepared-legs", action="store_true", help="headline run without the prepared_pipelined / prepared_resident legs")
    ap.add_argument("--patches", type=int, default=0, help="patches per slide (default: 100,000; config 3: 10,000; config 4: 50,000)")
    ap.add_argument("--slides-per-rank", type=int, default=0,
                    help="slides per rank per optimiser step (default 1; config 3: 52 = one full ragged call of dp.BATCH_ROWS = 524,288 rows). Several small fp32 slides of a rank go through ONE "
                         "ragged multi-slide call (toad_mil_multi_step_f32: the GEMMs run once over the concatenated bags)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE.json config (0 = the headline step; 5 = bench_extract.py)")
    ap.add_argument("--sustain-seconds", type=float, default=8.0, help="length of the sustained leg after the timed region (0 = skip)")
    ap.add_argument("--dropin", action="store_true", help="time only the reference's call sequence on the drop-in module")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="headline run without the extra drop-in timing")
    ap.add_argument("--no-ingest", action="store_true", help="headline run without the host-memory ingest legs (fp32 / fp16 bags over PCIe through BagPrefetcher)")
    ap.add_argument("--ingest-slides", type=int, default=12, help="slides per ingest leg")
    ap.add_argument("--no-per-slide", action="store_true", help="--config 3 without the per_slide leg (one optimiser step per 10k-patch slide: the reference's own semantics)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--ragged", action="store_true",
                    help="config 4 only: log-normal slide lengths (same 64 x 50k = 3.2 M patches in total, sigma 0.7, seeded) dealt with "
                         "dp.shard_by_length instead of 64 equal slides round robin - what a real cohort looks like (docs/README.md: 22k WSIs of very "
                         "different size); per_rank.patches_per_step / step_ms then show the imbalance. Reported under its own metric name.")
    ap.add_argument("--single-device", action="store_true",
                    help="plumbing test only: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate GPUs)")
    args = ap.parse_args()
    global BAG_DTYPE, BAG_PREPARED
    BAG_DTYPE = torch.float16 if args.bag_dtype == "fp16" else torch.float32
    BAG_PREPARED = args.bag_format == "prepared"

    from toad_amd import launch
    if args.config == 5:                                   # BASELINE config 5 lives in bench_extract.py (same contract, same flags)
        import bench_extract
        sys.argv = [os.path.join(REPO, "bench_extract.py"), "--gpus", str(args.gpus)] + \
                   (["--steps", str(args.steps)] if "--steps" in sys.argv else []) + \
                   (["--warmup", str(args.warmup)] if "--warmup" in sys.argv else []) + \
                   (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        return bench_extract.main()
    # `python bench.py --gpus N` started plainly: spawn the N ranks (torch.distributed.run, 127.0.0.1) and exit with their code
    launch.maybe_self_launch(__file__, sys.argv[1:], args.gpus, single_device=args.single_device)
    claim_stdout()                                         # from here on only emit() writes to the caller's stdout
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one process per GPU (or run `python bench.py --gpus N` plainly)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # a process group exists whenever the ranks were started by torch.distributed.run - also at world 1, where the gradient all-reduce
    # is then issued every step like at N > 1 (the collective path on one GPU: tests/test_gpu_launch_bench.py); the plain
    # `python bench.py` run has none and skips the collective inside the step
    under_torchrun = launch.launched_by_torchrun()
    if world > 1 or under_torchrun:
        launch.init_process_group(args.backend, device=dev if args.backend == "nccl" else None)

    if args.config == 2:
        return run_config2(args, dev)
    if args.dropin:
        n = args.patches or 100_000
        ms = time_dropin(n, args.steps, args.warmup, dev, host_reads=False)
        ms_h = time_dropin(n, args.steps, args.warmup, dev, host_reads=True)
        emit({"metric": f"drop-in train_loop body, {n}-patch bags", "value": round(1e3 / ms, 2), "unit": "slides/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "ms_per_step_with_host_reads": round(ms_h, 4),
                          "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "model(data, sex) + nn.CrossEntropyLoss x2 + loss.backward() + torch.optim.Adam.step() + zero_grad() "
                                                 "(utils/core_utils_mtl_concat.py:206-234) on toad_amd.TOAD_fc_mtl_concat, resident bags"}})
        return

    from toad_amd import TOAD_fc_mtl_concat, ops
    from toad_amd.dp import SlideShardedDP, shard_by_length, shard_round_robin

    torch.manual_seed(1)                                   # main_mtl_concat.py:89 default seed
    model = TOAD_fc_mtl_concat(dropout=False, n_classes=C)
    model.relocate()
    model.train()
    dp = SlideShardedDP(model, {"lr": 1e-4, "weight_decay": 1e-5}, always_reduce=under_torchrun)      # get_optim defaults (main_mtl_concat.py:93-96), HIP flat Adam

    ragged_lens = None
    if args.config == 4:
        n = args.patches or 50_000
        global_slides = 64
        if args.ragged:
            import numpy as np
            raw = np.random.Generator(np.random.PCG64(4)).lognormal(mean=0.0, sigma=0.7, size=global_slides)
            ragged_lens = np.maximum(1024, np.round(raw / raw.sum() * global_slides * n)).astype(np.int64).tolist()
            mine = shard_by_length(ragged_lens, rank, world)           # longest-processing-time greedy: cost ~ patches
            lens = [ragged_lens[i] for i in mine]
        else:
            mine = shard_round_robin(global_slides, rank, world)       # slide i -> rank i mod G
            lens = [n] * len(mine)
        # the rank's bags lie back to back in ONE resident buffer (what an ingest buffer filled slide after slide looks like): the ragged
        # multi-slide call takes consecutive bags as their own concatenation, without copying a row (ops._adjacent_rows)
        pool = torch.empty((sum(lens), L0), device=dev, dtype=BAG_DTYPE)
        slides = [[]]
        off = 0
        for i, ni in zip(mine, lens):
            bag, sex_, label_, site_ = make_slide(i, ni, dev)
            if getattr(bag, "is_prepared_bag", False):                 # --bag-format prepared: per-slide calls on prepared bags
                slides[0].append((bag, sex_, label_, site_))
                continue
            view = pool[off:off + ni]
            off += ni
            view.copy_(bag)
            slides[0].append((view, sex_, label_, site_))             # the same 64 resident slides every step
            del bag
        nbags = 1
        scaling = "strong"
    else:
        n = args.patches or (10_000 if args.config == 3 else 100_000)
        spr = args.slides_per_rank or (52 if args.config == 3 else 1)       # 52 x 10k = 520k rows: one full ragged call of the default size (dp.BATCH_ROWS)
        if spr > 1 and n <= SlideShardedDP.BATCH_MAX_PATCHES:
            BAG_PREPARED = False                                       # the ragged batch call concatenates raw fp32 bags
        nbags = 2                                                      # alternate two resident bags per slide slot
        slides = [[make_slide((rank * spr + s) * nbags + b, n, dev) for s in range(spr)] for b in range(nbags)]
        if spr > 1 and BAG_DTYPE == torch.float32 and not BAG_PREPARED:
            # several slides per step: like config 4, the bags of one step lie back to back in ONE resident buffer - the layout the ingest produces
            # (toad_amd.ingest.BagPrefetcher(arena_rows=dp.batch_rows) lands consecutive slides that way) - so the ragged multi-slide call takes
            # them as their own concatenation; separately allocated bags would be concatenated by a copy inside every timed step (round 5 measured
            # that copy at 0.88 ms of 11.4 on 52 x 10k patches, profiles/r05j_config3_kernel_stats.md)
            for b in range(nbags):
                landing = torch.empty((spr * n, L0), device=dev, dtype=torch.float32)
                for i, (bag, sx_, lb_, st_) in enumerate(slides[b]):
                    view = landing[i * n:(i + 1) * n]
                    view.copy_(bag)
                    slides[b][i] = (view, sx_, lb_, st_)
                del bag
        global_slides = spr * world
        scaling = "weak"
    patches_per_rank_step = sum(int(sl[0].shape[0]) for sl in slides[0])
    prepared = BAG_PREPARED and args.bag_dtype == "fp32" and n >= 64
    batched = len(slides[0]) > 1 and n <= SlideShardedDP.BATCH_MAX_PATCHES and args.bag_dtype == "fp32" and not prepared

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        dp.step(slides[i % nbags], global_slides)
    # Timed region: only the dominant HBM-bound kernel (the fused pool forward) is bracketed by HIP events (2 pre-created events
    # per library call). Bracketing all eight GEMM calls as well costs 18 event packets per call (~0.1 ms of stream time: 4 % of a
    # 100k-patch step, 2x of a 256-patch step), so the GEMM breakdown is measured in its own instrumented loop right after.
    # (every call of a 100k-patch step is bracketed, as the contract's "events over the timed region" reads; for short bags every 4th call: two event
    #  packets cost ~11 us of stream time, 0.5 % of the headline step but 3 % of a 10k-patch one - profiles/r06a_*)
    ev_stride = 1 if patches_per_rank_step >= 50_000 else 4
    ops.enable_timing(True, level=1, prealloc=2 * args.steps * len(slides[0]), stride=ev_stride)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = dp.step(slides[i % nbags], global_slides)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0               # this rank's own time for its K steps (before the closing barrier)
    sync()
    elapsed = time.perf_counter() - t0
    timing = ops.collect_timing()
    timed_calls = ops.timing_call_count()                  # library calls of the timed region (every ev_stride-th one carried events)
    k_instr = min(args.steps, 10)
    ops.enable_timing(True, level=2, prealloc=18 * k_instr * len(slides[0]))
    for i in range(k_instr):
        dp.step(slides[i % nbags], global_slides)
    timing_gemm = ops.collect_timing()
    ops.enable_timing(False)
    sync()
    last_loss = float(losses[-1][0].item()) * global_slides

    # sustained rate: keep stepping for >= --sustain-seconds (same barrier + synchronize bracketing); long enough for the clock
    # governor to settle and for an external utilisation sampler (the driver's gpu_busy) to land inside it
    sus_steps, sus_t = 0, 0.0
    if args.config in (0, 3) and args.sustain_seconds > 0:
        chunk = max(args.steps, 10)
        sync()
        t1 = time.perf_counter()
        while True:
            for i in range(chunk):
                dp.step(slides[i % nbags], global_slides)
            sync()
            sus_steps += chunk
            sus_t = time.perf_counter() - t1
            flag = torch.tensor([1.0 if sus_t >= args.sustain_seconds else 0.0], device=dev)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)             # every rank leaves the loop together
            if flag.item() > 0 or sus_steps >= 200000:
                break

    allreduce = time_allreduce(dp, dev, world, args.backend)
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed, sus_t], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, sus_t = float(t[0].item()), float(t[1].item())
        # what a scaling run needs to attribute lost efficiency: every rank's own step time (load imbalance / clock spread between
        # devices), its pool-forward launch time (a bandwidth-bound probe of that device) and the all-reduce time it measured
        pool_us = (timing["pool_fwd"][1] / timing["pool_fwd"][0] * 1e3) if "pool_fwd" in timing else float("nan")
        mine_t = torch.tensor([local_elapsed / args.steps * 1e3, pool_us, float(allreduce.get("us", float("nan"))), float(patches_per_rank_step)],
                              device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allr, mine_t)
        rows = [[float(v) for v in r.tolist()] for r in allr]
        per_rank = {"step_ms": [round(r[0], 4) for r in rows], "step_ms_min": round(min(r[0] for r in rows), 4),
                    "step_ms_max": round(max(r[0] for r in rows), 4), "pool_fwd_us": [round(r[1], 2) for r in rows],
                    "allreduce_us": [round(r[2], 2) for r in rows], "patches_per_step": [int(r[3]) for r in rows],
                    "what": "step_ms = a rank's own wall time per step up to its local synchronize (before the closing barrier); ms_per_step of "
                            "the line is the max over ranks including the barrier"}
        # what a first real multi-GPU run needs to attribute lost scaling without a re-run: the fastest rank's own step time over the job's
        # step time (max over ranks, closing barrier included). 1.0 = nobody waited; the gap is imbalance + collective + barrier skew.
        per_rank["scaling_efficiency_vs_per_rank_min"] = round(per_rank["step_ms_min"] / (elapsed / args.steps * 1e3), 4)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = global_slides * args.steps / elapsed
        metric = {0: "slides/sec fwd+bwd, 100k-patch x 1024-d bags", 3: "slides/sec fwd+bwd, 10k-patch x 1024-d bags (BASELINE config 3)",
                  4: "slides/sec fwd+bwd, 64 slides x 50k patches per step, slide-sharded DP (BASELINE config 4)"}[args.config]
        if args.config == 0 and n != 100_000:
            metric = f"slides/sec fwd+bwd, {n}-patch x 1024-d bags"
        if ragged_lens is not None:
            metric = ("slides/sec fwd+bwd, 64 slides of LOG-NORMAL length (3.2 M patches in total, %d..%d per slide) per step, length-balanced "
                      "slide-sharded DP (config 4 made ragged; not the BASELINE line)" % (min(ragged_lens), max(ragged_lens)))
        if args.bag_dtype == "fp16":
            metric += " STORED AS fp16 (not a BASELINE configuration)"
        if prepared:
            metric += " PREPARED BEFORE THE TIMED REGION (ingest-format experiment, not the headline)"
        out = {
            "metric": metric,
            "value": round(value, 3), "unit": "slides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32 (storage + accumulation; GEMM operands as 2 x f16 pieces, 3 MFMA terms)" + ("" if args.bag_dtype == "fp32" else "; bag stored as f16"),
            "data": "synthetic",
            "config": {"workload": f"TOAD_fc_mtl_concat(big, n_classes=18) fwd + 0.75/0.25 CE + bwd + Adam on "
                                   f"{len(slides[0])} x {n}-patch x 1024-d N(0,1) fp32 bag(s) per GPU per step, bags resident in HBM"
                                   + (" in the ingest format (toad_bag_prepare_f32 BEFORE the timed region: both fp16 pieces of every fp32 element, plane-tiled, 4 B/element)"
                                      if prepared else " as raw fp32 tensors; measuring the bag (abs-max, inside the first GEMM) and splitting it into the GEMM operand pieces happen inside every timed step")
                                   + (f"; 64 slides per optimiser step dealt round robin over {world} rank(s)" if args.config == 4 else "")
                                   + ("; the bags of a step lie back to back in one resident buffer (the layout toad_amd.ingest.BagPrefetcher(arena_rows=...) lands them in)"
                                      if batched else ""),
                       "arithmetic": "fp32 storage/accumulation; GEMM operands as two fp16 pieces (x*s = h+m, power-of-two scales), 3 MFMA terms = "
                                     "fp32-equivalent, verified vs fp64 (tools/split_emulation.py, tests/test_gpu_h2.py)",
                       "patches_per_slide": n, "slides_per_step": global_slides,
                       "batching": ("consecutive slides of a rank share ragged multi-slide calls of at most %d rows (" % SlideShardedDP.BATCH_ROWS + "toad_mil_multi_step_f32: trunk / "
                                    "attention GEMMs once over the concatenated bags, pooling + heads + loss per slide; bags that are not already "
                                    "adjacent in memory are concatenated inside the timed step)"
                                    if batched else "one library call per slide"),
                       "parallelism": f"slide-sharded dp{world}, one {4 * model.flat_parameters().numel() / 1e6:.2f} MB grad all-reduce/step"
                                      + (" (issued at world 1 too: started by torch.distributed.run)" if under_torchrun and world == 1 else "")},
        }
        if "pool_fwd" in timing:
            calls, tot_ms = timing["pool_fwd"]
            pool_t = tot_ms / calls * 1e-3
            calls_per_step = timed_calls / args.steps                    # per-slide calls: slides per rank; ragged batch call: 1
            slides_per_call = len(slides[0]) / calls_per_step
            # one launch pools every slide of the call (blockIdx.y = slide); ragged slides: the bytes of this rank's rows, per call on average
            pool_bytes = pool_fwd_bytes(n) * slides_per_call if ragged_lens is None else pool_fwd_bytes(patches_per_rank_step) / calls_per_step
            pool_bw = pool_bytes / pool_t
            out["roofline"] = {"bound": "hbm", "kernel": "gated_pool_fwd_kernel<2,3,4,true> + gated_pool_combine_kernel"
                                                         + (f" (batched launch: {slides_per_call:g} slides, blockIdx.y = slide)" if slides_per_call > 1 else ""),
                               "achieved": round(pool_bw / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": round(pool_bw / HBM_PEAK, 4), "traffic": measured_pool_traffic(n) if slides_per_call == 1 else None,
                               "algorithmic_bytes": int(pool_bytes), "us_per_launch": round(pool_t * 1e6, 2)}
            out["roofline_mfma"] = gemm_roofline(timing_gemm, k_instr * patches_per_rank_step)
            out["roofline_mfma"]["measured_in"] = f"{k_instr} instrumented steps right after the timed region (18 events per library call)"
            out["op_us_per_slide"] = {k: round(v[1] / (k_instr * len(slides[0])) * 1e3, 1) for k, v in timing_gemm.items()}
        if sus_steps:
            out["sustained"] = {"value": round(global_slides * sus_steps / sus_t, 3), "unit": "slides/s", "steps": sus_steps,
                                "seconds": round(sus_t, 3), "ms_per_step": round(sus_t / sus_steps * 1e3, 3)}
        out["allreduce"] = allreduce
        if per_rank is not None:
            out["per_rank"] = per_rank
        out["last_loss"] = round(last_loss, 5)
        if world == 1 and args.config == 0 and not prepared and args.bag_dtype == "fp32" and n >= 64 and ops.x16_ok(n) and not args.no_prepared_legs:
            # (b) the ingest pipeline's loop: conversion of bag i+1 on a side stream while step i runs, everything inside the timed loop
            sec = time_prepared_pipelined(dp, slides, global_slides, args.steps, args.warmup, dev, sync)
            out["prepared_pipelined"] = {"value": round(global_slides / sec, 3), "unit": "slides/s", "ms_per_step": round(sec * 1e3, 3), "steps": args.steps,
                                         "what": "step i on a plane-tiled two-piece bag while toad_bag_prepare_f32 converts raw fp32 bag i+1 on a side stream "
                                                 "(two alternating buffers, event-ordered); value = steps / wall of the whole loop, conversions included"}
            # (c) round 3's headline: bags converted before the timed region (an ingest-format number; not credited)
            pre = [[(ops.prepare_bag(sl[0]),) + tuple(sl[1:]) for sl in group] for group in slides]
            for i in range(3):
                dp.step(pre[i % nbags], global_slides)
            sync(); t2 = time.perf_counter()
            for i in range(args.steps):
                dp.step(pre[i % nbags], global_slides)
            sync(); pre_ms = (time.perf_counter() - t2) / args.steps * 1e3
            del pre
            out["prepared_resident"] = {"value": round(global_slides * 1e3 / pre_ms, 3), "unit": "slides/s", "ms_per_step": round(pre_ms, 3), "steps": args.steps,
                                        "what": "the step alone on bags converted BEFORE the timed region (toad_mil_step_xp_f32); the conversion it leaves out costs",
                                        "prepare_us_per_bag": time_prepare(n, dev)}
        if world == 1 and args.config == 0 and len(slides[0]) == 1 and not prepared and args.bag_dtype == "fp32" and not args.no_prepared_legs:
            # (d) the data-parallel trainer's own mode on the same bags: FIVE slides per optimiser step, landed back to back, through ONE ragged multi-slide
            # call (toad_mil_multi_step_f32: 500k rows fill the 256 x 256 tile plan and the per-call helpers are paid once per five slides). `value` above
            # stays the reference's semantics - one optimiser step per slide (utils/core_utils_mtl_concat.py:200-234).
            k5 = 5
            land = torch.empty((k5 * n, L0), device=dev, dtype=torch.float32)
            five = []
            for i in range(k5):
                bag, sx_, lb_, st_ = make_slide(100 + i, n, dev, prepared=False)
                land[i * n:(i + 1) * n].copy_(bag)
                five.append((land[i * n:(i + 1) * n], sx_, lb_, st_))
                del bag
            for i in range(2):
                dp.step(five, k5)
            sync(); t3 = time.perf_counter()
            k5_steps = max(4, args.steps // 4)
            for i in range(k5_steps):
                dp.step(five, k5)
            sync(); b_ms = (time.perf_counter() - t3) / k5_steps * 1e3
            del five, land
            out["batched"] = {"value": round(k5 * 1e3 / b_ms, 3), "unit": "slides/s", "slides_per_step": k5, "ms_per_step": round(b_ms, 3), "steps": k5_steps,
                              "what": "five 100k-patch slides per optimiser step through ONE ragged multi-slide call of 500k rows (SlideShardedDP's own batching; bags landed "
                                      "back to back as BagPrefetcher(arena_rows=...) lands them); not `value`, which keeps one optimiser step per slide"}
        if world == 1 and args.config == 0 and len(slides[0]) == 1 and not prepared and args.bag_dtype == "fp32" and not args.no_ingest:
            out["ingest"] = time_ingest(dp, n, dev, args.ingest_slides)
        if world == 1 and args.config == 3 and len(slides[0]) > 1 and not args.no_per_slide:
            # the reference's own training semantics on this bag size: ONE optimiser step per slide (utils/core_utils_mtl_concat.py:200-234; its loaders
            # are batch_size = 1, utils/utils.py:51-55). `value` above is the data-parallel trainer's mode (a rank's slides share one ragged call).
            one = [[make_slide(300 + b, n, dev, prepared=False)] for b in range(2)]
            for i in range(10):
                dp.step(one[i % 2], 1)
            k1 = max(10 * args.steps, 200)
            sync(); t4 = time.perf_counter()
            for i in range(k1):
                dp.step(one[i % 2], 1)
            sync(); ps_ms = (time.perf_counter() - t4) / k1 * 1e3
            ops.enable_timing(True, level=2, prealloc=18 * 10)
            for i in range(10):
                dp.step(one[i % 2], 1)
            tg = ops.collect_timing()
            ops.enable_timing(False)
            del one
            out["per_slide"] = {"value": round(1e3 / ps_ms, 1), "unit": "slides/s", "ms_per_step": round(ps_ms, 4), "steps": k1, "slides_per_step": 1,
                                "roofline_mfma_frac": gemm_roofline(tg, 10 * n)["frac"],
                                "op_us_per_slide": {k_: round(v[1] / 10 * 1e3, 1) for k_, v in tg.items()},
                                "what": "one optimiser step per 10,000-patch slide (toad_mil_step_f32 + toad_adam_step_f32): the reference's train_loop "
                                        "semantics; no event packets inside the timed loop"}
        if world == 1 and args.config in (0, 3) and not args.no_dropin:
            k = max(args.steps, 10)
            ms = time_dropin(n, k, 3, dev, host_reads=False)
            ms_h = time_dropin(n, k, 3, dev, host_reads=True)
            out["dropin"] = {"ms_per_step": round(ms, 4), "value": round(1e3 / ms, 2), "unit": "slides/s", "steps": k,
                             "ms_per_step_with_host_reads": round(ms_h, 4), "vs_fused_step": round(ms / (ms_per_step / len(slides[0])), 3),
                             "what": "model(data, sex) + nn.CrossEntropyLoss x2 + loss.backward() + torch.optim.Adam(model.parameters()).step() + "
                                     "zero_grad() (utils/core_utils_mtl_concat.py:206-234); host reads = the loop's .item() / int() per slide"}
        if world == 1 and not args.no_cpu_baseline:
            # the CPU port on ONE slide of this configuration's size (config 4: one 50,000-patch slide; the 64-slide step is 64 of them)
            out["cpu_baseline"] = cpu_baseline_step(n, budget_s=12.0 if n >= 50_000 else 15.0, min_reps=3 if n >= 50_000 else 10)      # ~10-15 s of host work: the lease is GPU time
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if dist.is_initialized():                                 # tear the communicator down first: the JSON line stays the LAST line on stdout
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
