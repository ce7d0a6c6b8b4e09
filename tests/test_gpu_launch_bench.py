"""GPU: bench.py started the way the driver starts it for N > 1 - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` (toad_amd/launch.py builds that command line) - with backend "nccl" (RCCL)
on the one GPU a test box has. A process group then exists at world 1 and the step issues its gradient all-reduce through RCCL exactly
as it does at N = 8 (SlideShardedDP(always_reduce=True)); BASELINE config 4's strong-scaling line must come back as ONE JSON line.
What this cannot show is the xGMI transfer between devices; everything else of `bench.py --config 4 --gpus 8` runs here.
Reference being replaced: the intra-bag nn.DataParallel of models/model_toad.py:79-81 (not reproduced)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_config4_through_the_torchrun_path_with_rccl(cuda):
    from toad_amd import launch
    cmd = launch.self_launch_cmd(os.path.join(REPO, "bench.py"), ["--config", "4", "--gpus", "1", "--steps", "2", "--warmup", "1",
                                                                   "--no-cpu-baseline", "--backend", "nccl"], 1)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                       # the JSON line is the only thing on stdout (RCCL's banner goes to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["steps"] == 2
    assert out["config"]["slides_per_step"] == 64 and out["config"]["patches_per_slide"] == 50000
    assert "torch.distributed.run" in out["config"]["parallelism"]
    ar = out["allreduce"]
    assert ar["world"] == 1 and ar["backend"].startswith("nccl") and "error" not in ar and ar["us"] > 0 and ar["unchanged_at_world_1"]
    assert "communicator" not in ar                                # the group torchrun's rendezvous created was used, not a private one
    assert out["value"] > 0 and abs(out["value"] - 64 * 1e3 / out["ms_per_step"]) <= 1e-2 * out["value"]
    for key in ("roofline", "roofline_mfma"):
        assert 0 < out[key]["frac"] < 1, key
    assert out["roofline_mfma"]["rows"] == 2 * 64 * 50000


@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra,scaling,slides", [([], "weak", 2), (["--config", "4"], "strong", 64)])
def test_world_two_plumbing_on_one_gpu_over_gloo(cuda, extra, scaling, slides):
    """Everything of `bench.py --gpus N` that does not need a second device, at N = 2: self-launch of two ranks (toad_amd/launch.py), rendezvous on
    127.0.0.1, slide sharding, the gradient all-reduce inside every step, barrier + max-over-ranks timing, the per-rank attribution keys and ONE
    JSON line from rank 0. RCCL refuses two ranks on one GPU, so the collective runs over gloo here (`--single-device --backend gloo`: a plumbing
    switch, never a measurement); the RCCL call itself is covered at world 1 above."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-dropin", "--sustain-seconds", "0"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 2 and out["config"]["slides_per_step"] == slides
    assert out["allreduce"]["world"] == 2 and out["allreduce"]["us"] > 0
    pr = out["per_rank"]
    assert len(pr["step_ms"]) == 2 and len(pr["allreduce_us"]) == 2 and len(pr["patches_per_step"]) == 2
    assert pr["patches_per_step"][0] == pr["patches_per_step"][1]              # both configurations shard evenly over two ranks
    assert max(pr["step_ms"]) <= out["ms_per_step"] * 1.05                      # the line's time is the max over ranks (barrier included)
    assert 0 < pr["scaling_efficiency_vs_per_rank_min"] <= 1.0001               # fastest rank's own step time / the job's step time
    assert out["value"] > 0 and abs(out["value"] - slides * 1e3 / out["ms_per_step"]) <= 1e-2 * out["value"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra,scaling,slides", [([], "weak", 8), (["--config", "4"], "strong", 64), (["--config", "4", "--ragged"], "strong", 64)],
                         ids=["headline", "config4", "config4_ragged"])
def test_world_eight_plumbing_on_one_gpu_over_gloo(cuda, extra, scaling, slides):
    """The driver's 8-GPU launch shape on the one device a test box has: `bench.py --gpus 8` spawns eight ranks (toad_amd/launch.py), every rank
    builds its shard (config 4: 8 of the 64 slides = 400,000 rows, ONE ragged multi-slide call at the default batch_rows of 524,288), steps with the gradient all-reduce inside, and rank 0
    prints ONE JSON line with the per-rank attribution. `--single-device --backend gloo` is a plumbing switch (RCCL refuses eight ranks on one
    GPU), never a measurement: the value is asserted to be self-consistent, not fast. `--ragged` deals 64 slides of log-normal length with
    dp.shard_by_length: the patch counts per rank must balance to a few percent and cover all 3.2 M patches exactly once."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--single-device", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-dropin", "--sustain-seconds", "0"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=860)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == scaling and out["steps"] == 2 and out["config"]["slides_per_step"] == slides
    assert out["allreduce"]["world"] == 8 and out["allreduce"]["us"] > 0
    pr = out["per_rank"]
    assert len(pr["step_ms"]) == len(pr["allreduce_us"]) == len(pr["patches_per_step"]) == 8
    if "--ragged" in extra:
        assert sum(pr["patches_per_step"]) == pytest.approx(64 * 50000, rel=2e-3) and "LOG-NORMAL" in out["metric"]
        assert max(pr["patches_per_step"]) <= 1.10 * min(pr["patches_per_step"]), pr["patches_per_step"]      # longest-processing-time greedy balances 64 slides over 8 ranks
    else:
        assert len(set(pr["patches_per_step"])) == 1 and pr["patches_per_step"][0] == (100000 if not extra else 8 * 50000)
    assert max(pr["step_ms"]) <= out["ms_per_step"] * 1.05
    assert 0 < pr["scaling_efficiency_vs_per_rank_min"] <= 1.0001
    assert out["value"] > 0 and abs(out["value"] - slides * 1e3 / out["ms_per_step"]) <= 1e-2 * out["value"]
    for key in ("roofline", "roofline_mfma"):
        assert 0 < out[key]["frac"] < 1, key
