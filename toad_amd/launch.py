"""One process per GPU: self-launch and process-group helpers for the bench / train drivers.

``python bench.py --gpus N`` started plainly (no WORLD_SIZE in the environment) re-executes itself under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`` - the command the
driver uses for N > 1 - so the slide-sharded data-parallel path (toad_amd/dp.py; the reference's only multi-GPU code is the
intra-bag nn.DataParallel of models/model_toad.py:79-81, deliberately not reproduced) needs no launcher knowledge from the caller.
"""
from __future__ import annotations

import datetime
import os
import socket
import subprocess
import sys
from typing import List, Optional


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def self_launch_cmd(script: str, argv: List[str], n_procs: int, port: Optional[int] = None) -> List[str]:
    """The torch.distributed.run command line that starts ``script argv`` as ``n_procs`` ranks on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_procs}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(script)] + list(argv)


def maybe_self_launch(script: str, argv: List[str], n_gpus: int, single_device: bool = False) -> None:
    """If ``n_gpus`` > 1 and this process was NOT started by torch.distributed.run, check the device count, spawn the ranks and
    exit with their return code. Returns (does nothing) for n_gpus == 1 or when already running as a rank."""
    if n_gpus <= 1 or launched_by_torchrun():
        return
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not single_device and have < n_gpus:
        raise SystemExit(f"{os.path.basename(script)}: --gpus {n_gpus} needs {n_gpus} visible HIP devices, found {have} "
                         f"(one process per GPU; the launcher itself is built in: this would have run "
                         f"`{' '.join(self_launch_cmd(script, argv, n_gpus, port=29500)[1:9])} ...`)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL / tensor sharing across processes)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    raise SystemExit(subprocess.call(self_launch_cmd(script, argv, n_gpus), env=env))


def init_process_group(backend: str, device=None, timeout_s: int = 120) -> None:
    """torch.distributed over RCCL (backend "nccl" IS RCCL on ROCm) or gloo. With no rendezvous variables in the environment a
    world of ONE is created on a free local port: the same collective code path (RCCL communicator, all-reduce kernels on the
    launch stream) then runs on a single GPU."""
    import torch.distributed as dist
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(free_port())
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
