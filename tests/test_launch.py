"""CPU: the built-in launcher (toad_amd/launch.py). `python bench.py --gpus N` started plainly must spawn its own N ranks through
torch.distributed.run on 127.0.0.1 (what the driver does by hand for N > 1) and fail only for lack of devices, never for lack of
a launcher."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_command_line():
    from toad_amd import launch
    cmd = launch.self_launch_cmd("/x/bench.py", ["--gpus", "8", "--steps", "3"], 8, port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "3"]
    launch.maybe_self_launch("/x/bench.py", [], 1)                      # one GPU: nothing to launch


def test_plain_start_with_more_gpus_than_devices_names_the_devices(monkeypatch):
    from toad_amd import launch
    monkeypatch.delenv("WORLD_SIZE", raising=False); monkeypatch.delenv("RANK", raising=False)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    with pytest.raises(SystemExit) as e:
        launch.maybe_self_launch("/x/bench.py", ["--gpus", "2"], 2)
    assert "visible HIP devices" in str(e.value) and "torch.distributed.run" in str(e.value)


@pytest.mark.timeout(300)
def test_plain_start_spawns_the_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_launch_probe.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "LAUNCH_PROBE world=2 sum=3" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_bench_stdout_carries_only_the_json_line():
    """bench.py's contract is ONE json line on stdout. Libraries print through C stdio (RCCL's version banner when a communicator is
    created): claim_stdout() points fd 1 at stderr for the rest of the run and emit() writes to the saved descriptor."""
    import json
    import subprocess
    code = ("import sys, os; sys.path.insert(0, %r); import bench; bench.claim_stdout(); print('python noise'); "
            "os.write(1, b'fd-level noise\\n'); import ctypes; ctypes.CDLL(None).puts(b'C stdio noise'); bench.emit({'value': 1.5})" % REPO)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("\n") == 1 and json.loads(r.stdout) == {"value": 1.5}
    for noise in ("python noise", "fd-level noise", "C stdio noise"):
        assert noise in r.stderr
