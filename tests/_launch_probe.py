"""Helper for tests/test_launch.py: started plainly with --gpus N it must re-execute itself as N ranks (toad_amd.launch), each rank
joins a gloo group on 127.0.0.1 and all-reduces its rank + 1; rank 0 prints the sum."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from toad_amd import launch

n = int(sys.argv[sys.argv.index("--gpus") + 1])
launch.maybe_self_launch(__file__, sys.argv[1:], n, single_device=True)      # single_device: no GPU count check (CPU test)
launch.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print(f"LAUNCH_PROBE world={dist.get_world_size()} sum={t.item():.0f}", flush=True)
dist.barrier()
dist.destroy_process_group()
