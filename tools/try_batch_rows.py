"""Experiment driver: bench.py with SlideShardedDP.BATCH_ROWS overridden in this process (python tools/try_batch_rows.py ROWS [bench args ...])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
from toad_amd.dp import SlideShardedDP
SlideShardedDP.BATCH_ROWS = rows
import bench
bench.main()
