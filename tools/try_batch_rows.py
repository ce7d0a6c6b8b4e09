"""Experiment driver: bench.py with SlideShardedDP.BATCH_ROWS (and, with ROWS:MAXPATCHES, BATCH_MAX_PATCHES) overridden in this process
(python tools/try_batch_rows.py ROWS[:MAXPATCHES] [bench args ...])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows, _, maxp = sys.argv[1].partition(":")
rows = int(rows)
sys.argv = ["bench.py"] + sys.argv[2:]
from toad_amd.dp import SlideShardedDP
SlideShardedDP.BATCH_ROWS = rows
if maxp:
    SlideShardedDP.BATCH_MAX_PATCHES = int(maxp)
import bench
bench.main()
