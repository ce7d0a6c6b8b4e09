#!/bin/bash
# Local wrapper: rebuild libtoad_hip.so if stale (the .so travels with the snapshot), then hand the command to gpurun.
#   tools/gpu.sh TIMEOUT 'command ...'
set -e
cd "$(dirname "$0")/.."
python -m toad_amd.build > /tmp/toad_build.log 2>&1 || { tail -30 /tmp/toad_build.log; exit 1; }
python -c "from toad_amd import _lib; _lib.load()" || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
