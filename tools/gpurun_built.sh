#!/bin/bash
# Build every native library in-tree (the GPU box only runs prebuilt files), then hand the
# command to gpurun.  usage: tools/gpurun_built.sh <timeout> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -v "warning: loop not unrolled\|^ *[0-9]* | \|^ *| \|warnings generated\|In file included" | tail -5
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
