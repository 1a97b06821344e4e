#!/bin/bash
# build in-tree (hipcc cross-compiles here), then run the given command on a GPU box:  tools/gpu.sh <timeout_s> '<command>'
set -e
cd /root/repo
python -c "import __graft_entry__ as g; g.build()" | grep -v "^/opt/rocm/bin/hipcc" || true
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
