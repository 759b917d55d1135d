#!/bin/bash
# Timeline of one steady-state launch group of another shape (tools/trace_group.py):  bash tools/shape_timeline.sh <tag> [bench.py args...]
#   -> gpurun_out/timeline_<tag>.md
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
timeout -k 5 600 rocprofv3 --kernel-trace --marker-trace --output-format csv -d /tmp/tl_$TAG -- python3 $R/bench.py --no-boundary --no-overlap-leg --cpu-seconds 0 --no-verify --steps 8 --warmup 2 --min-seconds 0.3 "$@" > /tmp/tl_$TAG.json 2> /tmp/tl_$TAG.err < /dev/null
python3 $R/tools/trace_group.py /tmp/tl_$TAG $R/gpurun_out/timeline_$TAG.md > /dev/null || tail -5 /tmp/tl_$TAG.err
cat $R/gpurun_out/timeline_$TAG.md
rm -rf /tmp/tl_$TAG
