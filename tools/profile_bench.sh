#!/bin/bash
# rocprofv3 passes of the default bench.py run (on the GPU box, through gpurun):
#   tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write}/...
# kernel-trace + stats in one pass, FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: TCC slots).
set -e
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --no-cpu-baseline --no-extras --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$tag.bench.json 2> $R/gpurun_out/prof_$tag.err || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/fetch -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/write -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || true
ls -R $R/gpurun_out/prof_$tag | head -30
