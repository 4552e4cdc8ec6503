#!/bin/bash
# rocprofv3 passes of the default bench.py run (on the GPU box, through gpurun):
#   tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write}/...  and  gpurun_out/prof_<tag>_L3/{stats,fetch,write}/...
# kernel-trace + stats in one pass, FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: TCC slots).
# The _L3 set is the shipped 3-layer recipe as a step of its own (bench.py --layers 3 --blocks 16: the shapes of recipe_L3.uniform_ids).
set -e
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --no-cpu-baseline --no-extras --no-parity"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$tag.bench.json 2> $R/gpurun_out/prof_$tag.err || true
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/fetch -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || true
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/write -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || true
if [ "${2:-L3}" = "L3" ]; then
export GNNLM_STATE_CACHE_GIB=0   # (the i.i.d. ids of the bench could only miss: profile the kernels, not the cache turnover)
ARGS3="--layers 3 --blocks 16 --pool 2 --steps 5 --warmup 3 --settle-s 0.1 --no-cpu-baseline --no-extras --no-parity"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_L3/stats -- python3 $R/bench.py $ARGS3 > $R/gpurun_out/prof_${tag}_L3.bench.json 2> $R/gpurun_out/prof_${tag}_L3.err || true
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_L3/fetch -- python3 $R/bench.py $ARGS3 > /dev/null 2>&1 || true
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_L3/write -- python3 $R/bench.py $ARGS3 > /dev/null 2>&1 || true
fi
ls -R $R/gpurun_out/prof_$tag | head -30
