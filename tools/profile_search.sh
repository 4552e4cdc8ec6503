#!/bin/bash
# rocprofv3 passes of the on-device IVF-PQ search (tools/ivfpq_bench.py: the reference's index shape over 103 M keys, 8192 queries):
#   tools/profile_search.sh <tag>   -> gpurun_out/prof_<tag>_search/{stats,...} + gpurun_out/prof_<tag>_search.txt (summary)
# kernel-trace + stats in one pass; SQ counters and FETCH_SIZE / WRITE_SIZE in passes of their own (no trace domains mixed in).
set -eu
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
out=$R/gpurun_out/prof_${tag}_search
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/tools/ivfpq_bench.py > $out/bench.log 2>&1 || true
{
  echo "== tools/ivfpq_bench.py under rocprofv3 --kernel-trace --stats"; grep -E "scan=|IVF-PQ search|launches" $out/bench.log || true
  echo; echo "== one search, per kernel (tools/ivf_trace.sh)"; bash $R/tools/ivf_trace.sh 2>&1 | head -16
  echo; echo "== SQ counters (tools/pmc_ivf8.sh: quad-cycle units for SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*, cycles for SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_*)"
  bash $R/tools/pmc_ivf8.sh 2>&1 | tail -8
  echo; echo "== HBM traffic, KB (tools/pmc_ivf8_mem.sh; FETCH_SIZE counts 128-B requests as 64 B on gfx950: double it)"
  bash $R/tools/pmc_ivf8_mem.sh 2>&1 | tail -12
} > $R/gpurun_out/prof_${tag}_search.txt 2>&1
f=$(ls $out/stats/*/*_kernel_stats.csv | head -1); cp $f $R/gpurun_out/prof_${tag}_search_kernel_stats.csv
cat $R/gpurun_out/prof_${tag}_search.txt | head -70
