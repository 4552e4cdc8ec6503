set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ivf_trace
NQ=2048 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ivf_trace -- python3 $R/tools/ivfpq_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,re
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(f'{R}/gpurun_out/ivf_trace/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
tail=rows[-48:]
for r in tail:
    n=r['Kernel_Name']; m=re.search(r'gnnlm::\(anonymous namespace\)::(\w+(?:<[^>(]*>)?)', n)
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if m or d>20: print(f"{d:9.1f} us  {m.group(1) if m else n[:70]}")
PY
