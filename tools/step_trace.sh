# per-dispatch trace of the last bench step (on the GPU box): tools/step_trace.sh [bench args] -> stdout
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/step_trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/step_trace -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-parity "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, re
R = os.environ['GRAFT_REPO_ROOT']
f = glob.glob(f'{R}/gpurun_out/step_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last step = from the last gather/half_to_float ... take the last 60 dispatches
for r in rows[-70:]:
    n = r['Kernel_Name']
    m = re.search(r'gnnlm::\(anonymous namespace\)::(\w+(?:<[^>(]*>)?)', n)
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}  wg {r.get('Workgroup_Size_X', '?'):>5}  {m.group(1) if m else n[:60]}")
PY
