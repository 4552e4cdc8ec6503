# effective clock and matrix-pipe occupancy of the star kernel (tools/star_bench.py, T = 8192)
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_star_clk
T=8192 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $R/gpurun_out/pmc_star_clk --output-format csv -- python3 $R/tools/star_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(f'{R}/gpurun_out/pmc_star_clk/**/*counter_collection.csv', recursive=True)[0]
t=glob.glob(f'{R}/gpurun_out/pmc_star_clk/**/*kernel_trace.csv', recursive=True)[0]
dur={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(t))}
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if 'star_attn_tab' in r['Kernel_Name']:
        acc[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
for d in sorted(acc,key=int)[-2:]:
    c=acc[d]; ns=dur[d]; cyc=c['GRBM_GUI_ACTIVE']/8
    print(f"{ns/1e3:.1f} us  clock {cyc/ns:.3f} GHz  MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc:.3f}", {k:f'{v:.3g}' for k,v in c.items()})
PY
