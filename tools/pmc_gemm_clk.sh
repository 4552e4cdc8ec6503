# effective shader clock and MFMA-pipe occupancy of the GEMM kernels (gemm_bench shapes): GRBM_GUI_ACTIVE / duration
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_gemm_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_gemm_clk --output-format csv -- python3 $R/tools/gemm_bench.py "${1:-sq 4096}" > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(f'{R}/gpurun_out/pmc_gemm_clk/**/*counter_collection.csv', recursive=True)[0]
t=glob.glob(f'{R}/gpurun_out/pmc_gemm_clk/**/*kernel_trace.csv', recursive=True)[0]
dur={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(t))}
acc=collections.defaultdict(lambda: collections.defaultdict(float)); name={}
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']:
        acc[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value']); name[r['Dispatch_Id']]=r['Kernel_Name'][30:90]
for d in sorted(acc,key=int)[-3:]:
    c=acc[d]; ns=dur[d]
    print(name[d], f"{ns/1e3:.1f} us  GUI_ACTIVE/8/ns = {c['GRBM_GUI_ACTIVE']/8/ns:.3f} GHz   MFMA_BUSY/(1024 SIMD)/cycles = {c['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(c['GRBM_GUI_ACTIVE']/8):.3f}", {k:f'{v:.3g}' for k,v in c.items()})
PY
