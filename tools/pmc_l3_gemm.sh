set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_l3
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_l3 --output-format csv -- python3 $R/bench.py --layers 3 --blocks 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(f'{R}/gpurun_out/pmc_l3/**/*counter_collection.csv', recursive=True)[0]
t=glob.glob(f'{R}/gpurun_out/pmc_l3/**/*kernel_trace.csv', recursive=True)[0]
dur={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(t))}
acc=collections.defaultdict(lambda: collections.defaultdict(float)); name={}
for r in csv.DictReader(open(f)):
    if 'gemm_nt_f32_sched' in r['Kernel_Name'] or 'gemm_nt_f32_dma' in r['Kernel_Name']:
        acc[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value']); name[r['Dispatch_Id']]=r['Kernel_Name'][40:80]
big=[d for d in acc if dur[d]>2_000_000]
for d in sorted(big,key=int)[-9:]:
    c=acc[d]; ns=dur[d]; cyc=c['GRBM_GUI_ACTIVE']/8
    print(name[d], f"{ns/1e3:.0f} us  clock {cyc/ns:.3f} GHz  MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc:.3f}  wait_any/wave {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f}")
PY
