# PMC passes of the int8-MFMA IVF-PQ scan (tools/ivfpq_bench.py): run on the GPU box, prints per-dispatch sums of ivfpq_scan8_kernel
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
export NQ=${NQ:-8192}
rm -rf $R/gpurun_out/pmc_ivf8a $R/gpurun_out/pmc_ivf8b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_ivf8a --output-format csv -- python3 $R/tools/ivfpq_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $R/gpurun_out/pmc_ivf8b --output-format csv -- python3 $R/tools/ivfpq_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in ('pmc_ivf8a','pmc_ivf8b'):
    for f in glob.glob(f'{R}/gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if 'ivfpq_scan8' in r['Kernel_Name'] or 'ivfpq_rescore' in r['Kernel_Name']:
                acc[(r['Kernel_Name'].split('::')[-1][:24], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
        for disp in sorted(acc, key=lambda k: int(k[1]))[-2:]:
            print(d, disp, {k: f'{v:.4g}' for k, v in acc[disp].items()})
    for f in glob.glob(f'{R}/gpurun_out/{d}/**/*kernel_trace.csv', recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if 'ivfpq_scan8' in r['Kernel_Name'] or 'ivfpq_rescore' in r['Kernel_Name']]
        for r in rows[-2:]:
            print(d, r['Kernel_Name'].split('::')[-1][:24], 'us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'grid', r.get('Grid_Size'))
PY
