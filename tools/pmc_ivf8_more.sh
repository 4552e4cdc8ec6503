# more SQ counters of the search filter (tools/ivfpq_bench.py): instruction mix, instruction fetch, LDS queues.  Run on the GPU box.
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_more$i
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_more$i --output-format csv -- python3 $R/tools/ivfpq_bench.py > /dev/null 2>&1 || true
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(f'{R}/gpurun_out/pmc_more*')):
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if 'ivfpq_scan8' in r['Kernel_Name']:
                acc[(r['Kernel_Name'].split('::')[-1][:26], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
        for disp in sorted(acc, key=lambda k: int(k[1]))[-2:]:
            print(os.path.basename(d), disp[0], {k: f'{v:.4g}' for k, v in acc[disp].items()})
PY
