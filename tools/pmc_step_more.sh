# SQ counters of the step's kernels (bench.py, default shapes): instruction mix and pipe activity per kernel.  Run on the GPU box.
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --settle-s 0.05 --no-cpu-baseline --no-extras --no-parity"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_step$i
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_step$i --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || true
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
want=('star_attn','knn_','causal_attn','gemm_nt_f32_dma','layernorm')
for d in sorted(glob.glob(f'{R}/gpurun_out/pmc_step*')):
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        seen=set()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('::')[-1].split('(')[0][:40]
            if any(w in k for w in want):
                acc[k][r['Counter_Name']] += float(r['Counter_Value'])
                if (k, r['Dispatch_Id']) not in seen: seen.add((k, r['Dispatch_Id'])); n[k]+=1
        for k in acc:
            print(os.path.basename(d), k, n[k], {c: f'{v/n[k]:.4g}' for c, v in acc[k].items()})
    for f in glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True):
        os.remove(f)
PY
rm -rf $R/gpurun_out/pmc_step*/*/*_agent_info.csv
