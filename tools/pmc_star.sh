set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
for e in 5 6 7 8; do echo EXP $e; GNNLM_LIB=$R/gnn-lm_amd/build/exp/libstab$e.so T=8192 python3 $R/tools/star_bench.py; done
T=8192 python3 $R/tools/star_bench.py
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_star1 --output-format csv -- python3 $R/tools/star_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_star2 --output-format csv -- python3 $R/tools/star_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in ('pmc_star1','pmc_star2'):
    for f in glob.glob(f'{R}/gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'star_attn_tab' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,sum(v)/len(v),len(v))
PY
