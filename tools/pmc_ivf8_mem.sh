# HBM traffic of the search kernels (tools/ivfpq_bench.py): FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots)
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_ivf8f $R/gpurun_out/pmc_ivf8w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_ivf8f --output-format csv -- python3 $R/tools/ivfpq_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_ivf8w --output-format csv -- python3 $R/tools/ivfpq_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in ('pmc_ivf8f','pmc_ivf8w'):
    for f in glob.glob(f'{R}/gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if 'ivfpq' in n or 'topk' in n:
                acc[(n.split('::')[-1][:28], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
        seen=set()
        for disp in sorted(acc, key=lambda k: -int(k[1])):
            if disp[0] in seen: continue
            seen.add(disp[0])
            print(d, disp[0], {k: f'{v:.4g}' for k, v in acc[disp].items()})
PY
