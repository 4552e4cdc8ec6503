set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ivf_trace
NQ=${NQ:-8192} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ivf_trace -- python3 $R/tools/ivfpq_bench.py > $R/gpurun_out/ivf_trace.log 2>&1
python3 - <<'PY'
# per-kernel time of ONE search: everything after the index images were built (last pack_* dispatch), halved (warm-up + timed call)
import collections,csv,glob,os,re
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(f'{R}/gpurun_out/ivf_trace/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
last=max(i for i,r in enumerate(rows) if 'pack_tiles' in r['Kernel_Name'] or 'pack_codes' in r['Kernel_Name'])
acc=collections.defaultdict(lambda:[0,0.0])
for r in rows[last+1:]:
    n=r['Kernel_Name']; m=re.search(r'gnnlm::\(anonymous namespace\)::(\w+(?:<[^>(]*>)?)', n)
    key=m.group(1) if m else 'torch: '+n[:60]
    acc[key][0]+=1; acc[key][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
tot=sum(v[1] for v in acc.values())/2
print(f"one search: {tot/1e3:.2f} ms of kernel time")
for k,v in sorted(acc.items(), key=lambda kv:-kv[1][1])[:40]:
    print(f"{v[1]/2:10.1f} us {v[0]//2:5d} x  {k}")
PY
