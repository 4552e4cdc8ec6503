# Calibration of FETCH_SIZE for sub-line accesses (tools/probes/fetch_calib.hip): reported KB per launch against the bytes requested.
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 $R/tools/probes/fetch_calib.hip -o /tmp/fetch_calib
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_calib $R/gpurun_out/pmc_calib_t
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_calib --output-format csv -- /tmp/fetch_calib
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc_calib_t --output-format csv -- /tmp/fetch_calib > /dev/null
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
req={'stream_kernel':6400<<20,'rows_kernel<4, 64>':(1<<25)*64,'dword_kernel':(1<<25)*4,'rows_kernel<8, 128>':(1<<25)*128}
acc=collections.defaultdict(list)
for f in glob.glob(f'{R}/gpurun_out/pmc_calib/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE': acc[r['Kernel_Name'].split('(')[0].replace('void ','')].append(float(r['Counter_Value'])*1024)
dur=collections.defaultdict(list)
for f in glob.glob(f'{R}/gpurun_out/pmc_calib_t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): dur[r['Kernel_Name'].split('(')[0].replace('void ','')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
def want(k):
    if 'stream' in k: return req['stream_kernel']
    if 'dword' in k: return req['dword_kernel']
    return req['rows_kernel<8, 128>'] if '128' in k else req['rows_kernel<4, 64>']       # (the cooperative variants fetch the same rows)
for k,v in acc.items():
    b=want(k); rep=v[-1]; us=dur[k][-1] if dur.get(k) else float('nan')
    print(f"{k:22s} requested {b/1e9:7.3f} GB   FETCH_SIZE reports {rep/1e9:7.3f} GB   reported/requested {rep/b:5.2f}   {us:8.1f} us = {b/us/1e3:7.1f} GB/s requested, {b/64/us/1e3 if 'rows' in k or 'dword' in k else float('nan'):6.1f}")
PY
