cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4_ff3
rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r4_ff3 -- python3 tools/first_frame_one.py cover.json 1920 1080 5 > gpurun_out/r4_ff3.log 2>&1
ls gpurun_out/r4_ff3/*/
python3 - <<'PY'
import csv,glob
k=glob.glob('gpurun_out/r4_ff3/*/*kernel_trace.csv')[0]
a=glob.glob('gpurun_out/r4_ff3/*/*hip_api_trace.csv')[0]
ks=list(csv.DictReader(open(k)))
est=[r for r in ks if r['Kernel_Name'].startswith('rtc_estimate')][0]
t0=int(est['Start_Timestamp'])
ev=[]
for r in ks:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if -400e3 < s-t0 < 1100e3: ev.append((s,'GPU  %-40s dur %7.1f us'%(r['Kernel_Name'][:40],(e-s)/1e3)))
for r in csv.DictReader(open(a)):
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if -400e3 < s-t0 < 1100e3: ev.append((s,'HOST %-40s dur %7.1f us'%(r['Function'][:40],(e-s)/1e3)))
for s,t in sorted(ev): print('%9.1f %s'%((s-t0)/1e3,t))
PY
