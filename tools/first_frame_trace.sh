cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sc in "cover.json 1920 1080 5" "dragons.json 3840 2160 5" "teapot.json 1920 1080 5"; do
  set -- $sc
  rm -rf gpurun_out/r4_ff
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_ff -- python3 tools/first_frame_one.py $1 $2 $3 $4 > gpurun_out/r4_ff.log 2>&1
  f=$(find gpurun_out/r4_ff -name "*kernel_trace.csv" | head -1)
  echo "== $1"
  python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
rows=[r for r in rows if r['Kernel_Name'].startswith('rtc_')]
t0=int(rows[0]['Start_Timestamp'])
for r in rows: print('%-34s start %9.1f us  dur %8.1f us' % (r['Kernel_Name'][:34], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
"
done
