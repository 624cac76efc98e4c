# Diagnostic: grid size, registers, LDS and scratch of every render-kernel launch of a command (rocprofv3 kernel trace)
#   bash tools/kernel_shapes.sh python3 tools/all_time.py cover
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/shapes
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<P
import csv, glob
f = glob.glob("gpurun_out/shapes/**/*kernel_trace.csv", recursive=True)[0]
seen = {}
for r in csv.DictReader(open(f)):
    if "render" in r["Kernel_Name"]:
        key = tuple(r.get(k) for k in ("Kernel_Name", "Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size"))
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n, s = seen.get(key, (0, 0))
        seen[key] = (n + 1, s + t)
for k, (n, s) in seen.items():
    print(k, "launches", n, "avg us", s / n / 1e3)
P
