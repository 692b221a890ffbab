cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k6ms -o k -- python $GRAFT_REPO_ROOT/tools/k6m_small_probe.py 2>&1 | grep "K6m backward"
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections
f=(glob.glob("gpurun_out/k6ms/**/*kernel_trace.csv",recursive=True))[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "point_grad" in r["Kernel_Name"]:
        name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        acc[name+" grid="+r["Grid_Size_X"]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in acc.items(): print("%-48s %5d launches  avg %.2f us  min %.2f us"%(k, len(v), sum(v)/len(v), min(v)))
PY
find gpurun_out/k6ms -name "*.csv" -size +1M -delete
