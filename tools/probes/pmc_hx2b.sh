cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "4 32" "4 0" "4 22"; do
  set -- $cfg
  rm -rf /tmp/p1
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p1 -- python $R/tools/probes/hx2_one.py $1 $2 > /dev/null 2>&1
  echo "=== stage $1 shape $2"
  python - <<PY
import csv, glob
f = glob.glob('/tmp/p1/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'hconv' in r['Kernel_Name']]
print(rows[0].keys())
by = {}
for r in rows:
    by.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
    by[r['Dispatch_Id']]['dur'] = (float(r.get('End_Timestamp', 0)) - float(r.get('Start_Timestamp', 0)))
for d, v in by.items():
    print(d, v)
PY
  k=$(find /tmp/p1 -name '*kernel_trace.csv'); python - <<PY
import csv
for r in csv.DictReader(open("$k")):
    if 'hconv' in r['Kernel_Name']:
        print('trace dur us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
PY
done
