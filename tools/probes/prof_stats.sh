# rocprofv3 kernel stats of a short bench run: per-kernel average durations (us) of the families named in $1 (regex); $2 = URSO_LIB_VARIANT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ps
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- env URSO_LIB_VARIANT=${2:-} python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-steps 0 > /tmp/ps_bench.txt 2>&1
tail -1 /tmp/ps_bench.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])
except Exception as e: print('bench line unreadable', e)"
python - "$1" <<'PY'
import csv, glob, re, sys
f = glob.glob('/tmp/ps/**/*kernel_stats.csv', recursive=True)[0]
pat = re.compile(sys.argv[1])
for r in csv.DictReader(open(f)):
    if pat.search(r['Name']):
        print('%-72s calls %5s avg %8.1f us' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3))
PY
