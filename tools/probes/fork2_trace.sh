#!/bin/bash
# kernel traces of the step on the single chain and with the complementary fork: per-symbol average duration side by side
R=$(pwd); O=$R/gpurun_out/fork2_trace; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for v in 0 2; do
  URSO_WGRAD_STREAM=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/m$v" -- python "$R/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --pcie-steps 0 > "$O/bench$v.json" 2> "$O/err$v.txt"
done
cd "$R"
python - <<'PY'
import csv, glob, collections
def load(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in csv.DictReader(open(f))]
    rows.sort()
    st = [i for i, r in enumerate(rows) if "mold_kernel" in r[2]]
    return rows[st[-3]:st[-1]]          # two replays
import re
def short(n):
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", n.replace("_Z", ""))
    return n[:48]
a, b = load("gpurun_out/fork2_trace/m0"), load("gpurun_out/fork2_trace/m2")
def stats(rows):
    d = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, q, t in rows: d[n][0] += 1; d[n][1] += (e - s) / 1e3
    return d
sa, sb = stats(a), stats(b)
print("wall per step: chain %.1f us   forked %.1f us" % ((a[-1][1] - a[0][0]) / 2e3, (b[-1][1] - b[0][0]) / 2e3))
print("sum of kernel durations per step: chain %.1f us   forked %.1f us" % (sum(v[1] for v in sa.values()) / 2, sum(v[1] for v in sb.values()) / 2))
print("%-60s %5s %10s %10s" % ("kernel", "n", "chain us", "forked us"))
for n in sorted(sa, key=lambda n: -sa[n][1]):
    print("%-60s %5d %10.1f %10.1f" % (n[:60], sa[n][0] // 2, sa[n][1] / 2, sb.get(n, [0, 0])[1] / 2))
qs = collections.Counter((r[3], r[4]) for r in b)
print("queues / streams of the forked step:", dict(qs))
PY
