cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rm -rf /tmp/trb; rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/trb -o t -- python bench.py --workload uplink --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > /tmp/trb.out 2>/tmp/trb.err
tail -1 /tmp/trb.out | cut -c1-200
f=$(find /tmp/trb -name "*hip_api_stats.csv" | head -1); head -12 $f | cut -d, -f1-5
f=$(find /tmp/trb -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]: print(r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], r['TotalDurationNs'])
PY
f=$(find /tmp/trb -name "*memory_copy_stats.csv" | head -1); [ -n "$f" ] && head -6 $f | cut -d, -f1-5
