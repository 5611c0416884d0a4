#!/bin/bash
# which copy commands the per-call scan still issues inside its per-subframe loop (sizes and directions)
cd "$(dirname "$0")/../shim/_build"; export TMPDIR=/tmp
./capture_gen /tmp/cap_100.bin 100 77 12 > /dev/null 2>&1
rm -rf /tmp/tr; rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d /tmp/tr -o t -- ./scan_gpu /tmp/cap_100.bin 30.72 > /dev/null 2>/tmp/tr.err
python3 - <<'PY'
import csv, glob, collections
def rows(pat):
    f = glob.glob('/tmp/tr/**/' + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k = rows('*kernel_trace.csv'); c = rows('*memory_copy_trace.csv')
t_pbch = max([int(r['End_Timestamp']) for r in k if 'k_pbch_decode' in r['Kernel_Name']] or [0])
print(c[0].keys() if c else 'no copies')
agg = collections.Counter()
ev = sorted([(int(r['Start_Timestamp']), 'K ' + r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:40]) for r in k if int(r['Start_Timestamp']) > t_pbch] +
            [(int(r['Start_Timestamp']), 'C %s %s' % (r.get('Direction', r.get('Kind', '?')), r.get('Size', r.get('Bytes', '?')))) for r in c if int(r['Start_Timestamp']) > t_pbch])
for t, e in ev[:40]: print(e)
PY
