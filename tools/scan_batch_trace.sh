#!/bin/bash
# where the 20 MHz batch scanner's PDSCH stage spends its time: HIP API calls and kernels of the LAST Scanner::run (the all-subframes batch), in order
cd "$(dirname "$0")/../shim/_build"; export TMPDIR=/tmp
./capture_gen /tmp/cap_100.bin 100 77 12 > /dev/null 2>&1
for i in 1 2 3; do ./scan_batch /tmp/cap_100.bin 30.72 2>&1 >/dev/null | grep timing; done
rm -rf /tmp/tr; rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/tr -o t -- ./scan_batch /tmp/cap_100.bin 30.72 > /dev/null 2>/tmp/tr.err
grep timing /tmp/tr.err
python3 - <<'PY'
import csv, glob, collections
def rows(pat):
    f = glob.glob('/tmp/tr/**/' + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k = rows('*kernel_trace.csv'); a = rows('*hip_api_trace.csv')
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
fe = [r for r in k if 'pdcch' in r['Kernel_Name']]
t0 = int(fe[-1]['Start_Timestamp']) - 1000000  # 1 ms before the last batch's PDCCH kernels
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + name(r)[:60]) for r in k if int(r['Start_Timestamp']) >= t0]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'A ' + r['Function']) for r in a if int(r['End_Timestamp']) >= t0 - 200000]
ev.sort()
tend = max(e[1] for e in ev if e[2].startswith('K'))
print("timeline of the last batch (us from the last front-end launch; A = HIP API on the host, K = kernel on the device):")
for s, e, n in ev:
    if s > tend + 3000000: break
    if (e - s) >= 15000 or n.startswith('K'): print(f"  {(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  {n}")
ag = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    if s <= tend + 3000000: ag[n][0] += 1; ag[n][1] += e - s
print("every HIP API call of the process that took 300 us or more, with the kernel that ran before it:")
allk = sorted((int(r['Start_Timestamp']), name(r)[:50]) for r in k)
tz = min(int(r['Start_Timestamp']) for r in a)
import bisect
for r in a:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if d >= 300000:
        i = bisect.bisect(allk, (int(r['Start_Timestamp']), '')) - 1
        print(f"  {(int(r['Start_Timestamp']) - tz) / 1e3:10.1f}  {d / 1e3:9.1f} us  {r['Function']:28s} after {allk[i][1] if i >= 0 else '-'}")
print("totals:")
for n, v in sorted(ag.items(), key=lambda x: -x[1][1])[:25]: print(f"   {n:70s} x{v[0]:4d}  {v[1] / 1e3:9.1f} us")
PY
