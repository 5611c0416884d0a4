#!/bin/bash
# where the per-call scan's time goes (1.4 MHz and 20 MHz captures): HIP API time by call, GPU time by kernel, per subframe of the loop
cd "$(dirname "$0")/../shim/_build"; export TMPDIR=/tmp
for cfg in "6 17 30 1.92" "100 77 12 30.72"; do
  set -- $cfg
  ./capture_gen /tmp/cap_$1.bin $1 $2 $3 > /dev/null 2>&1
  rm -rf /tmp/tr; rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/tr -o t -- ./scan_gpu /tmp/cap_$1.bin $4 > /dev/null 2>/tmp/tr.err
  grep timing /tmp/tr.err | sed "s/.*per-subframe loop/$1 RB (traced): loop/"
  python3 - $1 <<'PY'
import csv, glob, sys, collections
nrb = sys.argv[1]
def rows(pat):
    f = glob.glob('/tmp/tr/**/' + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k = rows('*kernel_trace.csv'); a = rows('*hip_api_trace.csv')
# the loop = everything after the last k_pbch_decode (the scanner decodes the MIB, then walks the subframes)
t_pbch = max([int(r['End_Timestamp']) for r in k if 'k_pbch_decode' in r['Kernel_Name']] or [0])
kk = [r for r in k if int(r['Start_Timestamp']) > t_pbch]
n_sf = sum(1 for r in kk if 'k_pack_cols' in r['Kernel_Name'])
agg = collections.defaultdict(lambda: [0, 0])
for r in kk:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
    agg[n][0] += 1; agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
print(f"{nrb} RB: {n_sf} subframes in the loop, {len(kk)/max(n_sf,1):.1f} kernels and {tot/max(n_sf,1)/1e3:.1f} us of kernel time per subframe")
for n, v in sorted(agg.items(), key=lambda x: -x[1][1])[:12]: print(f"   {n:24s} {v[0]/max(n_sf,1):5.2f} per subframe  avg {v[1]/v[0]/1e3:6.1f} us")
aa = [r for r in a if int(r['Start_Timestamp']) > t_pbch]
ag = collections.defaultdict(lambda: [0, 0])
for r in aa: ag[r['Function']][0] += 1; ag[r['Function']][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, v in sorted(ag.items(), key=lambda x: -x[1][1])[:6]: print(f"   API {n:24s} {v[0]/max(n_sf,1):6.1f} per subframe  {v[1]/max(n_sf,1)/1e3:6.1f} us per subframe (traced)")
PY
done
