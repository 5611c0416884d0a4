#!/bin/bash
# tools/isa_listing.sh <file-stem> <mangled-kernel-substring> <out.txt>: the gfx950 ISA of one kernel as hipcc compiles it for the library
# (same flags as openlte_amd/csrc/Makefile), with an instruction count per basic block (VALU / SALU / VMEM / LDS) in front of the listing.
# This is what the instruction-floor arguments in DESIGN.md are counted from; needs no GPU.
set -e
stem=$1; pat=$2; out=$3
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d); cd $tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$root/openlte_amd/csrc --save-temps -c $root/openlte_amd/csrc/$stem.hip -o x.o >/dev/null 2>&1
python3 - "$stem-hip-amdgcn-amd-amdhsa-gfx950.s" "$pat" > "$root/$out" <<'PY'
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0] and ':' in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
body = lines[start:end]
blocks, cur = [], ("entry", [])
for l in body[1:]:
    ls = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', ls):
        blocks.append(cur); cur = (ls.split(':')[0] + (' (' + ls.split(';', 1)[1].strip() + ')' if ';' in ls else ''), [])
    elif ls and not ls.startswith(';') and not ls.startswith('.'):
        cur[1].append(ls.split()[0])
blocks.append(cur)
print("kernel %s  (%s, hipcc --offload-arch=gfx950 -O3 -ffp-contract=off)" % (body[0].split(':')[0], sys.argv[1]))
print("instructions per basic block:")
for name, ins in blocks:
    c = collections.Counter('VALU' if i.startswith('v_') else 'SALU' if i.startswith('s_') else 'LDS' if i.startswith('ds_') else 'VMEM' if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other' for i in ins)
    print("  %-70s %5d  %s" % (name, len(ins), ' '.join('%s %d' % kv for kv in sorted(c.items()))))
print()
print('\n'.join(body))
PY
rm -rf $tmp
