#!/usr/bin/env python3
"""tools/kernel_resources.py <file.hip> [more hipcc flags]: registers, scratch, occupancy and static LDS of every kernel in the file
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950), one line per kernel with the name demangled."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+))", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    if m.group(1):
        cur = {"name": m.group(1)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(2).strip()] = int(m.group(3))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)[5:] if n.startswith("void ") else re.sub(r"\(.*", "", n)
    print("%-70s SGPR %3d  VGPR %3d  AGPR %3d  scratch %4d  occupancy %d  LDS %6d" % (n[:70], r.get("TotalSGPRs", -1), r.get("VGPRs", -1), r.get("AGPRs", -1),
          r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))
