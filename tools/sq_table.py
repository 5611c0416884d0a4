#!/usr/bin/env python3
"""Per-kernel table from the output of tools/pmc_sq.sh (one or more passes concatenated): instructions per wave and the share of
the kernel's time the vector ALUs were issuing (a wave64 instruction occupies its SIMD for four cycles; 1024 SIMDs) and the LDS pipes
were busy (one per CU, 256 of them).  busy_Mcyc = SQ_BUSY_CYCLES / 32 shader engines."""
import ast, re, sys
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"\('(k_\w+)', '(\d+)'\) launches (\d+) (\{.*\})", l.strip())
    if m:
        rows.setdefault((m.group(1), int(m.group(2))), {}).update(ast.literal_eval(m.group(4)))
print("%-16s %10s %8s %7s %6s %5s %5s %5s %9s %6s %8s" % ("kernel", "grid", "waves", "VALU/w", "SALU/w", "LDS/w", "RD/w", "WR/w", "busy_Mcyc", "valu%", "ldsbusy%"))
for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    w, busy = d["SQ_WAVES"], d.get("SQ_BUSY_CYCLES", 0) / 32
    g = lambda n: d.get(n, 0)
    print("%-16s %10d %8d %7.0f %6.0f %5.0f %5.1f %5.1f %9.2f %6.0f %8.0f" % (k[0], k[1], w, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w,
          g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_VMEM_WR") / w, busy / 1e6, 100 * g("SQ_INSTS_VALU") * 4 / 1024 / max(busy, 1), 100 * g("SQ_ACTIVE_INST_LDS") * 4 / 256 / max(busy, 1)))
