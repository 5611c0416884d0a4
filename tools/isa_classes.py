#!/usr/bin/env python3
"""tools/isa_classes.py <file.hip> <kernel-name substring> [...]: the vector instructions of a kernel by ISSUE CLASS.

Classes and their cycles per wave64 instruction on one SIMD are the ones measured on the MI355X in round 4
(profiles/r04_ubench_issue_rate_run2_wall.txt, 8 wavefronts per SIMD, 16 independent chains; profiles/r04_ubench_issue_mix.txt for mixes):

  fast   2.2-3.2  VOP2 integer add / sub / and / or / xor / not / mov, lshrrev_b32, ashrrev_i32, add / sub / mul / fmac f32, add / max f16, add_u16
  mid    2.5      v_bitop3_b32, v_fma_f32 (when their register operands do not collide on a bank: 4.4 then)
  slow   4.3      every VOP3 / VOP3P / SDWA / DPP form and the other VOP2s: pk_* i16 / f16 / f32, perm, bfe, bfi, alignbit, max / min,
                  lshlrev_b32 (!), mad, mul_lo, cvt, cmp, cndmask, sad, dot
  trans  8.3      rcp / rsq / sqrt / sin / cos / exp / log, fma_f16

and the rule the mixes showed: a SIMD that alternates between fast and slow instructions runs BOTH at the slow rate (alt_add_pkadd
4.52, three adds to one pk_add 4.13 average) -- the fast rate only pays in runs of fast instructions.  So next to the class counts the table
gives the share of fast / mid instructions that sit in runs of at least 8 of their kind: what a re-spelling of slow opcodes as fast
ones could at most recover is bounded by that structure, not by the class counts alone.

Per kernel: the whole function, and its hottest loop (the backward branch whose body holds the most vector instructions; inner loops win
ties).  Scalar, LDS, memory and wait instructions are counted separately; they issue from other ports.
"""
import collections
import re
import subprocess
import sys
import tempfile

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_add_f16", "v_max_f16", "v_add_u16", "v_add_co_u32", "v_sub_co_u32", "v_nop"}
MID = {"v_bitop3_b32", "v_fma_f32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_fma_f16", "v_rcp_f16", "v_rcp_iflag_f32", "v_div_fixup_f32",
         "v_div_fmas_f32", "v_div_scale_f32"}
CYC = {"fast": 2.2, "mid": 2.5, "slow": 4.3, "trans": 8.3}


def classify(op, line):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    forced_slow = op.endswith(("_e64", "_sdwa", "_dpp")) or " dst_sel:" in line or "row_" in line or "quad_perm" in line  # the VOP3 / SDWA / DPP encodings issue at the slow rate
    if base in TRANS:
        return "trans"
    if base in MID:
        return "mid"
    if base in FAST and not forced_slow:
        return "fast"
    return "slow"


def parse(asm, name_part):
    out = []
    for m in re.finditer(r"^(\S*%s\S*):\s*;\s*@" % re.escape(name_part), asm, re.M):
        i = m.end()
        k = asm.index(".Lfunc_end", i)
        out.append((m.group(1), asm[i:k].splitlines()))
    return out


def table(lines):
    cls, other = collections.Counter(), collections.Counter()
    runs, cur_kind, cur_len, in_runs = [], None, 0, 0
    ops = collections.Counter()
    for l in lines:
        l = l.split(";")[0].strip()
        m = re.match(r"([a-z_0-9]+)\b", l)
        if not m or l.endswith(":") or l.startswith("."):
            continue
        op = m.group(1)
        if op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            c = classify(op, l)
            cls[c] += 1
            ops[(c, re.sub(r"_(e32|e64)$", "", op))] += 1
            kind = "quick" if c in ("fast", "mid") else "slow"
        else:
            other["SALU" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_barrier", "s_load", "s_nop", "s_cbranch", "s_branch", "s_endpgm"))
                  else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "SMEM" if op.startswith("s_load")
                  else "wait/branch/other"] += 1
            kind = None  # scalar and memory instructions issue beside the vector ALU: they do not break a run
            continue
        if kind == cur_kind:
            cur_len += 1
        else:
            if cur_kind == "quick" and cur_len >= 8:
                in_runs += cur_len
            cur_kind, cur_len = kind, 1
    if cur_kind == "quick" and cur_len >= 8:
        in_runs += cur_len
    return cls, other, ops, in_runs


def loops(lines):
    """(start, end) line ranges of every backward branch."""
    labels = {}
    for n, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if m:
            labels[m.group(1)] = n
    out = []
    for n, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t < n:
                out.append((t, n))
    return out


def report(name, lines, f):
    import shutil
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() if shutil.which("c++filt") else name
    dem = re.sub(r"\(anonymous namespace\)::", "", dem)
    f.write("== %s\n" % re.sub(r"\(.*", "", dem))
    best = None
    for (a, b) in loops(lines):
        nv = sum(1 for l in lines[a:b] if re.match(r"\s*v_", l))
        if best is None or nv > best[0] or (nv == best[0] and b - a < best[2] - best[1]):
            best = (nv, a, b)
    for title, body in (("whole kernel", lines), ("hottest loop (%d lines)" % (best[2] - best[1]) if best else None, lines[best[1]:best[2]] if best else None)):
        if body is None:
            continue
        cls, other, ops, in_runs = table(body)
        tot = sum(cls.values())
        if not tot:
            continue
        cyc = sum(CYC[c] * n for c, n in cls.items())
        quick = cls["fast"] + cls["mid"]
        f.write("  %s: %d vector instructions -- fast %d (%.0f %%), mid %d (%.0f %%), slow %d (%.0f %%), trans %d (%.0f %%)\n"
                % (title, tot, cls["fast"], 100 * cls["fast"] / tot, cls["mid"], 100 * cls["mid"] / tot, cls["slow"], 100 * cls["slow"] / tot, cls["trans"], 100 * cls["trans"] / tot))
        f.write("      issue cycles if every class ran at its own rate: %.0f (%.2f per instruction); at the mixed-stream rate (everything but trans at 4.3): %.0f\n"
                % (cyc, cyc / tot, 4.3 * (tot - cls["trans"]) + 8.3 * cls["trans"]))
        f.write("      fast / mid instructions in runs of >= 8 of their kind: %d of %d (%.0f %% of the vector instructions) -- the part that can run at the fast rate as the code stands\n"
                % (in_runs, quick, 100 * in_runs / tot))
        f.write("      beside the vector ALU: %s\n" % ", ".join("%s %d" % kv for kv in sorted(other.items())))
        top = sorted(ops.items(), key=lambda kv: -kv[1])[:14]
        f.write("      most frequent: %s\n" % ", ".join("%s %d (%s)" % (k[1], n, k[0]) for k, n in top))


def main():
    src, names = sys.argv[1], sys.argv[2:]
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", src, "-o", t.name],
                       check=True, capture_output=True)
        asm = open(t.name).read()
    for n in names:
        for name, lines in parse(asm, n):
            report(name, lines, sys.stdout)


if __name__ == "__main__":
    main()
