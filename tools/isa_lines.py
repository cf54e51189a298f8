#!/usr/bin/env python3
"""Static instruction mix of one kernel instance by source line (needs a -gline-tables-only -save-temps .s).

  python tools/isa_lines.py <file.s> <mangled-name substring> [lo hi]

Prints, per source line of frenet_lattice_fused.hip (or whichever file the .loc directives name as file 1 / the main file),
the number of VALU fp64 / VALU int / VALU other / SALU / LDS / VMEM instructions attributed to it, and section totals for
[lo, hi] when given.  Static counts - loops are not weighted - but inside one loop body they are the per-iteration mix.
"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else None
hi = int(sys.argv[4]) if len(sys.argv) > 4 else None

F64 = re.compile(r"^v_(fma|mul|add|max|min|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|fract|floor|ceil|trunc|rndne|ldexp|frexp_mant|cmp_\w+|cmpx_\w+|cvt_f64\w*|cvt_\w+_f64)_f64|^v_cvt_f64|^v_cvt_\w+_f64")
INT = re.compile(r"^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|min|max|mul_lo|mul_hi|mul_u32|mad_u32|mad_i32|add3|lshl_add|lshl_or|and_or|or3|add_lshl|alignbit|perm|mbcnt|ffbh|ffbl|bcnt|cmp_\w+|cmpx_\w+)_(u|i)(16|24|32|64)|^v_(add|sub|subrev|addc|subb)_co|^v_(lshlrev|lshrrev|ashrrev)_b(32|64)|^v_(and|or|xor|not)_b32|^v_mul_(u32_u24|i32_i24|hi_u32_u24)|^v_mad_(u32_u24|i32_i24|u64_u32|i64_i32)|^v_mbcnt|^v_bfe|^v_bfi|^v_lshl_add|^v_add3|^v_lshl_or|^v_and_or|^v_or3|^v_xad|^v_add_lshl")


def classify(op):
    if op.startswith("s_"):
        if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
            return "wait"
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            return "branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if F64.match(op):
            return "f64"
        if INT.match(op):
            return "int"
        return "vother"
    return "other"


inside = False
cur = 0
per = collections.defaultdict(collections.Counter)
main_file = None
for line in open(path):
    s = line.strip()
    if not inside:
        if s.endswith(":") or ": ;" in s:
            if key in s.split(":")[0] and not s.startswith("."):
                inside = True
        continue
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        fno, ln = int(m.group(1)), int(m.group(2))
        if fno == 1:  # instructions of inlined helpers (other files) stay with the last line of the main file
            cur = (fno, ln)
        continue
    if not s or s.startswith((".", ";")) or s.endswith(":"):
        continue
    op = s.split()[0]
    per[cur][classify(op)] += 1

cols = ["f64", "int", "vother", "salu", "lds", "vmem", "branch", "wait"]
tot = collections.Counter()
print(f"{'file:line':>10s} " + " ".join(f"{c:>7s}" for c in cols))
for (fno, ln) in sorted(per):
    c = per[(fno, ln)]
    if lo is not None and (fno != 1 or ln < lo or ln > hi):
        continue
    tot.update(c)
    print(f"{fno:>3d}:{ln:<6d} " + " ".join(f"{c[k]:7d}" for k in cols))
print(f"{'total':>10s} " + " ".join(f"{tot[k]:7d}" for k in cols))
