#!/usr/bin/env python3
"""Audit of the kernels whose accumulators live in literally named AGPRs (csrc/mla_prefill_flash.hip, ...):
python tools/check_flash_asm.py <file.s> [kernel-name-substring ...]
The compiler must not touch the accumulator file in them: no v_accvgpr_* / a[...] operand outside ;;#ASMSTART..;;#ASMEND,
no scratch access, .vgpr_spill_count 0, .private_segment_fixed_size 0.  Exit status 1 on a violation."""
import re
import sys


def audit(path, names):
    text = open(path).read().splitlines()
    bad = []
    kernel, in_asm = None, False
    seen = set()
    for ln, line in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1) if any(n in m.group(1) for n in names) else None
            if kernel:
                seen.add(kernel)
        if kernel is None:
            continue
        if line.strip().startswith(".Lfunc_end"):
            kernel = None
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
        elif ";;#ASMEND" in line:
            in_asm = False
        elif not in_asm:
            code = line.split(";")[0]
            if re.search(r"\bv_accvgpr|\ba\[?\d+|scratch_", code):
                bad.append((ln, line.strip()))
    meta = "\n".join(text)
    for n in names:
        for m in re.finditer(r"\.name:\s+(\S*%s\S*)\n(?:.*\n){0,40}?" % re.escape(n), meta):
            pass
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)\n(?:.*\n){0,12}?\s*\.symbol:\s*(\S+)\.kd(?:.*\n){0,12}?\s*\.vgpr_spill_count:\s*(\d+)", meta):
        if any(n in m.group(2) for n in names) and (int(m.group(1)) or int(m.group(3))):
            bad.append((0, f"{m.group(2)}: private_segment {m.group(1)} vgpr_spill {m.group(3)}"))
    return seen, bad


if __name__ == "__main__":
    path = sys.argv[1]
    names = sys.argv[2:] or ["flash"]
    seen, bad = audit(path, names)
    print("audited:", sorted(seen))
    for ln, line in bad:
        print(f"VIOLATION line {ln}: {line}")
    sys.exit(1 if bad or not seen else 0)
