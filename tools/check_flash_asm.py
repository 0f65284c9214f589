#!/usr/bin/env python3
"""Audit of the gfx950 assembly of a kernel that pins its MFMA accumulators to the AGPR file through asm operands
(csrc/mla_prefill_flash.hip): python tools/check_flash_asm.py <file.s> <kernel-name-substring> [max_accvgpr_in_loop | literal]
Checks, for every kernel whose mangled name contains the substring:
  * .private_segment_fixed_size 0 and .vgpr_spill_count 0 (no scratch: a spill in a one-wave-per-SIMD MFMA loop is the
    3x slowdown this structure exists to avoid), no scratch_* instruction;
  * .agpr_count 256 (the accumulator really is in the AGPR file);
  * no basic block that holds >= 16 MFMAs (the QK and PV phases of a key block) moves accumulator registers
    (v_accvgpr_* count <= `max_accvgpr_in_loop`, default 0): the only accumulator traffic is the rare rescale branch, the
    zero fill and the read-out.
Exit status 1 on a violation."""
import re
import sys


def kernels(text, name):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = (m.group(1), []) if name in m.group(1) else (None, [])
        if cur is not None:
            body.append(line)
            if line.strip().startswith(".Lfunc_end"):
                yield cur, body
                cur = None



def _regs_of(tok):
    """Register numbers named by an operand token: v12 -> {('v', 12)}, v[10:13] -> v10..v13, s[4:5] -> s4, s5."""
    out = set()
    for kind, a, b in re.findall(r"\b([vs])\[(\d+):(\d+)\]", tok):
        out.update((kind, i) for i in range(int(a), int(b) + 1))
    for kind, a in re.findall(r"\b([vs])(\d+)\b", tok):
        out.add((kind, int(a)))
    return out


def audit_async_asm_loads(path, name):
    """Loads written as inline asm whose OUTPUT operand the compiler believes defined when the asm statement ends
    (`global_load_dword %0, ...` / `s_load_dword %0, ...` with the wait in a LATER asm statement -- csrc/mla_decode.hip heads its
    kernel that way to put the seqlens and page-table requests in front of everything else).  The destination register is in
    flight until that later hand-written `s_waitcnt`: any instruction in between that reads or writes it (a copy, a spill, a
    coalescing move the compiler is entitled to insert) would see a stale value.  This reads the compiled ISA linearly from each
    such load to the first asm-block `s_waitcnt` that covers it (vmcnt(0) for vector loads, lgkmcnt(0) for scalar ones) and
    reports every instruction that names the register.  Returns (kernels seen, violations)."""
    text = open(path).read()
    bad, seen = [], []
    for k, body in kernels(text, name):
        seen.append(k)
        in_asm, pending = False, {}  # (kind, n) -> line of the load
        for ln in body:
            x = ln.strip()
            if ";;#ASMSTART" in x:
                in_asm = True
                continue
            if ";;#ASMEND" in x:
                in_asm = False
                continue
            if not x or x.startswith((";", ".", "//")) or x.endswith(":"):
                continue
            code = x.split(";")[0]
            m = re.match(r"(global_load_dword(?:x\d)?|s_load_dword(?:x\d)?)\s+([^,]+),", code) if in_asm else None
            if m:
                for r in _regs_of(m.group(2)):
                    pending[r] = x
                # (the address operands of this load may not be pending registers either)
                hit = _regs_of(code[m.end():]) & set(pending)
                if hit - _regs_of(m.group(2)):
                    bad.append(f"{k}: `{x}` uses a register still in flight: {sorted(hit)}")
                continue
            if in_asm and code.startswith("s_waitcnt"):
                if "vmcnt(0)" in code:
                    pending = {r: v for r, v in pending.items() if r[0] != "v"}
                if "lgkmcnt(0)" in code:
                    pending = {r: v for r, v in pending.items() if r[0] != "s"}
                continue
            hit = _regs_of(code) & set(pending)
            if hit:
                bad.append(f"{k}: `{x}` touches {sorted(hit)} before the wait for `{pending[sorted(hit)[0]]}`")
        if pending:
            bad.append(f"{k}: asm loads never waited for: {sorted(pending)}")
    return seen, bad


def audit_dma_loops(path, name, min_mfma=8):
    """Kernels whose operand tiles arrive by LDS-DMA (global_load_lds, requested through asm: the compiler does not count
    them): inside a loop that multiplies (>= `min_mfma` MFMAs between its header and its back edge) no `s_waitcnt vmcnt`
    of the compiler's may sit where a DMA request is in flight (between a global_load_lds and the kernel's own
    `s_waitcnt vmcnt(0)`, which is inside ;;#ASMSTART .. ;;#ASMEND).  A wait the compiler inserts there -- for any plain
    load it sees in the loop -- counts in order and therefore also waits for the tiles just requested: the K loops of the
    tiled GEMMs lost 20 % that way (profiles/r05_ab_moe_tiled.txt).  Returns (kernels seen, violations)."""
    text = open(path).read()
    bad, seen = [], []
    for k, body in kernels(text, name):
        seen.append(k)
        labels = {}
        for i, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = i
        for i, ln in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
            if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
                continue
            loop = body[labels[m.group(1)]:i + 1]  # a back edge: header .. branch
            if sum("v_mfma" in x for x in loop) < min_mfma or not any("global_load_lds" in x for x in loop):
                continue
            # two passes over the body = two iterations: requests made late in one are in flight at the top of the next, until
            # the kernel's own `s_waitcnt vmcnt(0)`; a compiler wait is a violation only while a request is in flight
            in_asm, in_flight = False, False
            for rnd in range(2):
                for x in loop:
                    if ";;#ASMSTART" in x:
                        in_asm = True
                    elif ";;#ASMEND" in x:
                        in_asm = False
                    elif "global_load_lds" in x:
                        in_flight = True
                    elif in_asm and re.search(r"s_waitcnt\s+vmcnt\(0\)", x):
                        in_flight = False
                    elif not in_asm and in_flight and rnd == 1 and re.search(r"s_waitcnt\s+.*vmcnt", x):
                        bad.append((k, m.group(1), x.strip()))
    return seen, bad


def audit_literal(path, name):
    """Kernels that name a[0:255] literally in their asm statements: the compiler must not touch the AGPR file at all --
    no v_accvgpr_* and no a[...] / aN operand outside ;;#ASMSTART .. ;;#ASMEND -- and must not spill."""
    text = open(path).read()
    bad, seen = [], []
    for k, body in kernels(text, name):
        seen.append(k)
        in_asm = False
        for ln in body:
            if ";;#ASMSTART" in ln:
                in_asm = True
            elif ";;#ASMEND" in ln:
                in_asm = False
            elif not in_asm:
                code = ln.split(";")[0]
                if re.search(r"\bv_accvgpr|\ba\[\d+:\d+\]|\ba\d+\b|scratch_", code):
                    bad.append(f"{k}: compiler-generated accumulator / scratch access: {code.strip()}")
                    if len(bad) > 8:
                        break
    for m in re.finditer(r"\.name:\s*(\S+)(?:.*\n)*?.*?\.private_segment_fixed_size:\s*(\d+)(?:.*\n)*?.*?\.vgpr_spill_count:\s*(\d+)", text):
        if name in m.group(1) and (int(m.group(2)) or int(m.group(3))):
            bad.append(f"{m.group(1)}: private_segment {m.group(2)} vgpr_spill {m.group(3)}")
    if not seen:
        bad.append(f"no kernel matching {name!r} in {path}")
    return seen, bad


def audit(path, name, max_acc=0):
    text = open(path).read()
    bad, seen = [], []
    for k, body in kernels(text, name):
        seen.append(k)
        code = [ln.split(";")[0] for ln in body]
        if any("scratch_" in c for c in code):
            bad.append(f"{k}: scratch access")
        # basic blocks (split at labels and branches): one that holds MFMAs must not move accumulator registers
        blocks, cur = [], []
        for c in code:
            if re.match(r"^\.LBB", c.strip()) or "s_cbranch" in c or "s_branch" in c:
                blocks.append(cur)
                cur = []
            cur.append(c)
        blocks.append(cur)
        n_mfma_blocks = 0
        for b in blocks:
            n_mfma, n_acc = sum("v_mfma" in c for c in b), sum("v_accvgpr" in c for c in b)
            if n_mfma >= 16:
                n_mfma_blocks += 1
                if n_acc > max_acc:
                    bad.append(f"{k}: a basic block with {n_mfma} MFMAs also has {n_acc} v_accvgpr_* moves")
        if n_mfma_blocks < 2:
            bad.append(f"{k}: expected the QK and PV MFMA blocks, found {n_mfma_blocks} blocks with >= 16 MFMAs")
    for m in re.finditer(r"\.agpr_count:\s*(\d+)(?:.*\n)*?.*?\.name:\s*(\S+)(?:.*\n)*?.*?\.private_segment_fixed_size:\s*(\d+)(?:.*\n)*?.*?\.vgpr_spill_count:\s*(\d+)", text):
        agpr, kname, priv, spill = int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4))
        if name in kname:
            if agpr != 256 or priv or spill:
                bad.append(f"{kname}: agpr_count {agpr} private_segment {priv} vgpr_spill {spill}")
    if not seen:
        bad.append(f"no kernel matching {name!r} in {path}")
    return seen, bad


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "literal":
        seen, bad = audit_literal(sys.argv[1], sys.argv[2])
        print("audited (literal AGPR form):", seen)
        for b in bad:
            print("VIOLATION:", b)
        sys.exit(1 if bad else 0)
    seen, bad = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "flash", int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    print("audited:", seen)
    for b in bad:
        print("VIOLATION:", b)
    sys.exit(1 if bad else 0)
