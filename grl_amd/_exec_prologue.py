#!/usr/bin/env python3
"""Work around a register-allocation bug of ROCm 7.2's clang (AMD clang 22.0.0git) for gfx950.

Where lanes re-join after a divergent region the compiler restores the mask with `$exec = S_OR_B64 $exec, sN` as the
FIRST instruction of the join block.  The register allocator inserts its live-range-split copies, reloads and
rematerialised constants "after the block prologue"; `SIInstrInfo::isBasicBlockPrologue` does not count a scalar COPY
as prologue, so when phi elimination has put a lane-mask copy at the head of the join block, the allocator's vector
copies land IN FRONT of the restore and execute under the narrow mask of the region that just ended.  The matching copy
elsewhere runs with the full mask, and the lanes that were masked off receive stale register content (DESIGN.md section
4.1f: found on the 5-action rollout kernel by poisoning the register file).

Detection is exact, not textual.  In the final assembly a then-block's own last copies (which MUST run narrow) can sit
in front of the same `s_or_b64 exec` once branch folding has merged or tail-duplicated blocks, and no pattern tells the
two apart -- round 2's text filter moved one such then-tail copy of the target-network kernel.  So the build asks the
compiler for its machine code right after register allocation (`-mllvm -print-after=stack-slot-coloring`), where every
join block still is a block of its own that starts with its restore:

  find_misplaced(mir)   every block whose first exec-writing instruction WIDENS the mask (`$exec = S_OR_B64 $exec, ..`,
                        `S_OR_SAVEEXEC_B64`) although mask-dependent instructions stand in front of it.  Mask-independent
                        and allowed there: scalar instructions that do not read exec, SGPR spills to / from VGPR lanes.
  apply(asm, found)     in the assembly of the same compilation, block <function, bb.N> is rewritten to
                            [mask-independent instructions]  s_or_b64 exec, exec, sN  [the copies]
                        -- what the allocator assumed, its copies being whole-register copies.

Both steps FAIL CLOSED; the build (grl_amd/_build.py) stops on any entry of the returned problem list:
  * a misplaced instruction that is not a plain register copy or a rematerialised constant (a scratch reload, an ALU
    operation), or a widening restore other than `S_OR_B64 $exec`;
  * a block the assembly does not hold as a block of its own, whose restore uses another mask register, or whose vector
    destinations differ from the ones the machine code names;
  * a copy that cannot be moved without crossing a dependency: read-after-write, write-after-read or write-after-write
    against an instruction that stays in place (`v_writelane_b32` counts as a vector write, `v_readlane_b32` as a scalar
    write), or a copy that reads exec or vcc.
What this does not see: a miscompile of another kind.  The poisoned-register runs of the GPU suite (tests/conftest.py,
GRLX_POISON_REGISTERS) are the independent check.

Usage: _exec_prologue.py mir.txt in.s out.s"""
import re
import sys

# ------------------------------------------------------------------ machine code (after register allocation) -------
_FN = re.compile(r"^# Machine code for function (\S+): (.*)$")
_BB = re.compile(r"^bb\.(\d+)[^:]*:")
_SLOT = re.compile(r"^\d+B\t")
_PHYS = re.compile(r"\$(vgpr|agpr|sgpr)(\d+)((?:_(?:vgpr|agpr|sgpr)\d+)*)")
_WIDEN = re.compile(r"^\$exec = S_OR_B64(_term)? .*\$exec|= S_OR_SAVEEXEC_B64 ")
# other ways of writing exec whose effect on the mask the text does not tell: a block that begins with one of them behind
# mask-dependent instructions is refused (fail closed), unless the value is visibly a NARROWED copy of the current mask
_MAY_WIDEN = re.compile(r"^\$exec = (S_MOV_B64|S_XOR_B64|S_XNOR_B64|S_OR_B64|S_ORN2_B64|S_NOT_B64|S_CSELECT_B64)(_term)? ")
_MASK_REG = r"\$(sgpr\d+_sgpr\d+|vcc|exec)\b"
_FIXABLE_RESTORE = re.compile(r"^\$exec = S_OR_B64 (?:killed )?\$exec, (?:killed )?(?:renamable )?\$(sgpr\d+_sgpr\d+|vcc)\b")


def _mir_regs(text):
    out = set()
    for m in _PHYS.finditer(text):
        k = {"vgpr": "v", "agpr": "a", "sgpr": "s"}[m.group(1)]
        out.add((k, int(m.group(2))))
        for n in re.findall(r"\d+", m.group(3)):
            out.add((k, int(n)))
    return out


def _mir_opcode(t):
    m = re.search(r"(?:^|= )([A-Z][A-Za-z0-9_]+)\b", t)
    return m.group(1) if m else ""


def _mir_writes_exec(t):
    if re.match(r"^\$exec(_lo|_hi)? = ", t):
        return True
    return "implicit-def $exec" in t or "implicit-def dead $exec" in t


def _mir_classify(t):
    """'s' mask-independent, 'v' mask-dependent and movable (whole-register copy / rematerialised constant),
    'x' mask-dependent and not movable (or unknown)."""
    op = _mir_opcode(t)
    if op in ("SI_SPILL_S32_TO_VGPR", "SI_RESTORE_S32_FROM_VGPR", "IMPLICIT_DEF", "KILL", "DBG_VALUE", "S_NOP", "S_WAITCNT", "BUNDLE", "CFI_INSTRUCTION"):
        return "s"
    dst = t.split(" = ")[0] if " = " in t else ""
    vec_def = bool(re.search(r"\$(vgpr|agpr)\d", dst))
    if op == "COPY":
        if not vec_def:
            return "s" if "$exec" not in t.split(" = ", 1)[1] else "x"
        src = t.split(" = ", 1)[1]
        return "v" if re.search(r"COPY (killed |renamable |undef )*\$(vgpr|agpr|sgpr)\d", src) and "$exec" not in src and "$vcc" not in src else "x"
    if op in ("V_MOV_B32_e32", "V_MOV_B64_PSEUDO", "V_MOV_B64_e32", "V_ACCVGPR_WRITE_B32_e64", "V_ACCVGPR_READ_B32_e64", "V_ACCVGPR_MOV_B32", "AV_MOV_B32_IMM_PSEUDO", "AV_MOV_B64_IMM_PSEUDO"):
        src = t.split(" = ", 1)[1] if " = " in t else t
        body = src.replace("implicit $exec", "")
        return "v" if "$vcc" not in body and "$exec" not in body else "x"
    if op.startswith("S_") and "$exec" not in t and not vec_def:
        return "s"
    return "x"


def _narrowing_mov(first, head):
    """`$exec = S_MOV_B64[_term] X` where X was computed in this block as (current mask) AND something: the lowering of an
    `if`, which can only take lanes away."""
    m = re.match(r"^\$exec = S_MOV_B64(?:_term)? (?:killed )?(?:renamable )?" + _MASK_REG, first)
    if not m:
        return False
    src = m.group(1)
    masks = {"exec"}                                    # registers that hold a copy of the current mask
    for t in head:
        c = re.match(r"^(?:dead )?(?:renamable )?" + _MASK_REG + r" = COPY (?:killed )?\$exec\b", t)
        if c:
            masks.add(c.group(1))
    for t in reversed(head):
        d = re.match(r"^(?:dead )?(?:renamable )?" + _MASK_REG + r" = ([A-Z0-9_a-z]+) (.*)$", t)
        if not d or d.group(1) != src:
            continue
        ops = re.findall(_MASK_REG, d.group(3))
        if d.group(2) == "S_AND_B64":
            return any(o in masks for o in ops[:2])
        if d.group(2) == "S_ANDN2_B64":
            return bool(ops) and ops[0] in masks
        return False
    return False


def find_misplaced(lines):
    """Parse `-print-after=stack-slot-coloring` output (an iterable of lines).  Returns (found, problems):
    found = [dict(function, bb, restore_mask, vec_dst, head)] for the block heads apply() has to rewrite."""
    final = {}                                       # function -> blocks of its LAST dump that has no virtual registers
    cur = blk = None
    keep = False
    for line in lines:
        line = _SLOT.sub("", line.rstrip("\n"))
        m = _FN.match(line)
        if m:
            keep = "NoVRegs" in m.group(2)
            if keep:
                cur = final[m.group(1)] = []
            blk = None
            continue
        if not keep:
            continue
        if line.startswith("# End machine code"):
            keep = False
            continue
        m = _BB.match(line)
        if m:
            blk = (int(m.group(1)), [])
            cur.append(blk)
            continue
        if blk is not None:
            t = line.strip()
            if t and not t.startswith(("successors:", "liveins:", ";", "predecessors")):
                blk[1].append(t)
    found, problems = [], []
    for fn, blocks in final.items():
        for bb, ins in blocks:
            head, first = [], None
            for t in ins:
                if _mir_writes_exec(t):
                    first = t
                    break
                head.append(t)
            if first is None:
                continue                            # the block never writes the mask: ordinary code
            kinds = [_mir_classify(t) for t in head]
            if not _WIDEN.search(first):
                if _MAY_WIDEN.match(first) and not _narrowing_mov(first, head) and not all(k == "s" for k in kinds):
                    problems.append(f"{fn} bb.{bb}: mask-dependent instructions in front of `{first.split(',')[0]}`, which may widen the mask "
                                    "(this filter only knows S_OR_B64 $exec / S_OR_SAVEEXEC_B64 as restores and cannot tell)")
                continue                            # the block narrows the mask first: ordinary code
            if all(k == "s" for k in kinds):
                continue
            where = f"{fn} bb.{bb}"
            m = _FIXABLE_RESTORE.match(first)
            if not m:
                problems.append(f"{where}: mask-dependent instructions in front of a restore this filter does not rewrite: {first}")
                continue
            bad = [t for t, k in zip(head, kinds) if k == "x"]
            if bad:
                problems.append(f"{where}: in front of the exec restore, neither mask-independent nor a plain copy: {bad[0]}")
                continue
            vec_dst, vec_src = set(), set()
            for t, k in zip(head, kinds):
                if k == "v":
                    vec_dst |= {r for r in _mir_regs(t.split(" = ")[0]) if r[0] in "va"}
                    vec_src |= _mir_regs(t.split(" = ", 1)[1])
            mask = m.group(1)
            mask_regs = ("vcc",) if mask == "vcc" else tuple(int(n) for n in re.findall(r"\d+", mask))
            found.append(dict(function=fn, bb=bb, restore_mask=mask_regs, vec_dst=vec_dst, vec_src=vec_src, head=head))
    return found, problems


# ------------------------------------------------------------------------------------------ assembly ---------------
MBB = re.compile(r"^(\.LBB\d+_(\d+):|; %bb\.(\d+):)")
LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
EXEC_OR = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*(s\[(\d+):(\d+)\]|vcc)\s*$")
# whole-register copies the allocator inserts: copies, AGPR moves, rematerialised constants
VEC = re.compile(r"^\s*(v_mov_b32_e32|v_mov_b64_e32|v_mov_b32|v_mov_b64|v_accvgpr_read_b32|v_accvgpr_write_b32|v_accvgpr_mov_b32)\s+(\S+?),\s*(\S+)\s*$")
LANE = re.compile(r"^\s*(v_readlane_b32|v_writelane_b32)\s+(\S+?),\s*(\S+?),\s*(\S+)\s*$")
REG = re.compile(r"\b([vsa])\[(\d+):(\d+)\]|\b([vsa])(\d+)\b|\b(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0|scc)\b")


HAZARD_NOP = "\ts_nop 4"                    # inserted behind moved copies (rewrite_head)


def regs(tok):
    """Registers named in an operand string: ('v', 3), ('s', 26), ('vcc', 0), ('exec', 0) ..."""
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        elif m.group(4):
            out.add((m.group(4), int(m.group(5))))
        else:
            out.add((m.group(6).split("_")[0], 0))
    return out


def _code(line):
    return line.split(";")[0].rstrip()


def classify(code):
    """(kind, reads, writes) of one assembly instruction: 'v' movable copy, 's' mask-independent, 'x' anything else."""
    m = VEC.match(code)
    if m:
        return "v", regs(m.group(3)), regs(m.group(2))
    m = LANE.match(code)
    if m:
        if m.group(1) == "v_readlane_b32":
            return "s", regs(m.group(3)) | regs(m.group(4)), regs(m.group(2))
        return "s", regs(m.group(2)) | regs(m.group(3)) | regs(m.group(4)), regs(m.group(2))
    op = code.split()[0]
    if "saveexec" in op or op.startswith("v_cmpx"):
        return "x", {("exec", 0)}, {("exec", 0)}    # writes exec without naming it
    if op in ("s_waitcnt", "s_nop", "s_waitcnt_vscnt", "s_waitcnt_lgkmcnt", "s_waitcnt_vmcnt", "s_waitcnt_expcnt", "s_sleep"):
        return "s", set(), set()
    if op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_call", "s_barrier")):
        r = regs(code[len(op):])
        if ("exec", 0) in r:
            return "x", r, r
        # a scalar ALU / move / load: which operands it writes is not decoded, so all of them count as read AND written
        return "s", r | {("scc", 0)}, r | {("scc", 0)}
    return "x", set(), set()                        # vector ALU, memory, LDS, v_readfirstlane, branches ...


def rewrite_head(block_lines, want_mask=None, want_vec_dst=None, want_vec_src=None):
    """block_lines: the lines of ONE assembly block after its label.  Returns (new_lines, None) with the copies in front
    of the first `s_or_b64 exec, exec, X` moved behind it, or (None, reason)."""
    head, j = [], 0
    while j < len(block_lines):
        code = _code(block_lines[j])
        if not code.strip() or code.lstrip().startswith("."):
            head.append((block_lines[j], "c", set(), set()))
            j += 1
            continue
        m = EXEC_OR.match(code)
        if m:
            mask = ("vcc",) if m.group(1) == "vcc" else (int(m.group(2)), int(m.group(3)))
            break
        kind, rd, wr = classify(code)
        if kind == "x":
            return None, "in front of the restore: " + code.strip()
        head.append((block_lines[j], kind, rd, wr))
        j += 1
    else:
        return None, "no `s_or_b64 exec, exec, ..` in the block"
    if want_mask is not None and tuple(want_mask) != mask:
        return None, f"the restore uses {mask}, the machine code says {tuple(want_mask)}"
    code_head = [h for h in head if h[1] != "c"]
    vec_dst, vec_src = set(), set()
    for a, (_, k, rd, wr) in enumerate(code_head):
        if k != "v":
            continue
        vec_dst |= wr
        vec_src |= rd
        if {("exec", 0), ("vcc", 0)} & rd:
            return None, "a copy reads exec or vcc"
        if any(r[0] not in ("v", "a") for r in wr):
            return None, "a copy writes a scalar register"
        for l2, k2, rd2, wr2 in code_head[a + 1:]:       # everything that stays and came AFTER this copy is crossed by the move
            if k2 == "s" and ((wr & rd2) or (rd & wr2) or (wr & wr2)):
                return None, "moving a copy would cross a dependency with: " + _code(l2).strip()
    if not vec_dst:
        return None, "no copy in front of the restore"
    if want_vec_dst is not None and set(want_vec_dst) != vec_dst:
        return None, f"copies write {sorted(vec_dst)}, the machine code says {sorted(want_vec_dst)}"
    if want_vec_src is not None and set(want_vec_src) != vec_src:
        return None, f"copies read {sorted(vec_src)}, the machine code says {sorted(want_vec_src)}"
    # The copies now stand directly in front of the block's next instruction, behind the hazard recogniser's back: the wait states the
    # restore and the scalar instructions of the head used to provide are gone.  `s_nop 4` (five wait states) covers every distance a
    # VALU write of a VGPR needs on this family before a lane access / DPP read (1-2) or a matrix-core operand read of that register.
    new = [l for l, k, _, _ in head if k != "v"] + [block_lines[j]] + [l for l, k, _, _ in head if k == "v"] + [HAZARD_NOP] + block_lines[j + 1:]
    return new, None


def apply(asm_lines, found):
    """Rewrite the block heads find_misplaced() named.  Block numbers change between register allocation and the
    assembly (block placement renumbers), physical registers do not: a misplaced head is recognised in the assembly of
    ITS function as a block that starts with exactly its copies (same destination and source registers) and restores
    exec from the same mask register.  Every such block is rewritten (tail duplication may have cloned a join block);
    none at all is a problem.  Returns (lines, n_fixed, problems)."""
    by_fn = {}
    for f in found:
        by_fn.setdefault(f["function"], []).append(f)
    out, problems, fixed = [], [], 0
    fn = None
    hits = {id(f): 0 for f in found}
    i, n = 0, len(asm_lines)
    while i < n:
        line = asm_lines[i]
        out.append(line)
        i += 1
        if not MBB.match(line):
            lab = LABEL.match(line)
            if lab and lab.group(1) in by_fn:
                fn = lab.group(1)
            elif line.startswith(".Lfunc_end"):
                fn = None
            continue
        if fn is None:
            continue
        j = i
        while j < n and not (MBB.match(asm_lines[j]) or LABEL.match(asm_lines[j])):
            j += 1
        for f in by_fn[fn]:
            new, why = rewrite_head(asm_lines[i:j], f["restore_mask"], f["vec_dst"], f["vec_src"])
            if new is not None:
                out += new
                fixed += 1
                hits[id(f)] += 1
                i = j
                break
    for f in found:
        if not hits[id(f)]:
            problems.append(f"{f['function']} bb.{f['bb']} (block number after register allocation): the assembly has no block that starts with "
                            f"copies into {sorted(f['vec_dst'])} in front of `s_or_b64 exec, exec, {f['restore_mask']}`")
    return out, fixed, problems


def main(argv):
    with open(argv[1], errors="replace") as f:
        found, problems = find_misplaced(f)
    with open(argv[2]) as f:
        out, fixed, p2 = apply(f.read().split("\n"), found)
    with open(argv[3], "w") as f:
        f.write("\n".join(out))
    for p in problems + p2:
        print("fix_exec_prologue: PROBLEM:", p, file=sys.stderr)
    print(f"fix_exec_prologue: {len(found)} misplaced block head(s) in the machine code, {fixed} rewritten", file=sys.stderr)
    return 1 if problems or p2 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
