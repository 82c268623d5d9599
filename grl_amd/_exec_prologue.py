#!/usr/bin/env python3
"""Work around a register-allocation bug of ROCm 7.2's clang in gfx950 assembly.

Where lanes re-join after a divergent region the compiler restores the mask with `s_or_b64 exec, exec, sN` at the
head of the join block.  When a lane-mask copy (`s_mov_b64`) already sits at that head, the register allocator puts
its own live-range-split copies, reloads and rematerialised constants IN FRONT of the restore, so they execute with
the narrow mask of the region that just ended; the matching copy elsewhere runs with the full mask, and the lanes
that were masked off receive stale register content (found on the 5-action rollout kernel by
poisoning the register file).  This filter rewrites each such block head to

    [scalar / mask-independent instructions, unchanged order]  s_or_b64 exec, exec, sN  [vector copies]

which is what the allocator assumed: its copies are whole-register copies.  Only block heads made of plain copies
are touched; anything else is left alone and reported.  Usage: fix_exec_prologue.py in.s out.s"""
import re
import sys

MBB = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")
EXEC_OR = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)\s*$")
# vector writes the allocator inserts: copies, AGPR moves, rematerialised constants
VEC = re.compile(r"^\s*(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_read_b32|v_accvgpr_write_b32|v_accvgpr_mov_b32)\s+(\S+?),\s*(\S+)\s*$")
# instructions that do not depend on exec
SCAL = re.compile(r"^\s*(s_mov_b32|s_mov_b64|s_nop|v_readlane_b32|v_writelane_b32)\s+(.*)$")


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([vsa])\[(\d+):(\d+)\]|\b([vsa])(\d+)\b", tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def fix(lines):
    out = []
    fixed = skipped = 0
    i = 0
    n = len(lines)
    while i < n:
        out.append(lines[i])
        if not MBB.match(lines[i]):
            i += 1
            continue
        j = i + 1
        head = []                                   # (line, kind) of the block head
        while j < n:
            if MBB.match(lines[j]) or re.match(r"^\S+:", lines[j]):
                break                               # next block: this one has no exec restore at its head
            code = lines[j].split(";")[0].rstrip()
            if not code.strip():
                head.append((lines[j], "c"))        # comment / blank
            elif VEC.match(code):
                head.append((lines[j], "v"))
            elif SCAL.match(code) and "exec" not in code:
                head.append((lines[j], "s"))
            else:
                break
            j += 1
        code = lines[j].split(";")[0].rstrip() if j < n else ""
        if j < n and EXEC_OR.match(code) and any(k == "v" for _, k in head):
            vec_written = set()
            ok = True
            for l, k in head:
                c = l.split(";")[0]
                if k == "v":
                    m = VEC.match(c.rstrip())
                    vec_written |= regs(m.group(2))
                elif k == "s":
                    m = SCAL.match(c.rstrip())
                    ops = m.group(2).split(",")
                    if regs(",".join(ops[1:])) & vec_written:
                        ok = False                  # a scalar instruction consumes a moved vector write: leave the block alone
            if ok:
                out += [l for l, k in head if k != "v"]
                out.append(lines[j])
                out += [l for l, k in head if k == "v"]
                fixed += 1
                i = j + 1
                continue
            skipped += 1
        i += 1
    return out, fixed, skipped


def main(argv):
    src = open(argv[1]).read().split("\n")
    out, fixed, skipped = fix(src)
    open(argv[2], "w").write("\n".join(out))
    print(f"fix_exec_prologue: {fixed} block head(s) rewritten, {skipped} left alone", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
