#!/usr/bin/env python3
"""Checks a gfx950 ISA dump (hipcc -save-temps, *.s) for the DPP read hazard around inline-assembly
v_fmac_f64_dpp / v_mov_b64_dpp instructions (csrc/sweep_kernels.hpp: fmac_bcast, mov_bcast).

gfx9 rules: a DPP instruction must not read a VGPR that a VALU instruction wrote less than 2 wait
states earlier, nor follow a VALU write of EXEC by less than 5; and (the matrix pipe's result hazard,
which the compiler's recognizer pads for its own VALU readers) a VGPR written by v_mfma_* must not
be read by a DPP instruction within 19 wait states (16 passes of the fp64 16x16x4 form + 3).  The
compiler pads the instructions it knows; inline assembly is opaque to it, so a build is checked here
instead.  The instructions in front of a DPP instruction are walked BACKWARDS over the control-flow
graph: at a label every branch that targets it is a predecessor as well as the fall-through, so a
hazard that arrives through a jump is seen too (the branch instruction itself is one wait state).
    tools/ab_build.sh x -save-temps=obj; python tools/check_dpp_hazard.py build/ab/gpdla-hip-amdgcn-amd-amdhsa-gfx950.s
Exit status 1 and a listing if a hazard is found.
"""
import re
import sys

MFMA_STATES = 19


def regs(operand):
    """v[a:b] / v7 -> set of VGPR indices (other operands: empty)."""
    m = re.fullmatch(r"-?\|?v\[(\d+):(\d+)\]\|?", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", operand)
    return {int(m.group(1))} if m else set()


def parse(path):
    """[(op, operands, text)] of the whole dump and {label: index of the instruction behind it}."""
    insts, labels = [], {}
    for line in open(path):
        text = line.split(";")[0].strip()
        if not text or text.startswith("#"):
            continue
        if text.endswith(":"):
            labels[text[:-1]] = len(insts)
            continue
        if text.startswith("."):
            continue
        parts = text.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        insts.append((op, ops, text))
    return insts, labels


def main(path):
    insts, labels = parse(path)
    n = len(insts)
    # predecessors: the fall-through (unless the previous instruction never falls through) and every branch to a label here
    jumps = {}
    for i, (op, ops, _) in enumerate(insts):
        if (op == "s_branch" or op.startswith("s_cbranch")) and ops and ops[-1] in labels:
            jumps.setdefault(labels[ops[-1]], []).append(i)

    def preds(i):
        out = list(jumps.get(i, []))
        if i > 0 and insts[i - 1][0] not in ("s_branch", "s_endpgm", "s_setpc_b64"):
            out.append(i - 1)
        return out

    bad = checked = 0
    for i, (op, ops, text) in enumerate(insts):
        if not op.endswith("_dpp"):
            continue
        checked += 1
        src0 = regs(ops[1].split()[0])
        srcs = set()
        for o in ops[1:]:
            if o.split():
                srcs |= regs(o.split()[0])
        seen = set()
        stack = [(p, 0) for p in preds(i)]
        reported = set()
        while stack:
            j, states = stack.pop()
            if (j, states) in seen or states >= MFMA_STATES:
                continue
            seen.add((j, states))
            jop, jops, jtext = insts[j]
            is_valu = jop.startswith("v_")
            wrote = regs(jops[0]) if is_valu and jops else set()
            wrote_exec = is_valu and (jop.startswith("v_cmpx") or (jops and jops[0] == "exec"))
            if states < 2 and wrote & src0 and (j, "src") not in reported:
                reported.add((j, "src"))
                print(f"hazard: '{jtext}' writes the DPP source of '{text}' {states} wait state(s) earlier")
                bad += 1
            if states < 5 and wrote_exec and (j, "exec") not in reported:
                reported.add((j, "exec"))
                print(f"hazard: '{jtext}' writes EXEC {states} wait state(s) before '{text}'")
                bad += 1
            if jop.startswith("v_mfma") and wrote & srcs and (j, "mfma") not in reported:
                reported.add((j, "mfma"))
                print(f"hazard: '{jtext}' (matrix pipe) writes a source of '{text}' {states} wait state(s) earlier")
                bad += 1
            took = int(jops[0], 0) + 1 if jop == "s_nop" else 1
            for p in preds(j):
                stack.append((p, states + took))
    print(f"{checked} DPP instructions checked, {bad} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
