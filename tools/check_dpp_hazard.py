#!/usr/bin/env python3
"""Checks a gfx950 ISA dump (hipcc -save-temps, *.s) for the DPP read hazard around inline-assembly
v_fmac_f64_dpp / v_mov_b64_dpp instructions (csrc/sweep_kernels.hpp: fmac_bcast, mov_bcast).

gfx9 rule: a DPP instruction must not read a VGPR that a VALU instruction wrote less than 2 wait
states earlier, nor follow a VALU write of EXEC by less than 5.  The compiler pads the instructions
it knows; inline assembly is opaque to it, so a build is checked here instead:
    tools/ab_build.sh x -save-temps=obj; python tools/check_dpp_hazard.py build/ab/gpdla-hip-amdgcn-amd-amdhsa-gfx950.s
Exit status 1 and a listing if a hazard is found.
"""
import re
import sys


def regs(operand):
    """v[a:b] / v7 -> set of VGPR indices (other operands: empty)."""
    m = re.fullmatch(r"-?\|?v\[(\d+):(\d+)\]\|?", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", operand)
    return {int(m.group(1))} if m else set()


def main(path):
    bad = checked = 0
    window = []  # recent instructions, newest last: (wait states it takes, VGPRs a VALU op wrote, writes exec by VALU, text)
    for line in open(path):
        text = line.split(";")[0].strip()
        if not text or text.startswith(".") or text.endswith(":") or text.startswith("#"):
            continue  # (labels: the fall-through predecessor is the one checked)
        parts = text.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if op.endswith("_dpp"):
            checked += 1
            src0 = regs(ops[1].split()[0])
            states = 0
            for took, wrote, wrote_exec, t in reversed(window):
                if states < 2 and wrote & src0:
                    print(f"hazard: '{t}' writes the DPP source of '{text}' {states} wait state(s) earlier")
                    bad += 1
                if states < 5 and wrote_exec:
                    print(f"hazard: '{t}' writes EXEC {states} wait state(s) before '{text}'")
                    bad += 1
                states += took
                if states >= 5:
                    break
        took = 1
        if op == "s_nop":
            took = int(ops[0], 0) + 1
        is_valu = op.startswith("v_")
        wrote = regs(ops[0]) if is_valu and ops else set()
        wrote_exec = is_valu and (op.startswith("v_cmpx") or (ops and ops[0] == "exec"))
        window.append((took, wrote, wrote_exec, text))
        if len(window) > 8:
            window.pop(0)
    print(f"{checked} DPP instructions checked, {bad} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
