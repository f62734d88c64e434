#!/usr/bin/env python3
"""Static check of HAND-ISSUED vector-memory operations in a gfx950 ISA dump (hipcc -save-temps, *.s).

The kernels issue some loads as inline assembly and wait for them with hand-counted
`s_waitcnt vmcnt(n)` (csrc/sweep_multi_slim_kernel.hpp: the profile gathers; csrc/sweep_kernels.hpp:
the global -> LDS copies of glds16 / glds_quad).  The compiler's wait-count pass does not see such
loads: nothing stops it from re-using a destination register while the load is in flight, and a
miscounted wait reads a stale value without failing any test.  This checker follows every path of a
kernel's control-flow graph from each inline-assembly VMEM instruction and asserts:

 1. registers  -- a VGPR written by an inline-assembly `global_load_*` is neither read nor written
                  by ANY instruction until an `s_waitcnt vmcnt(n)` with n <= (number of VMEM operations
                  issued after the load on that path) has been passed (vmcnt counts loads, stores,
                  atomics and LDS-DMA together, in issue order: MI355X_MICROARCH.md);
 2. drain      -- no inline-assembly VMEM operation (register load or LDS-DMA) is still outstanding
                  at `s_endpgm` on any path (an LDS-DMA landing after the workgroup's LDS has been
                  handed to another workgroup, or a load landing in a released register);
 3. literals   -- for every inline-assembly `s_waitcnt vmcnt(n)`: over the inline-assembly loads it
                  is the FIRST covering wait of, the smallest number of younger VMEM operations on
                  any path equals n -- the literal is exactly what the instruction stream needs, not
                  looser (check 1 would also fire at the consumer) and not stricter than the tightest
                  path.  (Waits issued through __builtin_amdgcn_s_waitcnt are reported with -v, not
                  held to exactness: a `vmcnt(0)` drain is allowed to be stricter.)

Paths are followed through every branch, with ONE piece of path sensitivity: a uniform compare whose
result the compiler keeps in an SGPR pair and tests twice --
    s_cmp_lt_i32 sA, sB ; s_cselect_b64 s[F], -1, 0 ; s_cmp_ge_i32 sA, sB ; s_cbranch_scc1 SKIP ; <body> ;
    SKIP: s_andn2_b64 vcc, exec, s[F] ; s_cbranch_vccnz EXIT
(how hipcc lowers `if (rn >= steps) goto done;` in an unrolled loop) -- is remembered along the path
while none of sA, sB, s[F] is rewritten, and an edge that contradicts it (body skipped AND the loop
continued) is not followed.  Anything the tracker does not recognise is followed both ways, so a
new compiler idiom can only make the check stricter, never blind.

    python tools/check_vmem_hazard.py [-v] <dump.s> [kernel-name substring ...]
Exit status 1 and a listing if a violation is found.
"""
import re
import sys
from collections import defaultdict

VMEM_PREFIXES = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic",
                 "scratch_load", "scratch_store", "flat_load", "flat_store", "flat_atomic")
CAP = 70            # younger-operation counts are saturated here (vmcnt has 6 bits)
MAX_STATES = 4_000_000
MAX_TRACK_STATES = 400_000   # per hand-issued operation, once it is in flight
MAX_REPORTS = 12             # violations listed per hand-issued operation
SPECIAL = {"vcc": (106, 107), "vcc_lo": (106,), "vcc_hi": (107,), "exec": (126, 127), "exec_lo": (126,),
           "exec_hi": (127,), "m0": (124,)}
VCC = frozenset(SPECIAL["vcc"])
CMP = {"lt": ("lt", False), "ge": ("lt", True), "gt": ("gt", False), "le": ("gt", True), "eq": ("eq", False),
       "lg": ("eq", True)}
SCC_SAFE = ("s_mov", "s_cmov", "s_cselect", "s_cbranch", "s_branch", "s_nop", "s_waitcnt", "s_barrier", "s_load",
            "s_sleep", "s_setprio", "s_endpgm", "s_getreg", "s_setreg", "s_memtime", "s_sendmsg", "s_version")
NO_SDST = ("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_bitcmp", "global_store", "buffer_store",
           "scratch_store", "flat_store", "ds_write", "s_barrier", "s_setprio", "s_endpgm", "s_sleep")


def vgprs(text):
    """Every VGPR index mentioned by an instruction's operands (v7, v[4:5]; not a[..] / s[..])."""
    out = set()
    for m in re.finditer(r"(?<![\w.])v\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w.\[])v(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def sgprs(operand):
    """SGPR indices named by ONE operand (s7, s[4:5], vcc, exec, m0; anything else: empty)."""
    operand = operand.strip()
    if operand in SPECIAL:
        return frozenset(SPECIAL[operand])
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", operand)
    return frozenset({int(m.group(1))}) if m else frozenset()


class Inst:
    __slots__ = ("op", "text", "regs", "asm", "line", "vmem", "wait", "target", "kind", "dest", "ops", "swrite",
                 "kills_scc", "relevant")

    def __init__(self, op, text, asm, line):
        self.op, self.text, self.asm, self.line = op, text, asm, line
        operands = text[len(op):]
        self.regs = vgprs(operands)
        self.ops = [o.strip() for o in operands.split(",")] if operands.strip() else []
        self.vmem = op.startswith(VMEM_PREFIXES)
        m = re.search(r"vmcnt\((\d+)\)", text) if op == "s_waitcnt" else None
        self.wait = int(m.group(1)) if m else None
        self.target = None
        self.kind = "fall"
        if op == "s_branch":
            self.kind = "jump"
        elif op.startswith("s_cbranch"):
            self.kind = "cond"
        elif op == "s_endpgm":
            self.kind = "end"
        elif op in ("s_setpc_b64", "s_swappc_b64"):
            self.kind = "indirect"
        if self.kind in ("jump", "cond"):
            self.target = self.ops[-1]
        # destination of a register load: the first operand (LDS-DMA has none)
        self.dest = set()
        if self.vmem and "_load" in op and "_lds_" not in op and " lds" not in text:
            self.dest = vgprs(self.ops[0]) if self.ops else set()
        # SGPRs this instruction writes: the first operand of anything but compares, branches and stores
        self.swrite = frozenset()
        if self.ops and not op.startswith(NO_SDST):
            self.swrite = sgprs(self.ops[0])
        if op.startswith("v_cmp") and op.endswith("_e32"):
            self.swrite = self.swrite | VCC  # VOPC writes vcc implicitly
        if op.startswith("v_cmpx"):
            self.swrite = self.swrite | frozenset(SPECIAL["exec"])
        self.kills_scc = op.startswith("s_") and not op.startswith(SCC_SAFE)
        self.relevant = bool(self.swrite) or self.kills_scc


def parse_kernels(path, wanted=()):
    """{kernel name: (instructions, {label: index})} for every function in the dump."""
    kernels = {}
    name, insts, labels, in_asm = None, None, None, False
    for lineno, raw in enumerate(open(path), 1):
        line = raw.rstrip("\n")
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        text = line.split(";")[0].strip()
        if not text:
            continue
        m = re.fullmatch(r"([A-Za-z_.$][\w$.]*):", text)
        if m:
            label = m.group(1)
            if label.startswith(".Lfunc_end"):
                if name is not None:
                    kernels[name] = (insts, labels)
                name, insts, labels = None, None, None
            elif label.startswith("."):
                if insts is not None:
                    labels[label] = len(insts)
            else:
                name, insts, labels = label, [], {}
            continue
        if insts is None or text.startswith("."):
            continue
        op = text.split(None, 1)[0]
        insts.append(Inst(op, text, in_asm, lineno))
    if name is not None:  # (a dump cut without .Lfunc_end)
        kernels[name] = (insts, labels)
    if wanted:
        kernels = {k: v for k, v in kernels.items() if any(w in k for w in wanted)}
    return kernels


def successors(insts, labels, i):
    ins = insts[i]
    if ins.kind == "end":
        return []
    if ins.kind == "jump":
        return [labels[ins.target]]
    if ins.kind == "cond":
        return [labels[ins.target], i + 1]
    if ins.kind == "indirect":
        raise ValueError(f"line {ins.line}: indirect branch '{ins.text}' (calls are expected to be inlined)")
    return [i + 1] if i + 1 < len(insts) else []


# ---- the path's knowledge of uniform compares -------------------------------------------------
# facts = (scc, bind, known, valid)
#   pred  = (op, a, b, site): the compare `op a, b` as evaluated at instruction index `site`; a later
#           compare of the same operands is the SAME predicate while neither operand has been rewritten
#           (`valid` holds the predicates for which that is still so) -- rewriting an operand AFTER the
#           compare (hipcc does: s_cmp_ge s24, s51 ; s_mov_b32 s24, 20 ; s_cbranch_scc1) does not change
#           what scc and the flag registers already hold
#   scc   = (pred, negated) or None;  bind = frozenset of (sgpr set, (pred, negated))
#   known = frozenset of (pred, truth): decided by a branch earlier on the path AND still held in a
#           bound SGPR pair (a flag the compiler will test again)
#   consts = frozenset of (sgpr index, value): SGPRs holding an immediate (s_mov_b32 sN, imm).  hipcc
#           turns a jump out of two nested loops into a state code -- s_mov_b32 s24, 20 on the way out,
#           s_cmp_lg_u32 s24, 17 / s24, 0 in a dispatch block that also holds the loop's back edge --
#           and a compare of a known code with an immediate decides its branch outright.
EMPTY = (None, frozenset(), frozenset(), frozenset(), frozenset())


def facts_step(facts, ins, index):
    """Effect of a non-branch instruction on the facts."""
    scc, bind, known, valid, consts = facts
    op, ops = ins.op, ins.ops
    if ins.swrite:
        w = ins.swrite
        bind = frozenset((k, v) for k, v in bind if not (k & w))
        valid = frozenset(p for p in valid if not ((sgprs(p[1]) | sgprs(p[2])) & w))
        consts = frozenset(c for c in consts if c[0] not in w)
        if op == "s_mov_b32" and len(ops) == 2 and len(w) == 1 and re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", ops[1]):
            consts = consts | {(next(iter(w)), int(ops[1], 0))}
    m = re.fullmatch(r"s_cmp_(lt|ge|gt|le|eq|lg)_(i32|u32|u64)", op)
    if m and len(ops) == 2:
        base, neg = CMP[m.group(1)]
        name = base + "_" + m.group(2)
        vals = []
        for o in ops:
            r = sgprs(o)
            c = dict(consts).get(next(iter(r))) if len(r) == 1 else None
            vals.append(c if c is not None else (int(o, 0) if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", o) else None))
        if vals[0] is not None and vals[1] is not None:  # both sides known: the compare is decided
            truth = {"lt": vals[0] < vals[1], "gt": vals[0] > vals[1], "eq": vals[0] == vals[1]}[base]
            scc = (("const", truth), neg)
        else:
            pred = next((p for p in valid if p[:3] == (name, ops[0], ops[1])), None)
            if pred is None:
                pred = (name, ops[0], ops[1], index)
                valid = valid | {pred}
            scc = (pred, neg)
    elif ins.kills_scc:
        scc = None
    if op == "s_cselect_b64" and len(ops) == 3 and scc is not None and ops[1:] in (["-1", "0"], ["0", "-1"]):
        bind = bind | {(sgprs(ops[0]), (scc[0], scc[1] ^ (ops[1] == "0")))}
    elif op in ("s_andn2_b64", "s_and_b64") and len(ops) == 3 and ops[1] == "exec":
        src = dict(bind).get(sgprs(ops[2]))
        if src is not None:  # nonzero iff the predicate (exec is not empty where a wave branches on it)
            bind = bind | {(sgprs(ops[0]), (src[0], src[1] ^ (op == "s_andn2_b64")))}
    live = {v[0] for _, v in bind}
    known = frozenset(kv for kv in known if kv[0] in live)
    valid = frozenset(p for p in valid if p in live or (scc is not None and p == scc[0]))
    return (scc, bind, known, valid, consts)


def facts_edge(facts, ins, taken):
    """Facts on the taken / fall-through edge of a conditional branch; None if the edge contradicts them."""
    scc, bind, known, valid, consts = facts
    sym = None
    if ins.op in ("s_cbranch_scc1", "s_cbranch_scc0") and scc is not None:
        sym = (scc[0], scc[1] ^ (ins.op == "s_cbranch_scc0"))  # (pred, negated) that is TRUE on the taken edge
    elif ins.op in ("s_cbranch_vccnz", "s_cbranch_vccz"):
        v = dict(bind).get(VCC)
        if v is not None:
            sym = (v[0], v[1] ^ (ins.op == "s_cbranch_vccz"))
    if sym is None:
        return facts
    pred, neg = sym
    value = (not neg) if taken else neg  # the predicate's truth on this edge
    if pred[0] == "const":
        return facts if pred[1] == value else None
    for p, t in known:
        if p == pred:
            return facts if t == value else None
    if pred in {v[0] for _, v in bind}:  # remembered only while a flag register still holds it
        known = known | {(pred, value)}
    return (scc, bind, known, valid, consts)


def check_kernel(name, insts, labels, report):
    """Returns (number of inline-assembly VMEM operations followed, violations, {wait line: summary})."""
    bad = 0
    hand = [i for i, ins in enumerate(insts) if ins.asm and ins.vmem]
    coded = set()  # SGPRs some compare tests against an immediate: the only ones tracked as constants
    for ins in insts:
        if ins.op.startswith("s_cmp") and len(ins.ops) == 2 and re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", ins.ops[1]):
            coded |= sgprs(ins.ops[0])
    first_cover = defaultdict(list)  # index of an s_waitcnt -> [younger counts of the loads it first covers]

    def advance(i, facts):
        """(successor index, facts) pairs of instruction i reached with `facts`."""
        ins = insts[i]
        if ins.kind == "cond":
            out = []
            for taken, s in ((True, labels[ins.target]), (False, i + 1)):
                f = facts_edge(facts, ins, taken)
                if f is not None:
                    out.append((s, f))
            return out
        nf = facts_step(facts, ins, i) if (ins.relevant or ins.op.startswith("s_cmp")) else facts
        if nf[4] and not all(c[0] in coded for c in nf[4]):
            nf = nf[:4] + (frozenset(c for c in nf[4] if c[0] in coded),)
        return [(s, nf) for s in successors(insts, labels, i)]

    # what a path from the kernel's entry knows when it reaches each hand-issued operation (the state
    # code of a loop it sits in was set BEFORE the operation)
    reach = defaultdict(set)
    hand_set = set(hand)
    if not any(insts[i].dest for i in hand):  # LDS-DMA only (the drain check): no context needed
        for i in hand:
            reach[i].add(EMPTY)
    else:
        # (only instructions from which a hand-issued operation can still be reached are visited)
        preds = defaultdict(list)
        for i in range(len(insts)):
            for t in successors(insts, labels, i):
                preds[t].append(i)
        useful, work = set(hand), list(hand)
        while work:
            for q in preds[work.pop()]:
                if q not in useful:
                    useful.add(q)
                    work.append(q)
        seen0, stack0 = set(), [(0, EMPTY)]
        while stack0:
            st = stack0.pop()
            if st in seen0 or st[0] not in useful:
                continue
            seen0.add(st)
            if len(seen0) > MAX_STATES:
                raise RuntimeError(f"{name}: more than {MAX_STATES} states from the entry")
            if st[0] in hand_set:
                reach[st[0]].add(st[1])
            stack0.extend(advance(*st))
    for start in hand:
        load = insts[start]
        seen = set()
        stack = [(s, 0, f) for f0 in reach[start] for s, f in advance(start, f0)]
        flagged = set()
        while stack:
            state = stack.pop()
            if state in seen:
                continue
            seen.add(state)
            if len(flagged) >= MAX_REPORTS or len(seen) > MAX_TRACK_STATES:
                if len(seen) > MAX_TRACK_STATES:  # fails closed: an operation nothing waits for wanders through the whole kernel
                    bad += 1
                    report(f"{name}: gave up after {MAX_TRACK_STATES} states behind the inline-assembly '{load.text}' of line "
                           f"{load.line}: no covering wait close by")
                break
            i, y, facts = state
            ins = insts[i]
            if ins.wait is not None and ins.wait <= y:
                first_cover[i].append(y)
                continue  # the load has completed on this path
            if load.dest & ins.regs and i not in flagged:
                flagged.add(i)
                bad += 1
                report(f"{name}: line {ins.line} '{ins.text}' touches v{sorted(load.dest & ins.regs)} while the inline-assembly "
                       f"load of line {load.line} '{load.text}' may be in flight ({y} younger VMEM operation(s), no covering wait)")
            if ins.kind == "end":
                if "end" not in flagged:
                    flagged.add("end")
                    bad += 1
                    report(f"{name}: the inline-assembly '{load.text}' of line {load.line} can still be outstanding at "
                           f"s_endpgm (line {ins.line})")
                continue
            ny = min(CAP, y + 1) if ins.vmem else y
            for s, f in advance(i, facts):
                stack.append((s, ny, f))
    for i, counts in sorted(first_cover.items()):
        ins = insts[i]
        if ins.asm and min(counts) != ins.wait:
            bad += 1
            report(f"{name}: line {ins.line} inline-assembly '{ins.text}': the tightest path has {min(counts)} younger VMEM "
                   f"operation(s) behind the load it waits for, the literal says {ins.wait}")
    return len(hand), bad, {insts[i].line: (insts[i].text, insts[i].asm, min(c), max(c)) for i, c in first_cover.items()}


def main(path, wanted=(), verbose=False):
    kernels = parse_kernels(path, wanted)
    total = bad = 0
    for name, (insts, labels) in sorted(kernels.items()):
        n, b, waits = check_kernel(name, insts, labels, print)
        total += n
        bad += b
        if verbose and n:
            print(f"{name}: {n} inline-assembly VMEM operation(s) followed")
            for line, (text, asm, lo, hi) in sorted(waits.items()):
                print(f"    line {line}: {text:32s} {'asm' if asm else 'builtin/compiler'}: first covering wait, younger operations {lo}..{hi}")
    print(f"{len(kernels)} kernel(s), {total} inline-assembly VMEM operation(s) followed, {bad} violation(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-v"]
    sys.exit(main(args[0], tuple(args[1:]), verbose="-v" in sys.argv))
