"""Static checks of the shipped ISA for hazards the compiler cannot see behind inline assembly.

* tools/check_dpp_hazard.py: the DPP read hazard around the v_fmac_f64_dpp / v_mov_b64_dpp of the
  register epilogues (csrc/sweep_kernels.hpp) -- VALU write -> DPP read, VALU write of EXEC -> DPP,
  matrix-pipe write -> DPP read -- walked backwards over the control-flow graph (branch targets too).
* tools/check_vmem_hazard.py: the hand-issued vector-memory operations (the profile gathers of
  csrc/sweep_multi_slim_kernel.hpp, the global -> LDS copies of csrc/sweep_kernels.hpp): no
  destination register touched before a covering `s_waitcnt vmcnt(n)`, nothing outstanding at
  s_endpgm, every hand-written vmcnt literal exactly what the instruction stream needs.

Both run on a build of the shipped source (hipcc cross-compiles without a GPU) and on hand-made
positive and negative cases."""
import glob
import importlib.util
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_hipcc = pytest.mark.skipif(shutil.which("hipcc") is None, reason="no hipcc: the ISA of the shipped source cannot be produced")


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def shipped_isa(tmp_path_factory):
    """The gfx950 ISA dump of csrc/gpdla.hip as __graft_entry__.build() compiles it."""
    sys.path.insert(0, ROOT)
    from gp_dla_detection_amd import _lib
    tmp = tmp_path_factory.mktemp("isa")
    out = tmp / "libgpdla_isa.so"
    subprocess.check_call(["hipcc", *_lib.HIPCC_FLAGS, "-save-temps=obj", os.path.join(_lib.CSRC, "gpdla.hip"), "-o", str(out)],
                          cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dumps = glob.glob(str(tmp / "*gfx950*.s"))
    assert len(dumps) == 1, dumps
    return dumps[0]


DPP = "\tv_fmac_f64_dpp v[4:5], -v[0:1], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"


def test_dpp_checker_sees_a_hazard_and_accepts_padding(tmp_path, capsys):
    chk = _tool("check_dpp_hazard")
    bad = tmp_path / "bad.s"
    bad.write_text("\tv_mov_b64_e32 v[0:1], v[8:9]\n\ts_nop 0\n" + DPP)
    assert chk.main(str(bad)) == 1
    ok = tmp_path / "ok.s"
    ok.write_text("\tv_mul_f64 v[0:1], v[2:3], v[2:3]\n\ts_nop 1\n" + DPP + "\tds_read_b64 v[0:1], v9\n\ts_waitcnt lgkmcnt(0)\n" + DPP)
    assert chk.main(str(ok)) == 0
    cmpx = tmp_path / "cmpx.s"
    cmpx.write_text("\tv_cmpx_lt_i32_e32 v1, v2\n\ts_nop 2\n" + DPP)
    assert chk.main(str(cmpx)) == 1
    # a hazard that arrives through a JUMP: the writer sits in front of the branch, the DPP behind its target
    jump = tmp_path / "jump.s"
    jump.write_text("\tv_mov_b64_e32 v[0:1], v[8:9]\n\ts_branch .LBB0_2\n.LBB0_1:\n\ts_nop 7\n.LBB0_2:\n" + DPP)
    assert chk.main(str(jump)) == 1
    far = tmp_path / "far.s"
    far.write_text("\tv_mov_b64_e32 v[0:1], v[8:9]\n\ts_nop 0\n\ts_branch .LBB0_2\n.LBB0_1:\n\ts_nop 7\n.LBB0_2:\n" + DPP)
    assert chk.main(str(far)) == 0
    # the matrix pipe's result read by a DPP instruction too early, and late enough
    mfma = "\tv_mfma_f64_16x16x4_f64 v[0:7], v[20:21], v[22:23], v[0:7]\n"
    early = tmp_path / "mfma_early.s"
    early.write_text(mfma + "\ts_nop 7\n" + DPP)
    assert chk.main(str(early)) == 1
    late = tmp_path / "mfma_late.s"
    late.write_text(mfma + "\ts_nop 7\n\ts_nop 7\n\ts_nop 2\n" + DPP)
    assert chk.main(str(late)) == 0
    capsys.readouterr()


GATHER = "\t;;#ASMSTART\n\tglobal_load_dwordx2 v[10:11], v[10:11], off\n\t;;#ASMEND\n"
OTHER = "\t;;#ASMSTART\n\tglobal_load_dwordx2 v[12:13], v[12:13], off\n\t;;#ASMEND\n"
DMA = "\t;;#ASMSTART\n\ts_mov_b32 m0, s4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[20:21], off\n\t;;#ASMEND\n"


def _asm_wait(n):
    return f"\t;;#ASMSTART\n\ts_waitcnt vmcnt({n})\n\t;;#ASMEND\n"


def _kernel(body):
    return "k_case:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"


def test_vmem_checker_on_hand_made_cases(tmp_path, capsys):
    chk = _tool("check_vmem_hazard")

    def run(name, body):
        path = tmp_path / (name + ".s")
        path.write_text(_kernel(body))
        return chk.main(str(path))

    use = "\tv_mul_f64 v[2:3], v[10:11], v[10:11]\n"
    # the exact wait: one younger operation, vmcnt(1)
    assert run("ok", GATHER + OTHER + _asm_wait(1) + use + "\ts_waitcnt vmcnt(0)\n") == 0
    # positive 1: the destination is read behind a wait that does not cover the load (two younger operations, vmcnt(2)... and none)
    assert run("uncovered", GATHER + OTHER + OTHER.replace("12:13", "14:15") + _asm_wait(3) + use + "\ts_waitcnt vmcnt(0)\n") == 1
    assert run("no_wait", GATHER + use + "\ts_waitcnt vmcnt(0)\n") == 1
    # ... or re-used by the compiler (written) while the load is in flight: the round-4 race
    assert run("reused", GATHER + "\tv_mov_b64_e32 v[10:11], 0\n\ts_waitcnt vmcnt(0)\n") == 1
    # positive 2: still outstanding at s_endpgm -- a register load and an LDS-DMA copy
    assert run("undrained", GATHER + "\tv_mov_b32_e32 v1, 0\n") == 1
    assert run("dma_undrained", DMA + "\tv_mov_b32_e32 v1, 0\n") == 1
    assert run("dma_drained", DMA + "\ts_waitcnt vmcnt(0)\n") == 0
    # positive 3: a hand-written literal stricter than the instruction stream needs (two younger operations, vmcnt(1))
    # (the two younger operations are the compiler's own loads: the only hand-issued load the wait covers has two behind it)
    theirs = "\tglobal_load_dwordx2 v[12:13], v[30:31], off\n\tglobal_load_dwordx2 v[14:15], v[30:31], off offset:8\n"
    assert run("too_strict", GATHER + theirs + _asm_wait(1) + use + "\ts_waitcnt vmcnt(0)\n") == 1
    assert run("exact", GATHER + theirs + _asm_wait(2) + use + "\ts_waitcnt vmcnt(0)\n") == 0
    # the guard idiom of an unrolled loop: the flag kept in an SGPR pair decides BOTH branches, so "body skipped,
    # loop continued" is not a path (with the second branch unknown the consumer would be flagged)
    loop = ("\ts_mov_b32 s9, 0\n.LBB0_1:\n" + GATHER + OTHER + _asm_wait(1) + use +
            "\ts_add_i32 s9, s9, 1\n\ts_cmp_lt_i32 s9, s8\n\ts_cselect_b64 s[2:3], -1, 0\n\ts_cmp_ge_i32 s9, s8\n"
            "\ts_cbranch_scc1 .LBB0_2\n" + OTHER.replace("12:13", "14:15") + "\ts_waitcnt vmcnt(0)\n.LBB0_2:\n"
            "\ts_andn2_b64 vcc, exec, s[2:3]\n\ts_cbranch_vccnz .LBB0_3\n\ts_branch .LBB0_1\n.LBB0_3:\n\ts_waitcnt vmcnt(0)\n")
    assert run("guard_idiom", loop) == 0
    capsys.readouterr()


@needs_hipcc
def test_shipped_kernels_have_no_dpp_hazard(shipped_isa, capsys):
    text = open(shipped_isa).read()
    assert text.count("v_fmac_f64_dpp") > 1000 and "v_mov_b64_dpp" in text  # (the check below is not vacuous)
    assert _tool("check_dpp_hazard").main(shipped_isa) == 0, capsys.readouterr().out
    capsys.readouterr()
    # ... and none of the sweep kernels may start keeping things in scratch: a private array of 224
    # bytes per lane in the k <= 40 epilogue once moved 18 GB of HBM traffic per launch (DESIGN.md
    # section 4); what they have today is a handful of prologue spills
    sizes = {m.group(1): int(m.group(2)) for m in re.finditer(
        r"\.name:\s+(\S+)\n(?:(?!\.name:).)*?\.private_segment_fixed_size:\s+(\d+)", text, re.S)}
    sweeps = {k: v for k, v in sizes.items() if "k_sweep" in k}
    assert len(sweeps) >= 10, sorted(sizes)
    assert max(sweeps.values()) <= 64, {k: v for k, v in sweeps.items() if v > 64}
    # the headline kernel (three lines, k <= 20) keeps everything in registers since the K-step's two
    # reciprocals became one (round 5): no scratch at all
    headline = [v for k, v in sweeps.items() if "k_sweep_slimILi3" in k]
    assert headline == [0], headline


@needs_hipcc
def test_shipped_hand_issued_memory_operations(shipped_isa, capsys):
    """Every inline-assembly global_load / global_load_lds of the shipped build: destination registers
    untouched until a covering wait, nothing outstanding at s_endpgm, hand-written vmcnt literals exact."""
    chk = _tool("check_vmem_hazard")
    kernels = chk.parse_kernels(shipped_isa)
    gathers = {k: [i for i in v[0] if i.asm and i.vmem and i.dest] for k, v in kernels.items()}
    with_gathers = {k for k, v in gathers.items() if v}
    # (not vacuous: the four k_sweep_multi_slim<ND> carry 4 (priming) + 8 (K-steps) gathers of ND loads each,
    # and the LDS-DMA copies of every sweep / training kernel are followed as well)
    assert sorted(len(gathers[k]) for k in with_gathers) == [12, 24, 36, 48], {k: len(v) for k, v in gathers.items() if v}
    assert all("k_sweep_multi_slim" in k for k in with_gathers)
    assert sum(1 for v in kernels.values() for i in v[0] if i.asm and i.vmem) > 300
    # the hand-written waits of those kernels, and what the instruction stream needs at each (check 3)
    for name in with_gathers:
        nd = int(re.search(r"k_sweep_multi_slimILi(\d)E", name).group(1))
        n, bad, waits = chk.check_kernel(name, *kernels[name], print)
        assert bad == 0, capsys.readouterr().out
        asm_waits = [(text, lo) for text, asm, lo, hi in waits.values() if asm]
        k_step = [lo for text, lo in asm_waits if f"vmcnt({3 * nd})" in text]
        assert len(k_step) == 8 and set(k_step) == {3 * nd}, (name, asm_waits)   # sweep_multi_slim_kernel.hpp: (kAhead - 1) ND
        assert any("vmcnt(0)" in text for text, lo in asm_waits), (name, asm_waits)  # the drain in front of the epilogue
    assert chk.main(shipped_isa) == 0, capsys.readouterr().out
    capsys.readouterr()
