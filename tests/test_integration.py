"""The reference-side bindings of INTEGRATION.md compile.  No MATLAB (and no mex.h) exists in the
build image, so the MEX gateway integration/process_qsos_gpdla_mex.c -- the replacement of the loop
process_qsos.m:88-233 -- is compiled (syntax + types, -fsyntax-only) against declarations of exactly
the documented mex.h / matrix.h calls it uses, written out here.  Compile-checking is the point:
this is not an oracle and runs nothing."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATEWAY = os.path.join(ROOT, "integration", "process_qsos_gpdla_mex.c")
GATEWAY_MULTI = os.path.join(ROOT, "integration", "process_qsos_multi_gpdla_mex.c")

# The MATLAB C Matrix / MEX API as documented (R2018a+ signatures; mwSize = size_t, mwIndex = size_t)
MEX_H = open(os.path.join(ROOT, "tests", "mex_mock", "mex.h")).read()


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_process_qsos_gateway_compiles_against_the_header(tmp_path):
    (tmp_path / "mex.h").write_text(MEX_H)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", str(tmp_path),
                        "-I", os.path.join(ROOT, "include"), GATEWAY], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and it really compiles to an object whose only undefined symbols are the MEX API and the C-ABI
    obj = tmp_path / "gateway.o"
    subprocess.run(["gcc", "-std=c99", "-c", "-fPIC", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"),
                    GATEWAY, "-o", str(obj)], check=True)
    undefined = set(re.findall(r"\bU (\w+)", subprocess.run(["nm", str(obj)], capture_output=True, text=True).stdout))
    declared = set(re.findall(r"\b(mx\w+|mex\w+)\(", MEX_H))
    ours = {u for u in undefined if u.startswith("gpdla_")}
    assert ours == {"gpdla_process_cells", "gpdla_default_config", "gpdla_last_error"}
    assert {u for u in undefined if u.startswith(("mx", "mex"))} <= declared
    assert undefined - ours - declared <= {"log", "memcpy", "memset", "_GLOBAL_OFFSET_TABLE_", "__stack_chk_fail"}


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_multi_dla_gateway_compiles_against_the_header(tmp_path):
    """integration/process_qsos_multi_gpdla_mex.c, the replacement of the loop
    multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-495: same check as the single-DLA gateway."""
    (tmp_path / "mex.h").write_text(MEX_H)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", str(tmp_path),
                        "-I", os.path.join(ROOT, "include"), GATEWAY_MULTI], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    obj = tmp_path / "gateway_multi.o"
    subprocess.run(["gcc", "-std=c99", "-c", "-fPIC", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"),
                    GATEWAY_MULTI, "-o", str(obj)], check=True)
    undefined = set(re.findall(r"\bU (\w+)", subprocess.run(["nm", str(obj)], capture_output=True, text=True).stdout))
    declared = set(re.findall(r"\b(mx\w+|mex\w+)\(", MEX_H))
    ours = {u for u in undefined if u.startswith("gpdla_")}
    assert ours == {"gpdla_process_cells_multi", "gpdla_default_config", "gpdla_last_error"}
    assert {u for u in undefined if u.startswith(("mx", "mex"))} <= declared
    assert undefined - ours - declared <= {"log", "pow", "memcpy", "memset", "_GLOBAL_OFFSET_TABLE_", "__stack_chk_fail"}


def test_multi_dla_gateway_returns_the_variables_the_script_saves():
    """multi :498-510: every per-quasar variable of variables_to_save is a field of the result struct
    (plus MAP_inds, which the script fills at :441 but does not save)."""
    src = open(GATEWAY_MULTI).read()
    fields = re.search(r"static const char \*fields\[\] = \{(.*?)\};", src, re.S).group(1)
    names = re.findall(r'"(\w+)"', fields)
    saved = ["min_z_dlas", "max_z_dlas", "sample_log_likelihoods_dla", "base_sample_inds", "log_priors_no_dla",
             "log_priors_dla", "log_priors_lls", "log_likelihoods_no_dla", "MAP_z_dlas", "MAP_log_nhis",
             "log_likelihoods_dla", "log_likelihoods_lls", "log_posteriors_no_dla", "log_posteriors_dla",
             "log_posteriors_lls", "model_posteriors", "p_no_dlas", "p_dlas", "p_lls", "all_exceptions",
             "sample_log_likelihoods_lls"]
    assert names == saved + ["MAP_inds"]
    from gp_dla_detection_amd import io
    assert set(saved) <= set(io.SAVED_VARIABLES_MULTI)


def test_gateway_returns_the_variables_the_script_saves():
    """process_qsos.m:239-244: the field list of the result struct is the script's own."""
    src = open(GATEWAY).read()
    fields = re.search(r"static const char \*fields\[\] = \{(.*?)\};", src, re.S).group(1)
    names = re.findall(r'"(\w+)"', fields)
    assert names == ["min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla", "log_likelihoods_no_dla",
                     "sample_log_likelihoods_dla", "log_likelihoods_dla", "log_posteriors_no_dla",
                     "log_posteriors_dla", "model_posteriors", "p_no_dlas", "p_dlas"]
    from gp_dla_detection_amd import io
    assert tuple(names) == io.SAVED_VARIABLES
