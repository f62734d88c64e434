"""The reference-side MEX gateways of integration/ RUN on the GPU: compiled against tests/mex_mock (a
small implementation of the documented mex.h / matrix.h calls they use -- no MATLAB exists in the
image), handed MATLAB-shaped arguments (column-major arrays, cells, structs) and compared, variable by
variable and in MATLAB's shapes, with the Python surface over the same C-ABI.  What this pins is the
gateways' own code: the model priors they restate in C (process_qsos.m:122-131, multi :189-216), the
cell hand-over, and the row-major -> column-major transposes of every table
(process_qsos.m:88-233; multi_dlas/process_qsos_multiple_dlas_meanflux.m:141-495)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import gp_dla_detection_amd as gp
from gp_dla_detection_amd import _lib, synthetic
from gp_dla_detection_amd.parameters import MultiParameters, Parameters

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VP = C.c_void_p


def build_gateway(tmp_path, source):
    out = tmp_path / (os.path.splitext(source)[0] + ".so")
    hip_rt = os.path.dirname(_lib._preload_hip_runtime()._name)
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared",
                    "-I", os.path.join(ROOT, "tests", "mex_mock"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "mex_mock", "mex_mock.c"), os.path.join(ROOT, "integration", source),
                    "-o", str(out), _lib.DEFAULT_LIB_PATH, "-lm", f"-Wl,-rpath,{os.path.dirname(_lib.DEFAULT_LIB_PATH)}",
                    f"-Wl,-rpath,{hip_rt}"], check=True, capture_output=True)
    _lib.load()  # (the HIP runtime and libgpdla.so are in the process before the gateway binds to them)
    lib = C.CDLL(str(out))
    for name, res, args in (("mock_doubles", VP, [VP, C.c_size_t, C.c_size_t, C.c_size_t]),
                            ("mock_logicals", VP, [VP, C.c_size_t, C.c_size_t]),
                            ("mock_uint32s", VP, [VP, C.c_size_t, C.c_size_t, C.c_size_t]),
                            ("mock_cell", VP, [C.c_size_t]), ("mock_set_cell", None, [VP, C.c_size_t, VP]),
                            ("mock_struct", VP, [C.c_int, C.POINTER(C.c_char_p)]), ("mock_set_field", None, [VP, C.c_int, VP]),
                            ("mock_ndim", C.c_int, [VP]), ("mock_dim", C.c_size_t, [VP, C.c_int]), ("mock_kind", C.c_int, [VP]),
                            ("mock_message", C.c_char_p, []), ("mock_call", C.c_int, [C.c_int, C.POINTER(VP), C.POINTER(VP)]),
                            ("mxGetField", VP, [VP, C.c_size_t, C.c_char_p]), ("mxGetData", VP, [VP]),
                            ("mxDestroyArray", None, [VP])):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


class Mock:
    """Builds MATLAB-shaped arguments in the mock runtime and reads results back (column-major)."""

    def __init__(self, lib):
        self.lib = lib

    def doubles(self, a, shape=None):
        a = np.asarray(a, dtype=np.float64)
        shape = tuple(shape or (a.shape if a.ndim > 1 else (a.size, 1)))
        f = np.asfortranarray(a.reshape(shape))
        d = shape + (1,) * (3 - len(shape))
        return self.lib.mock_doubles(f.ctypes.data_as(VP), *d)

    def logicals(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        return self.lib.mock_logicals(a.ctypes.data_as(VP), a.size, 1)

    def uint32s(self, a):
        f = np.asfortranarray(a, dtype=np.uint32)
        return self.lib.mock_uint32s(f.ctypes.data_as(VP), *f.shape)

    def cell(self, items):
        c = self.lib.mock_cell(len(items))
        for i, it in enumerate(items):
            self.lib.mock_set_cell(c, i, it)
        return c

    def struct(self, **fields):
        names = (C.c_char_p * len(fields))(*[k.encode() for k in fields])
        s = self.lib.mock_struct(len(fields), names)
        for i, v in enumerate(fields.values()):
            self.lib.mock_set_field(s, i, v)
        return s

    def call(self, *args):
        prhs = (VP * len(args))(*args)
        out = VP()
        rc = self.lib.mock_call(len(args), prhs, C.byref(out))
        return rc, out, self.lib.mock_message().decode()

    def field(self, s, name):
        """Field `name` of result struct `s` as a NumPy array in MATLAB's shape."""
        f = self.lib.mxGetField(s, 0, name.encode())
        assert f, name
        dims = tuple(self.lib.mock_dim(f, i) for i in range(self.lib.mock_ndim(f)))
        dt = np.uint32 if self.lib.mock_kind(f) == 2 else np.float64
        n = int(np.prod(dims))
        buf = (C.c_uint32 if dt == np.uint32 else C.c_double) * max(n, 1)
        flat = np.frombuffer(buf.from_address(self.lib.mxGetData(f)), dtype=dt, count=n).copy()
        return flat.reshape(dims, order="F")


def matlab_inputs(mk, model, samples, spectra, cat, multi):
    m = mk.struct(rest_wavelengths=mk.doubles(model["rest_wavelengths"]), mu=mk.doubles(model["mu"]),
                  M=mk.doubles(model["M"]), log_omega=mk.doubles(model["log_omega"]),
                  log_c_0=mk.doubles([model["log_c_0"]]), log_tau_0=mk.doubles([model["log_tau_0"]]),
                  log_beta=mk.doubles([model["log_beta"]]))
    sf = dict(offset_samples=mk.doubles(samples["offset_samples"]), log_nhi_samples=mk.doubles(samples["log_nhi_samples"]),
              nhi_samples=mk.doubles(samples["nhi_samples"]))
    if multi:
        sf["lls_nhi_samples"] = mk.doubles(samples["lls_nhi_samples"])
    s = mk.struct(**sf)
    # every second mask as doubles: the gateway converts those, logical ones go as they are
    sp = mk.struct(wavelengths=mk.cell([mk.doubles(q["wavelengths"]) for q in spectra]),
                   flux=mk.cell([mk.doubles(q["flux"]) for q in spectra]),
                   noise_variance=mk.cell([mk.doubles(q["noise_variance"]) for q in spectra]),
                   pixel_mask=mk.cell([mk.doubles(np.asarray(q["pixel_mask"], dtype=np.float64)) if i % 2
                                       else mk.logicals(q["pixel_mask"]) for i, q in enumerate(spectra)]),
                   z_qsos=mk.doubles([q["z_qso"] for q in spectra]))
    pr = mk.struct(z_qsos=mk.doubles(cat["z_qsos"]), dla_ind=mk.logicals(cat["dla_ind"]))
    return m, s, sp, pr


def test_process_qsos_gateway_returns_what_the_python_surface_returns(tmp_path):
    lib = build_gateway(tmp_path, "process_qsos_gpdla_mex.c")
    mk = Mock(lib)
    model = synthetic.make_model(20)
    samples = synthetic.make_samples(40)
    spectra = [synthetic.make_spectrum(300 + i, n, model, mask_fraction=0.05) for i, n in enumerate([210, 64, 333, 120, 9])]
    cat = synthetic.make_prior_catalog()
    p = Parameters()
    want = gp.process_qsos(model, samples, spectra, prior_catalog=cat, params=p)
    rc, res, msg = mk.call(*matlab_inputs(mk, model, samples, spectra, cat, multi=False))
    assert rc == 0, msg
    nq, S = len(spectra), 40
    for name in ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla", "log_likelihoods_no_dla",
                 "log_likelihoods_dla", "log_posteriors_no_dla", "log_posteriors_dla", "p_no_dlas", "p_dlas"):
        got = mk.field(res, name)
        assert got.shape == (nq, 1), name                       # process_qsos.m:74-82: nan(num_quasars, 1)
        # (the priors are restated in C -- log of the same counts by another libm -- and the posteriors follow them)
        tol = 0.0 if name in ("min_z_dlas", "max_z_dlas", "log_likelihoods_no_dla", "log_likelihoods_dla") else 1e-14
        np.testing.assert_allclose(got[:, 0], want[name], rtol=0, atol=tol, equal_nan=True, err_msg=name)
    got = mk.field(res, "sample_log_likelihoods_dla")
    assert got.shape == (nq, S)                                  # sample_log_likelihoods_dla(quasar_ind, i), :196
    np.testing.assert_array_equal(got, want["sample_log_likelihoods_dla"])
    got = mk.field(res, "model_posteriors")
    assert got.shape == (nq, 2)
    np.testing.assert_allclose(got, want["model_posteriors"], rtol=0, atol=1e-15)
    lib.mxDestroyArray(res)
    # argument errors come back as MATLAB errors (mexErrMsgIdAndTxt), not crashes
    rc, _, msg = mk.call(mk.doubles([1.0]))
    assert rc == 1 and "usage" in msg


def test_multi_dla_gateway_returns_what_the_python_surface_returns(tmp_path):
    lib = build_gateway(tmp_path, "process_qsos_multi_gpdla_mex.c")
    mk = Mock(lib)
    p = MultiParameters(max_dlas=3)
    model = synthetic.make_model(20)
    samples = synthetic.make_samples(56)
    spectra = [synthetic.make_spectrum(400 + i, n, model, mask_fraction=0.04) for i, n in enumerate([180, 77, 260, 50])]
    spectra[3] = dict(wavelengths=np.zeros(0), flux=np.zeros(0), noise_variance=np.zeros(0),
                      pixel_mask=np.zeros(0, dtype=bool), z_qso=2.7)   # an empty spectrum: all_exceptions
    cat = synthetic.make_prior_catalog()
    Z_lls, Z_dla = 0.31, 0.69
    z = np.array([q["z_qso"] for q in spectra])
    lp = gp.dla_existence_prior_multi(cat["z_qsos"], cat["dla_ind"], z, Z_lls, Z_dla, p)
    want = gp.process_qsos_multiple_dlas_meanflux(model, samples, spectra, lp, params=p)
    nq, S, md = len(spectra), 56, 3
    m, s, sp, pr = matlab_inputs(mk, model, samples, spectra, cat, multi=True)
    params = mk.struct(max_dlas=mk.doubles([3.0]))
    # replay the indices the Python run drew (the variable the script saves, :476, [nq x S x (max_dlas - 1)])
    base = np.transpose(want["base_sample_inds"], (0, 2, 1))
    rc, res, msg = mk.call(m, s, sp, pr, mk.doubles([Z_lls]), mk.doubles([Z_dla]), params, mk.uint32s(base))
    assert rc == 0, msg
    for name in ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_lls", "log_likelihoods_no_dla",
                 "log_likelihoods_lls", "log_posteriors_no_dla", "log_posteriors_lls", "p_no_dlas", "p_dlas", "p_lls",
                 "all_exceptions"):
        got = mk.field(res, name)
        assert got.shape == (nq, 1), name                       # multi :104-139
        tol = 0.0 if name in ("min_z_dlas", "max_z_dlas", "log_likelihoods_no_dla", "log_likelihoods_lls", "all_exceptions") else 1e-14
        np.testing.assert_allclose(got[:, 0], want[name], rtol=0, atol=tol, equal_nan=True, err_msg=name)
    for name in ("log_priors_dla", "log_likelihoods_dla", "log_posteriors_dla"):
        got = mk.field(res, name)
        assert got.shape == (nq, md), name
        np.testing.assert_allclose(got, want[name], rtol=0, atol=1e-14 if name != "log_likelihoods_dla" else 0.0,
                                   equal_nan=True, err_msg=name)
    got = mk.field(res, "sample_log_likelihoods_dla")            # (quasar, sample, model), multi :110, :477
    assert got.shape == (nq, S, md)
    np.testing.assert_array_equal(got, np.transpose(want["sample_log_likelihoods_dla"], (0, 2, 1)))
    got = mk.field(res, "sample_log_likelihoods_lls")
    assert got.shape == (nq, S)
    np.testing.assert_array_equal(got, want["sample_log_likelihoods_lls"])
    got = mk.field(res, "base_sample_inds")
    assert got.shape == (nq, S, md - 1) and got.dtype == np.uint32
    np.testing.assert_array_equal(got, base)
    for name in ("MAP_z_dlas", "MAP_log_nhis", "MAP_inds"):       # (quasar, model, slot), multi :126-128, :439-445
        got = mk.field(res, name)
        assert got.shape == (nq, md, md), name
        np.testing.assert_array_equal(got, want[name], err_msg=name)
    got = mk.field(res, "model_posteriors")
    assert got.shape == (nq, 2 + md)
    np.testing.assert_allclose(got, want["model_posteriors"], rtol=0, atol=1e-14, equal_nan=True)
    assert mk.field(res, "all_exceptions")[3, 0] == 1.0 and np.isnan(mk.field(res, "p_dlas")[3, 0])
    lib.mxDestroyArray(res)
