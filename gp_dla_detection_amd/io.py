"""Readers/writers for the .mat files either side of the hot path.

The reference hands data between stages through MATLAB files saved with ``-v7.3``, i.e. HDF5
behind a 512-byte MATLAB header: ``learned_qso_model_*.mat`` (learn_qso_model.m:113-123),
``dla_samples.mat`` (generate_dla_samples.m:59-63), ``preloaded_qsos.mat`` (preload_qsos.m:64-79),
``catalog.mat`` (build_catalogs.m:86-91) and the outputs ``processed_qsos_*.mat``
(process_qsos.m:236-250; multi_dlas/process_qsos_multiple_dlas_meanflux.m:498-523), which
``CDDF_analysis`` then opens with ``h5py.File`` and indexes as ``f['p_dlas'][0, :]``,
``f['model_posteriors'][()].T``, ``f['test_ind'][0, :]`` (qso_loader.py:84-112, calc_cddf.py:104).

* ``-v7.3`` files are read and written with the package's own minimal HDF5 implementation
  (:mod:`.hdf5`): the interpreter this package runs under has no h5py (libhdf5 and h5py exist in
  the build image's conda environment and are used to cross-validate it, tests/test_consumers.py).  MATLAB's
  conventions are applied on top of it here: arrays are stored with their dimensions reversed
  (column-major data in a row-major container), logicals as uint8 and chars as uint16 with the
  ``MATLAB_class`` / ``MATLAB_int_decode`` attributes, cell arrays as object references into the
  ``#refs#`` group, empty arrays through ``MATLAB_empty``.
* v5/v7 files (``save -v7``) go through ``scipy.io``.

All functions here speak MATLAB orientation (``sample_log_likelihoods_dla`` is [num_quasars x S],
multi-DLA: [num_quasars x S x max_dlas]) unless stated otherwise.
"""
from __future__ import annotations

import datetime
import os

import numpy as np

from . import hdf5

MATLAB_HEADER_TEXT = "MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: {date} HDF5 schema 1.00 ."


def _is_hdf5(path: str) -> bool:
    with open(path, "rb") as f:
        head = f.read(520)
    return b"MATLAB 7.3" in head[:128] or head[:8] == hdf5.SIGNATURE or head[512:520] == hdf5.SIGNATURE


# ---------------------------------------------------------------------------------------------
# -v7.3 <-> NumPy, MATLAB conventions
# ---------------------------------------------------------------------------------------------

def matlab_userblock(created: str | None = None) -> bytes:
    """The 512-byte block MATLAB puts in front of the HDF5 superblock: 116 bytes of text, 8 bytes
    of subsystem offset, version 0x0200 and the endian indicator 'IM'; zero-filled to 512."""
    date = created or datetime.datetime.now().strftime("%a %b %d %H:%M:%S %Y")
    text = MATLAB_HEADER_TEXT.format(date=date).encode("ascii")[:116].ljust(116, b" ")
    return (text + b"\x00" * 8 + b"\x00\x02" + b"IM").ljust(512, b"\x00")


def _from_dataset(f: hdf5.File, ds, deref_depth: int = 2):
    """One HDF5 object of a -v7.3 file as MATLAB holds it."""
    if isinstance(ds, hdf5.Group):  # struct: fields are the group's members
        return {k: _from_dataset(f, ds[k], deref_depth) for k in ds.keys()}
    cls = ds.attrs.get("MATLAB_class", "")
    if "MATLAB_empty" in ds.attrs:
        dims = tuple(int(x) for x in ds.read().ravel())
        return [] if cls == "cell" else np.zeros(dims, dtype=bool if cls == "logical" else np.float64)
    a = ds.read()
    if ds.is_reference:
        if deref_depth <= 0:
            return a.T
        flat = [_from_dataset(f, f.dereference(r), deref_depth - 1) for r in a.T.ravel(order="F")]
        return flat  # cell arrays come back as flat lists in MATLAB's (column-major) element order
    a = a.T  # undo the dimension reversal
    if cls == "logical":
        return a.astype(bool)
    if cls == "char":
        return "".join(chr(c) for c in a.ravel(order="F"))
    return a


def loadmat73(path: str, names=None) -> dict:
    """Variables of a ``-v7.3`` file in MATLAB orientation ({name: array | str | list (cell) |
    dict (struct)}).  ``names``: the variables wanted (default: all but ``#refs#``)."""
    out = {}
    with hdf5.File(path) as f:
        for n in (names if names is not None else [k for k in f.keys() if not k.startswith("#")]):
            if n in f:
                out[n] = _from_dataset(f, f[n])
    return out


class _MatWriter:
    """savemat for ``-v7.3``: values in MATLAB orientation."""

    def __init__(self, path: str, created: str | None = None):
        self.w = hdf5.FileWriter(path, userblock=matlab_userblock(created))
        self._refs = 0

    @staticmethod
    def _class_of(a: np.ndarray) -> str:
        if a.dtype.kind == "b":
            return "logical"
        if a.dtype.kind == "f":
            return "double" if a.dtype.itemsize == 8 else "single"
        return {"i": "int", "u": "uint"}[a.dtype.kind] + str(8 * a.dtype.itemsize)

    def _array(self, name: str, value, compress: bool):
        a = np.asarray(value)
        if a.dtype.kind == "O":
            raise TypeError(f"{name}: object arrays are written as cells; pass a list")
        if a.ndim < 2:  # MATLAB has no 0-d / 1-d arrays: scalars are 1x1, vectors columns
            a = a.reshape(-1, 1) if a.ndim == 1 else a.reshape(1, 1)
        attrs = {"MATLAB_class": self._class_of(a)}
        if a.dtype.kind == "b":
            attrs["MATLAB_int_decode"] = np.int32(1)
        if a.size == 0:
            attrs["MATLAB_empty"] = np.int32(1)
            return self.w.create_dataset(name, np.array(a.shape, dtype=np.uint64), attrs=attrs)
        data = np.ascontiguousarray(a.T)  # dimensions reversed, as MATLAB stores them
        if compress and data.nbytes >= 4096:
            chunks = list(data.shape)
            while int(np.prod(chunks)) * data.dtype.itemsize > (1 << 20):  # ~1 MiB chunks, as MATLAB
                ax = int(np.argmax(chunks))
                chunks[ax] = (chunks[ax] + 1) // 2
            return self.w.create_dataset(name, data, attrs=attrs, chunks=chunks, compression="gzip")
        return self.w.create_dataset(name, data, attrs=attrs)

    def _string(self, name: str, value: str):
        codes = np.array([ord(c) for c in value], dtype=np.uint16).reshape(-1, 1)  # 1 x len, reversed
        attrs = {"MATLAB_class": "char", "MATLAB_int_decode": np.int32(2)}
        if codes.size == 0:
            attrs["MATLAB_empty"] = np.int32(1)
            return self.w.create_dataset(name, np.array([0, 0], dtype=np.uint64), attrs=attrs)
        return self.w.create_dataset(name, codes, attrs=attrs)

    def _cell(self, name: str, items, compress: bool):
        if "/#refs#" not in self.w._groups:
            self.w.create_group("#refs#")
        refs = []
        for it in items:
            self._refs += 1
            refs.append(self.put(f"#refs#/{self._refs:08d}", it, compress))
        if not refs:
            return self.w.create_dataset(name, np.array([0, 0], dtype=np.uint64),
                                         attrs={"MATLAB_class": "cell", "MATLAB_empty": np.int32(1)})
        arr = np.empty((1, len(refs)), dtype=object)  # an N x 1 cell, dimensions reversed
        arr[0, :] = refs
        return self.w.create_dataset(name, arr, attrs={"MATLAB_class": "cell"})

    def put(self, name: str, value, compress: bool = False):
        if isinstance(value, str):
            return self._string(name, value)
        if isinstance(value, (list, tuple)):
            return self._cell(name, value, compress)
        return self._array(name, value, compress)

    def put_streamed(self, name: str, matlab_shape, dtype, column_blocks):
        """A MATLAB array too large to transpose in memory.  ``column_blocks`` yields consecutive
        slabs of the STORED (dimension-reversed) array along its first axis -- i.e. slabs along the
        LAST MATLAB dimension, each already reversed."""
        a = np.dtype(dtype)
        cls = "double" if a.kind == "f" and a.itemsize == 8 else self._class_of(np.zeros(0, a))
        return self.w.create_dataset_streamed(name, tuple(matlab_shape)[::-1], a, column_blocks,
                                              attrs={"MATLAB_class": cls})

    def open_chunked(self, name: str, matlab_shape, dtype, stored_chunks):
        """A MATLAB array written chunk by chunk while it is being produced
        (:meth:`hdf5.FileWriter.open_chunked_dataset`); ``stored_chunks``: chunk extents of the
        STORED (dimension-reversed) array."""
        a = np.dtype(dtype)
        cls = "double" if a.kind == "f" and a.itemsize == 8 else self._class_of(np.zeros(0, a))
        return self.w.open_chunked_dataset(name, tuple(matlab_shape)[::-1], a, stored_chunks,
                                           attrs={"MATLAB_class": cls})

    def close(self):
        self.w.close()


def savemat73(path: str, variables: dict, compress: bool = False, created: str | None = None) -> None:
    """``save(path, ..., '-v7.3')`` for a dict of variables in MATLAB orientation: arrays, scalars,
    strings, and lists (written as N x 1 cell arrays).  ``compress``: chunked + deflate datasets
    like MATLAB's own files (default: contiguous, which every HDF5 reader takes and which can be
    memory-mapped)."""
    w = _MatWriter(path, created)
    try:
        for k, v in variables.items():
            w.put(k, v, compress)
    finally:
        w.close()


def _load_mat(path: str, names):
    """{name: array} in MATLAB orientation for the requested variables (cells -> lists)."""
    if _is_hdf5(path):
        return loadmat73(path, names)
    from scipy.io import loadmat
    raw = loadmat(path, variable_names=list(names), squeeze_me=False)
    out = {}
    for n in names:
        if n not in raw:
            continue
        v = raw[n]
        out[n] = [np.asarray(c).squeeze() for c in v.ravel()] if v.dtype == object else v
    return out


def _vec(a):
    return np.asarray(a, dtype=np.float64).reshape(-1)


# ---------------------------------------------------------------------------------------------
# inputs of the path
# ---------------------------------------------------------------------------------------------

def load_learned_model(path: str) -> dict:
    """Variables of process_qsos.m:30-35."""
    names = ("rest_wavelengths", "mu", "M", "log_omega", "log_c_0", "log_tau_0", "log_beta")
    m = _load_mat(path, names)
    missing = [n for n in names if n not in m]
    if missing:
        raise KeyError(f"{path} lacks {missing}")
    G = _vec(m["rest_wavelengths"]).size
    M = np.asarray(m["M"], dtype=np.float64)
    if M.shape[0] != G:
        M = M.T
    return dict(rest_wavelengths=_vec(m["rest_wavelengths"]), mu=_vec(m["mu"]), M=np.asfortranarray(M),
                log_omega=_vec(m["log_omega"]), log_c_0=float(_vec(m["log_c_0"])[0]),
                log_tau_0=float(_vec(m["log_tau_0"])[0]), log_beta=float(_vec(m["log_beta"])[0]))


def load_dla_samples(path: str) -> dict:
    """Variables of process_qsos.m:38-40 (+ lls_nhi_samples when present, set_lls_parameters.m:63)."""
    m = _load_mat(path, ("offset_samples", "log_nhi_samples", "nhi_samples", "lls_nhi_samples",
                         "lls_log_nhi_samples"))
    out = {k: _vec(v) for k, v in m.items()}
    for need in ("offset_samples", "log_nhi_samples", "nhi_samples"):
        if need not in out:
            raise KeyError(f"{path} lacks {need}")
    return out


def load_catalog(path: str, names=("z_qsos", "thing_ids", "plates", "mjds", "fiber_ids", "snrs",
                                   "filter_flags", "los_inds", "dla_inds")) -> dict:
    """catalog.mat (build_catalogs.m:86-91): the per-quasar columns as flat vectors.  ``los_inds`` /
    ``dla_inds`` are containers.Map objects in the reference's file, which HDF5 stores opaquely; they
    are returned only when they are plain arrays."""
    m = _load_mat(path, names)
    return {k: (np.asarray(v).reshape(-1) if isinstance(v, np.ndarray) else v) for k, v in m.items()}


def load_preloaded_qsos(path: str, z_qsos, test_ind=None) -> list:
    """The ragged cell arrays of preload_qsos.m:64-79 as a list of per-quasar dicts, after the
    ``test_ind`` subset of process_qsos.m:56-61 (boolean mask or index array, 0-based).  In a
    -v7.3 file only the selected cells are dereferenced."""
    keys = ("all_wavelengths", "all_flux", "all_noise_variance", "all_pixel_mask")
    z = _vec(z_qsos)

    def pick(n):
        if n != z.size:
            raise ValueError(f"{n} spectra but {z.size} redshifts")
        if test_ind is None:
            return np.arange(n)
        t = np.asarray(test_ind)
        return np.flatnonzero(t) if t.dtype == bool else t

    if _is_hdf5(path):
        # the native cell reader where it applies (PreloadedReader.read_csr), split back into per-quasar views
        with PreloadedReader(path) as r:
            idx = pick(r.num_quasars)
            csr = r.read_csr(idx, z)
        o = csr["offsets"]
        return [dict(wavelengths=csr["wavelengths"][o[j]:o[j + 1]], flux=csr["flux"][o[j]:o[j + 1]],
                     noise_variance=csr["noise_variance"][o[j]:o[j + 1]], pixel_mask=csr["pixel_mask"][o[j]:o[j + 1]],
                     z_qso=float(z[i])) for j, i in enumerate(idx)]
    else:
        m = _load_mat(path, keys)
        idx = pick(len(m["all_wavelengths"]))
        cells = {k: [m[k][i] for i in idx] for k in keys}
    return [dict(wavelengths=_vec(cells["all_wavelengths"][j]), flux=_vec(cells["all_flux"][j]),
                 noise_variance=_vec(cells["all_noise_variance"][j]),
                 pixel_mask=np.asarray(cells["all_pixel_mask"][j]).reshape(-1).astype(np.uint8),
                 z_qso=float(z[i])) for j, i in enumerate(idx)]


# ---------------------------------------------------------------------------------------------
# native cell reader (csrc/h5cells.c): the cells of a -v7.3 cell array, many at a time, in threads
# ---------------------------------------------------------------------------------------------

_H5CELLS_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "h5cells.c")
_H5CELLS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libgpdla_h5cells.so")
_h5cells = None


def build_h5cells(force: bool = False) -> str:
    """gcc -O2 -fopenmp csrc/h5cells.c -lz -> csrc/libgpdla_h5cells.so (host code: no GPU involved)."""
    import subprocess
    if force or not os.path.exists(_H5CELLS_LIB) or os.path.getmtime(_H5CELLS_LIB) < os.path.getmtime(_H5CELLS_SRC):
        tmp = f"{_H5CELLS_LIB}.{os.getpid()}.tmp"  # the ranks of a run may all get here at once: publish atomically
        try:
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-Wall", _H5CELLS_SRC, "-lz", "-o", tmp],
                                  stderr=subprocess.DEVNULL if not force else None)
            os.replace(tmp, _H5CELLS_LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return _H5CELLS_LIB


def _load_h5cells():
    """The native cell reader, or None when it cannot be built / loaded (no gcc or zlib on the box):
    :class:`PreloadedReader` then reads every cell with the Python reader -- same arrays, slower."""
    global _h5cells
    if _h5cells is None:
        import ctypes as C
        try:
            # an existing library is used as it is (__graft_entry__.build() refreshes it); a missing one is built
            lib = C.CDLL(_H5CELLS_LIB if os.path.exists(_H5CELLS_LIB) else build_h5cells())
            lib.gpdla_h5cells_sizes.restype = C.c_int
            lib.gpdla_h5cells_sizes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p,
                                                C.c_void_p, C.c_int]
            lib.gpdla_h5cells_read.restype = C.c_int64
            lib.gpdla_h5cells_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_int32,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            _h5cells = lib
        except Exception:  # (a missing compiler must not stop a run: the Python reader is complete)
            _h5cells = False
    return _h5cells or None


class PreloadedReader:
    """Random access to the ragged cell arrays of a ``-v7.3`` preloaded_qsos.mat (preload_qsos.m:64-79):
    the file is opened once, the four reference tables are read, and a rank then dereferences only
    the quasars of its own block -- pixel counts come from the dataset headers alone, so sharding a
    run by pixel count (distributed.shard_bounds) reads no spectrum.  :meth:`read` is the pure-Python
    reader (one thread: ~4000-11000 quasars/s depending on the host core and on whether the cells
    are compressed; header parsing under the GIL, so threads do not help).
    :meth:`read_csr` is the fast path: the same cells through csrc/h5cells.c (object headers, chunk
    B-trees and zlib in C, cells spread over threads), straight into the flat arrays a batch
    upload takes; cells outside that reader's subset, or a box without gcc, fall back to the
    Python reader cell by cell."""

    KEYS = ("all_wavelengths", "all_flux", "all_noise_variance", "all_pixel_mask")
    _DTYPES = (np.float64, np.float64, np.float64, np.uint8)

    def __init__(self, path: str):
        if not _is_hdf5(path):
            raise ValueError(f"{path}: not a -v7.3 file (use load_preloaded_qsos for -v7 files)")
        self.path = path
        self._f = hdf5.File(path)
        self._refs = {}
        for k in self.KEYS:
            if k not in self._f:
                self._f.close()
                raise KeyError(f"{path} lacks {k}")
            self._refs[k] = self._f[k].read().T.ravel(order="F")
        self.num_quasars = int(self._refs[self.KEYS[0]].size)
        self._map = None  # (address, length) of the file's memory map, for the native reader
        self.threads = max(1, min(16, len(os.sched_getaffinity(0))))

    def _native(self):
        lib = _load_h5cells()
        if lib is not None and self._map is None:
            view = np.frombuffer(self._f._mm, dtype=np.uint8)  # the map hdf5.File already holds
            self._map = (view.ctypes.data, view.size, view)
        return lib

    def _python_count(self, i) -> int:
        ds = self._f.dereference(self._refs["all_wavelengths"][i])
        if len(ds.shape) == 1 and "MATLAB_empty" in ds.attrs:
            return 0
        return int(np.prod(ds.shape))

    def pixel_counts(self, indices=None) -> np.ndarray:
        """Stored pixels per quasar (``numel(all_wavelengths{i})``) for the 0-based ``indices``."""
        idx = np.arange(self.num_quasars) if indices is None else np.asarray(indices)
        lib = self._native()
        if lib is None or idx.size == 0:
            return np.array([self._python_count(i) for i in idx], dtype=np.int64)
        addrs = np.ascontiguousarray(self._refs["all_wavelengths"][idx], dtype=np.uint64)
        counts = np.empty(idx.size, dtype=np.int64)
        sizes = np.empty(idx.size, dtype=np.int32)
        lib.gpdla_h5cells_sizes(self._map[0], self._map[1], self._f.userblock_size, addrs.ctypes.data, idx.size, counts.ctypes.data,
                                sizes.ctypes.data, self.threads)
        for j in np.flatnonzero(counts < 0):  # outside the native reader's subset (e.g. an empty cell)
            counts[j] = self._python_count(idx[j])
        return counts

    def read_csr(self, indices, z_qsos) -> dict:
        """The spectra of the 0-based ``indices`` as the flat arrays a batch upload takes
        (``api.spectra_to_csr``'s dict: offsets, wavelengths, flux, noise_variance, pixel_mask,
        z_qsos), read by the native reader where it applies."""
        z = _vec(z_qsos)
        if z.size != self.num_quasars:
            raise ValueError(f"{self.num_quasars} spectra but {z.size} redshifts")
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        counts = self.pixel_counts(idx)
        offsets = np.zeros(idx.size + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        total = int(offsets[-1])
        out = dict(offsets=offsets, z_qsos=np.ascontiguousarray(z[idx], dtype=np.float64))
        lib = self._native()
        for key, dt, name in zip(self.KEYS, self._DTYPES, ("wavelengths", "flux", "noise_variance", "pixel_mask")):
            flat = np.empty(total, dtype=dt)
            status = np.full(idx.size, -1, dtype=np.int8)
            if lib is not None and idx.size:
                addrs = np.ascontiguousarray(self._refs[key][idx], dtype=np.uint64)
                byte_off = np.ascontiguousarray(offsets[:-1] * flat.itemsize)
                lib.gpdla_h5cells_read(self._map[0], self._map[1], self._f.userblock_size, addrs.ctypes.data, idx.size, flat.itemsize,
                                       int(flat.dtype.kind == "f"), flat.ctypes.data, byte_off.ctypes.data,
                                       counts.ctypes.data, status.ctypes.data, self.threads)
            for j in np.flatnonzero(status):  # the Python reader, for whatever the native one left
                v = self._vec(key, idx[j], dt)
                if v.size != counts[j]:
                    raise ValueError(f"{self.path}: {key}{{{idx[j] + 1}}} has {v.size} elements, "
                                     f"all_wavelengths{{{idx[j] + 1}}} {counts[j]}")
                flat[offsets[j]:offsets[j + 1]] = v
            out[name] = flat
        return out

    def _vec(self, key, i, dt):
        # the cells are plain numeric / logical vectors: their class attributes are not needed,
        # only an empty cell (stored as its dimensions, MATLAB_empty) must be told apart
        ds = self._f.dereference(self._refs[key][i])
        if len(ds.shape) == 1 and "MATLAB_empty" in ds.attrs:
            return np.zeros(0, dtype=dt)
        return np.asarray(ds.read(), dtype=dt).reshape(-1)

    def read(self, indices, z_qsos) -> list:
        """Per-quasar dicts (as :func:`load_preloaded_qsos` returns) for the 0-based ``indices``;
        ``z_qsos``: the catalogue's redshift column for ALL quasars of the file."""
        z = _vec(z_qsos)
        if z.size != self.num_quasars:
            raise ValueError(f"{self.num_quasars} spectra but {z.size} redshifts")
        vec = self._vec

        return [dict(wavelengths=vec("all_wavelengths", i, np.float64), flux=vec("all_flux", i, np.float64),
                     noise_variance=vec("all_noise_variance", i, np.float64),
                     pixel_mask=vec("all_pixel_mask", i, np.uint8), z_qso=float(z[i]))
                for i in np.asarray(indices)]

    def close(self):
        self._map = None  # (drops the view of the map before hdf5.File closes it)
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---------------------------------------------------------------------------------------------
# outputs of the path
# ---------------------------------------------------------------------------------------------

#: process_qsos.m:236-244 (result variables; the run metadata of :236-238 are keyword arguments)
SAVED_VARIABLES = ("min_z_dlas", "max_z_dlas", "log_priors_no_dla", "log_priors_dla",
                   "log_likelihoods_no_dla", "sample_log_likelihoods_dla", "log_likelihoods_dla",
                   "log_posteriors_no_dla", "log_posteriors_dla", "model_posteriors", "p_no_dlas",
                   "p_dlas")

#: multi_dlas/process_qsos_multiple_dlas_meanflux.m:498-510
SAVED_VARIABLES_MULTI = ("min_z_dlas", "max_z_dlas", "sample_log_likelihoods_dla", "base_sample_inds",
                         "log_priors_no_dla", "log_priors_dla", "log_priors_lls",
                         "log_likelihoods_no_dla", "MAP_z_dlas", "MAP_log_nhis", "log_likelihoods_dla",
                         "log_likelihoods_lls", "log_posteriors_no_dla", "log_posteriors_dla",
                         "log_posteriors_lls", "model_posteriors", "p_no_dlas", "p_dlas", "p_lls",
                         "all_exceptions", "sample_log_likelihoods_lls")

#: how this package's multi-DLA arrays ([nq, model, S] etc.) map onto MATLAB's axes (multi :110-131)
_MULTI_TO_MATLAB = {"sample_log_likelihoods_dla": (0, 2, 1),   # [nq, md, S] -> [nq, S, md]
                    "base_sample_inds": (0, 2, 1)}              # [nq, md-1, S] -> [nq, S, md-1]


def _metadata(w: _MatWriter, results: dict, run_metadata: dict, extra=()):
    for k, v in run_metadata.items():
        w.put(k, v)
    for k in ("prior_z_qso_increase", "max_z_cut", "num_lines") + tuple(extra):
        if k in results and k not in run_metadata:
            w.put(k, np.float64(results[k]))


def _streamed_table(w: _MatWriter, name: str, table: np.ndarray, block_rows: int = 64):
    """[nq, S] -> the MATLAB variable [nq x S], i.e. a stored (S, nq) array, written in slabs of
    ``block_rows`` samples so that the 13 GB table of a DR12Q run is never duplicated."""
    nq, S = table.shape
    w.put_streamed(name, (nq, S), np.float64,
                   (np.ascontiguousarray(table[:, i:i + block_rows].T) for i in range(0, S, block_rows)))


def save_processed_qsos(path: str, results: dict, test_ind=None, **run_metadata) -> None:
    """process_qsos.m:236-250 as a ``-v7.3`` file ``CDDF_analysis`` reads unchanged: the result
    variables (column vectors, ``sample_log_likelihoods_dla`` [nq x S], ``model_posteriors``
    [nq x 2]), ``test_ind`` (logical column, read back as ``f['test_ind'][0, :]``,
    qso_loader.py:95) and the run metadata the script echoes (training_release, training_set_name,
    dla_catalog_name, prior_ind, release, test_set_name, ...)."""
    w = _MatWriter(path)
    try:
        _metadata(w, results, run_metadata)
        if test_ind is not None:
            w.put("test_ind", np.asarray(test_ind, dtype=bool).reshape(-1, 1))
        for k in SAVED_VARIABLES:
            v = np.asarray(results[k], dtype=np.float64)
            if k == "sample_log_likelihoods_dla" and v.nbytes > (256 << 20):
                _streamed_table(w, k, v)
            else:
                w.put(k, v)
        # The MAP sample of generate_ascii_catalog.m:73-80, which the evidence kernel finds anyway: an
        # extra, stored as single_MAP_*.  NOT under the multi-DLA names: QSOLoader takes a file with a
        # 'MAP_log_nhis' key for a multi-DLA file and indexes it [nq, model, slot] (qso_loader.py:106-109,
        # 160-162) -- found by running the reference's loader on these files.
        for k in ("MAP_inds", "MAP_z_dlas", "MAP_log_nhis"):
            if k in results and np.ndim(results[k]) == 1:
                w.put("single_" + k, np.asarray(results[k], dtype=np.float64))
    finally:
        w.close()


def save_processed_qsos_multi(path: str, results: dict, test_ind=None, **run_metadata) -> None:
    """multi_dlas/process_qsos_multiple_dlas_meanflux.m:498-523 as a ``-v7.3`` file:
    ``sample_log_likelihoods_dla`` [nq x S x max_dlas] (read back as ``[max_dlas, S, nq]``,
    calc_cddf.py:266), ``base_sample_inds`` uint32 [nq x S x max_dlas-1] (:116), ``MAP_*``
    [nq x model x slot] (:129-131; ``f['MAP_log_nhis'][()].T``, qso_loader.py:107-109),
    ``model_posteriors`` [nq x 2+max_dlas], ``all_exceptions`` (:139)."""
    w = _MatWriter(path)
    try:
        _metadata(w, results, run_metadata, extra=("k", "min_z_cut", "num_dla_samples",
                                                   "normalization_min_lambda", "normalization_max_lambda"))
        if test_ind is not None:
            w.put("test_ind", np.asarray(test_ind, dtype=bool).reshape(-1, 1))
        for k in SAVED_VARIABLES_MULTI + ("MAP_inds",):
            if k not in results:
                if k == "MAP_inds":
                    continue
                raise KeyError(f"results lack {k}")
            v = np.asarray(results[k])
            if k in _MULTI_TO_MATLAB:
                v = np.transpose(v, _MULTI_TO_MATLAB[k])
            w.put(k, v.astype(np.uint32) if k == "base_sample_inds" else v.astype(np.float64))
    finally:
        w.close()


class ProcessedStreamWriter:
    """``save_processed_qsos`` / ``save_processed_qsos_multi`` for a run whose results arrive batch
    by batch: the per-sample tables -- 1.6 GB for a DR12Q shard, the only large variables -- are
    transposed into MATLAB's order and written as one column of HDF5 chunks per batch WHILE the next
    batch is swept (:meth:`append`, called by the pipeline's download thread); :meth:`finish`
    writes the per-quasar variables and the metadata.  The file holds the same variables with the
    same shapes and classes as the one-shot writers'; the large ones are chunked (as MATLAB's own
    ``-v7.3`` files are) instead of contiguous.  ``batch``: the chunks' extent along the quasar axis --
    every batch but the last must hold a whole multiple of it."""

    _TARGET = 8 << 20  # bytes per chunk, about

    def __init__(self, path: str, num_quasars: int, num_samples: int, batch: int, max_dlas: int = 0):
        self.nq, self.S, self.B, self.md = int(num_quasars), int(num_samples), max(1, int(batch)), int(max_dlas)
        self.w = _MatWriter(path)
        self.done = 0
        S, nq, B = self.S, self.nq, min(self.B, max(1, self.nq))
        self.B = B

        def rows(itemsize):  # chunk extent along the sample axis: about _TARGET bytes per chunk, and an even
            want = max(1, min(S, self._TARGET // (itemsize * B)))  # split of S (every chunk is stored whole:
            return -(-S // -(-S // want))                          # a ragged last chunk row would be padding)
        self.streams = {}
        try:
            if self.md:  # stored (dimension-reversed): [md, S, nq], [md - 1, S, nq], [S, nq]
                self.streams["sample_log_likelihoods_dla"] = (
                    self.w.open_chunked("sample_log_likelihoods_dla", (nq, S, self.md), np.float64, (1, rows(8), B)), 8)
                if self.md > 1:
                    self.streams["base_sample_inds"] = (
                        self.w.open_chunked("base_sample_inds", (nq, S, self.md - 1), np.uint32, (1, rows(4), B)), 4)
                self.streams["sample_log_likelihoods_lls"] = (
                    self.w.open_chunked("sample_log_likelihoods_lls", (nq, S), np.float64, (rows(8), B)), 8)
            else:
                self.streams["sample_log_likelihoods_dla"] = (
                    self.w.open_chunked("sample_log_likelihoods_dla", (nq, S), np.float64, (rows(8), B)), 8)
        except Exception:
            self.w.close()
            raise

    @property
    def streamed(self):
        return tuple(self.streams)

    def append(self, at: int, tables: dict) -> None:
        """The rows ``at .. at + n`` of the run: ``tables[name]`` = this package's orientation,
        [n, S] or [n, model, S]."""
        n, B = None, self.B
        for name, (st, _) in self.streams.items():
            t = np.asarray(tables[name])
            n = t.shape[0] if n is None else n
            if at % B or t.shape[0] != n or (n % B and at + n != self.nq):
                raise ValueError(f"batch [{at}, {at + n}) does not sit on the {B}-quasar chunk grid")
            cs = st.chunks[-2]
            for c0 in range(0, n, B):  # a batch may span several chunk columns
                if t.ndim == 2:
                    tt = np.ascontiguousarray(t[c0:c0 + B].T)  # [S, <= B]
                    for r0 in range(0, self.S, cs):
                        st.write_chunk((r0, at + c0), tt[r0:r0 + cs])
                else:
                    for mdl in range(t.shape[1]):
                        tt = np.ascontiguousarray(t[c0:c0 + B, mdl, :].T)
                        for r0 in range(0, self.S, cs):
                            st.write_chunk((mdl, r0, at + c0), tt[None, r0:r0 + cs])
        self.done += n or 0

    def finish(self, results: dict, test_ind=None, **run_metadata) -> None:
        """The remaining variables of ``results`` (all rows) and the metadata; closes the file."""
        w = self.w
        try:
            if self.done != self.nq:
                raise ValueError(f"{self.done} of {self.nq} quasars were appended")
            for st, _ in self.streams.values():
                st.close()
            extra = ("k", "min_z_cut", "num_dla_samples", "normalization_min_lambda",
                     "normalization_max_lambda") if self.md else ()
            _metadata(w, results, run_metadata, extra=extra)
            if test_ind is not None:
                w.put("test_ind", np.asarray(test_ind, dtype=bool).reshape(-1, 1))
            if self.md:
                for k in SAVED_VARIABLES_MULTI + ("MAP_inds",):
                    if k in self.streams:
                        continue
                    if k not in results:
                        if k in ("MAP_inds", "base_sample_inds"):
                            continue
                        raise KeyError(f"results lack {k}")
                    v = np.asarray(results[k])
                    if k in _MULTI_TO_MATLAB:
                        v = np.transpose(v, _MULTI_TO_MATLAB[k])
                    w.put(k, v.astype(np.uint32) if k == "base_sample_inds" else v.astype(np.float64))
            else:
                for k in SAVED_VARIABLES:
                    if k not in self.streams:
                        w.put(k, np.asarray(results[k], dtype=np.float64))
                for k in ("MAP_inds", "MAP_z_dlas", "MAP_log_nhis"):  # (see save_processed_qsos)
                    if k in results and np.ndim(results[k]) == 1:
                        w.put("single_" + k, np.asarray(results[k], dtype=np.float64))
        finally:
            w.close()

    def abort(self) -> None:
        try:
            self.w.close()
        except Exception:
            pass


def load_processed_qsos(path: str) -> dict:
    """A processed_qsos_*.mat written by the reference or by this package, in this package's
    orientation (vectors flat; multi-DLA 3-D arrays as [nq, model, S] / [nq, model, slot])."""
    m = loadmat73(path) if _is_hdf5(path) else _load_mat(path, SAVED_VARIABLES_MULTI + SAVED_VARIABLES
                                                        + ("test_ind", "MAP_inds"))
    out = {}
    for k, v in m.items():
        if k.startswith("single_MAP_"):  # save_processed_qsos: the single-DLA MAP extras
            k = k[len("single_"):]
        if isinstance(v, np.ndarray):
            if v.ndim == 2 and 1 in v.shape and k not in ("model_posteriors",):
                v = v.reshape(-1)
            elif v.ndim == 3 and k in _MULTI_TO_MATLAB:
                v = np.transpose(v, (0, 2, 1))
        out[k] = v
    return out


# ---------------------------------------------------------------------------------------------
# chunk files of a sharded run (CDDF_analysis/sbatch_reunion.py:13-63)
# ---------------------------------------------------------------------------------------------

def chunk_filename(directory: str, test_set_name: str, lo: int, hi: int, multi: bool = False,
                   width: int = 6) -> str:
    """Name of the chunk holding quasars [lo, hi) (0-based positions within the run's ``test_ind``
    selection) of a sharded run: the reference's output name (process_qsos.m:246-248; multi
    :512-521) followed by the zero-padded range, so that sorting the names orders the chunks the
    way ``mat_combine`` must be handed them."""
    stem = f"processed_qsos_multi_meanflux{test_set_name}" if multi else f"processed_qsos_{test_set_name}"
    return f"{directory}/{stem}_{lo:0{width}d}-{hi:0{width}d}.mat"


#: largest piece combine_processed_chunks reads / writes at a time
COMBINE_SLAB_BYTES = 256 << 20


def combine_processed_chunks(paths, out_path: str, test_ind=None) -> None:
    """What ``mat_combine`` (CDDF_analysis/sbatch_reunion.py:13-63) does -- every variable whose
    quasar axis has the first chunk's length is concatenated along that axis, everything else is
    taken from the first chunk -- written as a MATLAB ``-v7.3`` file and streamed in slabs of at
    most 256 MB (``Dataset.read_slab``: memory-mapped for contiguous sources, chunk by chunk for the
    chunked tables the streamed writer produces; the 3-D tables of the multi-DLA driver,
    ``[max_dlas, S, nq]``, by (model, range of samples)), so that the 13 GB sample table of a DR12Q
    run -- 52 GB for four DLA models -- is never held in memory (the reference's script peaks at ~150 GB, sbatch_reunion.py:6-7).  ``test_ind``: the combined run's selection; by default the
    OR of the chunks' own masks (``mat_combine`` keeps the first chunk's, which then selects only
    that chunk's quasars)."""
    paths = list(paths)
    files = [hdf5.File(p) for p in paths]
    try:
        first = files[0]
        names = [k for k in first.keys() if not k.startswith("#")]
        sizes = []
        for f in files:
            sizes.append(int(f["p_dlas"].shape[-1]))
        w = hdf5.FileWriter(out_path, userblock=matlab_userblock())
        try:
            for name in names:
                ds = first[name]
                attrs = {k: v for k, v in ds.attrs.items()}
                if name == "test_ind":
                    mask = test_ind
                    if mask is None:
                        mask = np.zeros(ds.shape, dtype=bool)
                        for f in files:
                            mask |= f["test_ind"].read().astype(bool)
                    w.create_dataset(name, np.asarray(mask, dtype=np.uint8).reshape(ds.shape), attrs=attrs)
                    continue
                per_quasar = len(ds.shape) >= 1 and ds.shape[-1] == sizes[0] and all(
                    name in f and f[name].shape[:-1] == ds.shape[:-1] and f[name].shape[-1] == n
                    for f, n in zip(files, sizes))
                if not per_quasar or "MATLAB_empty" in attrs:
                    w.create_dataset(name, ds.read(), attrs=attrs)
                    continue
                total = sum(sizes)
                shape = ds.shape[:-1] + (total,)
                if len(shape) == 1:
                    w.create_dataset(name, np.concatenate([f[name].read() for f in files]), attrs=attrs)
                    continue
                # slab by slab: contiguous sources are memory-mapped, chunked ones (what
                # ProcessedStreamWriter writes) are read chunk row by chunk row.  2-D tables [S, nq]
                # go by ranges of the slowest dimension; 3-D tables [max_dlas, S, nq] -- whose slowest
                # dimension is the model, one "row" of it 13 GB for a DR12Q run -- by (model, range
                # of samples): consecutive pieces of the row-major bytes either way, <= 256 MB each
                sources = [f[name] for f in files]
                budget = COMBINE_SLAB_BYTES
                if len(shape) == 2:
                    row_bytes = shape[1] * ds.dtype.itemsize
                    step = max(1, budget // max(row_bytes, 1))
                    blocks = (np.concatenate([d.read_slab(i, i + step) for d in sources], axis=-1)
                              for i in range(0, shape[0], step))
                else:
                    row_bytes = int(np.prod(shape[2:])) * ds.dtype.itemsize
                    step = max(1, budget // max(row_bytes, 1))
                    blocks = (np.concatenate([d.read_slab(m_, m_ + 1, axis1=(i, i + step)) for d in sources], axis=-1)
                              for m_ in range(shape[0]) for i in range(0, shape[1], step))
                w.create_dataset_streamed(name, shape, ds.dtype, blocks, attrs=attrs, in_order=True)
        finally:
            w.close()
    finally:
        for f in files:
            f.close()
