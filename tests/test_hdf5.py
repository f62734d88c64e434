"""The package's own HDF5 (-v7.3) reader/writer: a file written by MATLAB itself, round trips, and
byte-level checks of the structures the writer emits against the HDF5 File Format Specification.
No libhdf5 exists in this image to cross-read with: where h5py is importable the cross-read runs,
otherwise the writer is self-validated (stated in DESIGN.md)."""
import os
import struct

import numpy as np
import pytest

from gp_dla_detection_amd import hdf5, io

MATLAB_FILE = None
try:  # a genuine MATLAB 7.4 -v7.3 file that SciPy ships with its test data
    import scipy.io.matlab
    MATLAB_FILE = os.path.join(os.path.dirname(scipy.io.matlab.__file__), "tests", "data",
                               "testhdf5_7.4_GLNX86.mat")
except ImportError:  # pragma: no cover
    pass


@pytest.mark.skipif(not (MATLAB_FILE and os.path.exists(MATLAB_FILE)), reason="SciPy's MATLAB -v7.3 sample absent")
def test_reads_a_file_written_by_matlab():
    """Bytes produced by MATLAB + libhdf5: user block 512, superblock 0, symbol-table root group,
    version-1 object header, version-2 contiguous layout, MATLAB_class attribute."""
    with hdf5.File(MATLAB_FILE) as f:
        assert f.userblock_size == 512 and f.userblock().startswith(b"MATLAB 7.0 MAT-file")
        assert f.superblock_version == 0 and f.stored_base == 512 and f.eof == os.path.getsize(MATLAB_FILE)
        assert f.keys() == ["testdouble"]
        d = f["testdouble"]
        assert d.shape == (9, 1) and d.dtype == np.dtype("<f8") and d.attrs == {"MATLAB_class": "double"}
        np.testing.assert_allclose(d.read().ravel(), np.arange(9) * np.pi / 4, rtol=0, atol=1e-15)
        assert d[0, 0] == 0.0 and d[()].shape == (9, 1)
    m = io.loadmat73(MATLAB_FILE)
    assert m["testdouble"].shape == (1, 9)  # MATLAB orientation: a 1 x 9 row vector


def sample_variables():
    rng = np.random.default_rng(8)
    ragged = [rng.standard_normal(n) for n in (5, 1, 300, 17)]
    return {
        "scalar": np.float64(3.25),
        "row": rng.standard_normal((1, 7)),
        "col": rng.standard_normal(11),                       # 1-D: written as a column
        "matrix": rng.standard_normal((6, 4)),
        "cube": rng.standard_normal((3, 5, 2)),
        "counts": rng.integers(0, 2**32 - 1, size=(4, 3), dtype=np.uint32),
        "small_ints": np.arange(-5, 5, dtype=np.int16).reshape(2, 5),
        "flags": rng.uniform(size=(9, 1)) < 0.5,
        "name": "dr12q_minus_concordance",
        "empty": np.zeros((0, 3)),
        "ragged": ragged,
        "masks": [r > 0 for r in ragged],
        "big": rng.standard_normal((300, 41)),
    }


def check_round_trip(got, want):
    assert set(got) == set(want)
    assert got["scalar"].shape == (1, 1) and got["scalar"][0, 0] == 3.25
    for k in ("row", "matrix", "cube", "counts", "small_ints", "big"):
        assert got[k].dtype == np.asarray(want[k]).dtype
        np.testing.assert_array_equal(got[k], want[k])
    np.testing.assert_array_equal(got["col"], want["col"].reshape(-1, 1))
    assert got["flags"].dtype == bool and np.array_equal(got["flags"], want["flags"])
    assert got["name"] == want["name"]
    assert got["empty"].shape == (0, 3)
    assert len(got["ragged"]) == 4
    for a, b in zip(got["ragged"], want["ragged"]):
        np.testing.assert_array_equal(a.ravel(), b)
    for a, b in zip(got["masks"], want["masks"]):
        assert a.dtype == bool and np.array_equal(a.ravel(), b)


@pytest.mark.parametrize("compress", [False, True])
def test_mat73_round_trip(tmp_path, compress):
    """savemat73 -> loadmat73 for every value kind the path's files hold (double arrays of rank
    1-3, uint32, logical, char, empty, cell arrays of ragged double / logical vectors), contiguous
    and chunked + deflate (the layout MATLAB itself writes)."""
    want = sample_variables()
    p = str(tmp_path / "v.mat")
    io.savemat73(p, want, compress=compress)
    check_round_trip(io.loadmat73(p), want)
    with hdf5.File(p) as f:
        assert f["matrix"].shape == (4, 6)          # dimensions reversed on disk, as MATLAB / h5py show them
        assert f["cube"].shape == (2, 5, 3)
        assert f["matrix"].attrs["MATLAB_class"] == "double"
        assert f["flags"].attrs["MATLAB_class"] == "logical" and f["flags"].dtype == np.uint8
        assert f["name"].attrs["MATLAB_class"] == "char" and f["name"].dtype == np.dtype("<u2")
        assert f["ragged"].is_reference and f["ragged"].attrs["MATLAB_class"] == "cell"
        assert "#refs#" in f and len(f["#refs#"]) == 8
        kind = f["big"]._layout_info()[0]
        assert kind == ("chunked" if compress else "contiguous")
        if compress:
            assert f["big"]._filters[0][0] == 1     # deflate
            assert os.path.getsize(p) < 300 * 41 * 8 + 60000


def test_writer_structures_follow_the_spec(tmp_path):
    """Byte-level checks of a written file against the HDF5 File Format Specification (III.A
    superblock 0, III.B/C B-tree and symbol nodes, III.D local heap, IV.A version-1 object
    headers) -- the things libhdf5 would verify when opening it."""
    p = str(tmp_path / "s.mat")
    names = [f"var_{i:03d}" for i in range(45)]  # > 8 per SNOD, > 32 SNODs would need 2 levels: here 6 SNODs
    io.savemat73(p, {n: np.full((2, 3), float(i)) for i, n in enumerate(names)}, created="Thu Jan 01 00:00:00 2026")
    b = open(p, "rb").read()
    # MATLAB user block
    assert b[:19] == b"MATLAB 7.3 MAT-file" and b[116:124] == b"\x00" * 8 and b[124:128] == b"\x00\x02IM"
    assert b[128:512] == b"\x00" * 384
    # superblock version 0 at 512; every address below is relative to 512
    sb = 512
    assert b[sb:sb + 8] == hdf5.SIGNATURE
    assert b[sb + 8:sb + 16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])
    leaf_k, internal_k, flags = struct.unpack_from("<HHI", b, sb + 16)
    assert (leaf_k, internal_k, flags) == (4, 16, 0)
    base, freespace, eof, driver = struct.unpack_from("<4Q", b, sb + 24)
    assert base == 512 and freespace == hdf5.UNDEF and driver == hdf5.UNDEF
    assert eof == len(b)                                   # the end-of-file address is absolute
    name_off, root, cache, _, btree, heap = struct.unpack_from("<QQIIQQ", b, sb + 56)
    assert name_off == 0 and cache == 1

    def at(addr, n):
        return b[sb + addr: sb + addr + n]

    # root object header: version 1, one symbol-table message naming the same B-tree and heap
    ver, _, nmsg, refcount, hsize = struct.unpack_from("<BBHII", at(root, 12))
    assert (ver, nmsg, refcount) == (1, 1, 1) and root % 8 == 0
    mtype, msize, mflags = struct.unpack_from("<HHB", at(root + 16, 5))
    assert mtype == 0x11 and msize == 16 and hsize == 24
    assert struct.unpack_from("<QQ", at(root + 24, 16)) == (btree, heap)
    # local heap
    assert at(heap, 4) == b"HEAP" and at(heap, 8)[4] == 0
    hsz, free_off, hdata = struct.unpack_from("<QQQ", at(heap + 8, 24))
    seg = at(hdata, hsz)
    assert seg[:8] == b"\x00" * 8                            # the empty name at offset 0
    nxt, fsz = struct.unpack_from("<QQ", seg, free_off)
    assert nxt == 1 and free_off + fsz == hsz                # one free block, H5HL_FREE_NULL terminated
    # B-tree: one level-0 node over 6 symbol nodes, keys are heap offsets of names in strcmp order
    assert at(btree, 4) == b"TREE"
    ntype, level, used, left, right = struct.unpack_from("<BBHQQ", at(btree + 4, 20))
    assert (ntype, level, used, left, right) == (0, 0, 6, hdf5.UNDEF, hdf5.UNDEF)
    body = struct.unpack_from(f"<{2 * used + 1}Q", at(btree + 24, (2 * used + 1) * 8))
    keys, children = body[0::2], body[1::2]

    def heap_name(off):
        return seg[off: seg.index(b"\x00", off)].decode()

    assert keys[0] == 0 and [heap_name(k) for k in keys[1:]] == [names[min(8 * (i + 1), 45) - 1] for i in range(6)]
    assert len(at(btree, 24 + 65 * 8)) == 544                # a full node (2K = 32 children) is allocated
    seen = []
    for c in children:
        assert at(c, 4) == b"SNOD" and at(c, 6)[4] == 1
        count = struct.unpack_from("<H", at(c + 6, 2))[0]
        assert 1 <= count <= 8
        for e in range(count):
            noff, ohdr, ctype = struct.unpack_from("<QQI", at(c + 8 + 40 * e, 20))
            seen.append(heap_name(noff))
            assert ctype == 0 and ohdr % 8 == 0
            # the dataset's object header: messages are 8-byte multiples and fill the header exactly
            v, _, nm, rc, hs = struct.unpack_from("<BBHII", at(ohdr, 12))
            assert v == 1 and rc == 1
            pos, types = 0, []
            for _ in range(nm):
                t, sz = struct.unpack_from("<HH", at(ohdr + 16 + pos, 4))
                assert sz % 8 == 0
                types.append(t)
                pos += 8 + sz
            assert pos == hs and types[:4] == [1, 3, 5, 8] and types[4:] == [0x0C]
    assert seen == sorted(names)                             # entries sorted by name within and across nodes
    # layout message of one dataset: version 3, contiguous, address + size inside the file
    ds = hdf5.File(p)["var_007"]
    ver, cls, addr, size = struct.unpack_from("<BBQQ", ds._layout)
    assert (ver, cls, size) == (3, 1, 48) and sb + addr + size <= len(b) and addr % 8 == 0
    np.testing.assert_array_equal(ds.read(), np.full((3, 2), 7.0))


def test_large_groups_and_chunk_trees(tmp_path):
    """Multi-level B-trees on both sides: a cell array of 700 elements (#refs# holds 700 links:
    88 symbol nodes under 3 level-0 nodes under a level-1 root) and a dataset of 200 chunks
    (4 level-0 chunk nodes under a level-1 root)."""
    rng = np.random.default_rng(9)
    cells = [rng.standard_normal(int(n)) for n in rng.integers(1, 40, 700)]
    table = rng.standard_normal((200, 64))
    p = str(tmp_path / "big.mat")
    w = io._MatWriter(p)
    w.put("cells", cells)
    w.w.create_dataset("chunked", table, chunks=(1, 64), compression="gzip", shuffle=True)
    w.close()
    with hdf5.File(p) as f:
        refs = f["#refs#"]
        assert len(refs) == 700 and refs.keys() == sorted(refs.keys())
        bt = struct.unpack_from("<QQ", [m for m in refs._msgs if m.type == 0x11][0].data)[0]
        assert f._bytes(bt, 6)[5] == 1                       # the root node is a level-1 node
        got = io.loadmat73(p, ["cells"])["cells"]
        assert len(got) == 700
        for a, b in zip(got, cells):
            np.testing.assert_array_equal(a.ravel(), b)
        ds = f["chunked"]
        kind, addr, cdims = ds._layout_info()
        assert kind == "chunked" and cdims == (1, 64) and f._bytes(addr, 6)[5] == 1
        assert [x[0] for x in ds._filters] == [2, 1]         # shuffle, then deflate
        np.testing.assert_array_equal(ds.read(), table)
        np.testing.assert_array_equal(ds[17, :], table[17])


def test_streamed_table_equals_in_memory(tmp_path):
    rng = np.random.default_rng(10)
    t = rng.standard_normal((37, 500))
    a, b = str(tmp_path / "a.mat"), str(tmp_path / "b.mat")
    w = io._MatWriter(a)
    io._streamed_table(w, "sample_log_likelihoods_dla", t, block_rows=64)
    w.close()
    io.savemat73(b, {"sample_log_likelihoods_dla": t})
    for p in (a, b):
        with hdf5.File(p) as f:
            assert f["sample_log_likelihoods_dla"].shape == (500, 37)   # [S, nq], calc_cddf.py:217-220
            np.testing.assert_array_equal(f["sample_log_likelihoods_dla"].read().T, t)


def test_errors_are_reported(tmp_path):
    p = tmp_path / "x.mat"
    p.write_bytes(b"MATLAB 5.0 MAT-file" + b" " * 600)
    with pytest.raises(hdf5.HDF5Error, match="no HDF5 superblock"):
        hdf5.File(str(p))
    good = str(tmp_path / "g.mat")
    io.savemat73(good, {"a": np.arange(100.0)})
    cut = tmp_path / "cut.mat"
    cut.write_bytes(open(good, "rb").read()[:-200])
    with pytest.raises(hdf5.HDF5Error, match="truncated"):
        hdf5.File(str(cut))
    with hdf5.File(good) as f, pytest.raises(KeyError):
        f["missing"]
    with pytest.raises(hdf5.HDF5Error):
        w = hdf5.FileWriter(str(tmp_path / "d.h5"))
        w.create_dataset("a", np.zeros(3))
        w.create_dataset("a", np.zeros(3))


def test_cross_read_with_h5py_when_available(tmp_path):
    h5py = pytest.importorskip("h5py")
    want = sample_variables()
    p = str(tmp_path / "v.mat")
    io.savemat73(p, want, compress=True)
    with h5py.File(p, "r") as f:
        np.testing.assert_array_equal(f["matrix"][()].T, want["matrix"])
        np.testing.assert_array_equal(f["cube"][()].T, want["cube"])
        assert f["matrix"].attrs["MATLAB_class"] == b"double"
        first = f[f["ragged"][0, 0]]
        np.testing.assert_array_equal(first[()].ravel(), want["ragged"][0])


def test_random_round_trips_hypothesis(tmp_path_factory):
    """Property test of the writer against the reader: random dtypes, ranks 1-3, extents that do
    and do not divide the chunk extents, contiguous / chunked / chunked+deflate(+shuffle)."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp
    dtypes = st.sampled_from(["<f8", "<f4", "<i1", "<i2", "<i4", "<i8", "<u1", "<u2", "<u4", "<u8"])
    base = tmp_path_factory.mktemp("hyp")
    counter = [0]

    @settings(max_examples=40, deadline=None, derandomize=True)
    @given(data=st.data(), dtype=dtypes, shape=hnp.array_shapes(min_dims=1, max_dims=3, min_side=1, max_side=9),
           mode=st.sampled_from(["contiguous", "chunked", "gzip", "gzip+shuffle"]))
    def run(data, dtype, shape, mode):
        arr = data.draw(hnp.arrays(dtype, shape, elements=(st.floats(-1e6, 1e6, width=32) if dtype[1] == "f"
                                                            else st.integers(0, 100))))
        counter[0] += 1
        p = str(base / f"h{counter[0]}.h5")
        w = hdf5.FileWriter(p)
        if mode == "contiguous":
            w.create_dataset("a", arr, attrs={"note": "x", "n": np.int32(7)})
        else:
            chunks = tuple(data.draw(st.integers(1, s)) for s in shape)
            w.create_dataset("a", arr, chunks=chunks, compression="gzip" if mode.startswith("gzip") else None,
                             shuffle=mode.endswith("shuffle"), attrs={"note": "x", "n": np.int32(7)})
        w.close()
        with hdf5.File(p) as f:
            d = f["a"]
            assert d.shape == arr.shape and d.dtype == arr.dtype
            np.testing.assert_array_equal(d.read(), arr)
            assert d.read().flags.writeable  # every layout: a caller may modify what it loaded
            lo = data.draw(st.integers(0, shape[0]))
            hi = data.draw(st.integers(lo, shape[0] + 2))
            np.testing.assert_array_equal(d.read_slab(lo, hi), arr[lo:hi])  # streamed recombination reads these
            if len(shape) >= 2:  # ... and, for the 3-D tables of the multi-DLA driver, ranges of the second dimension
                lo1 = data.draw(st.integers(0, shape[1]))
                hi1 = data.draw(st.integers(lo1, shape[1] + 2))
                np.testing.assert_array_equal(d.read_slab(lo, hi, axis1=(lo1, hi1)), arr[lo:hi, lo1:hi1])
            assert d.attrs["note"] == "x" and d.attrs["n"] == 7
            assert f.eof == os.path.getsize(p)

    run()


@pytest.mark.skipif(not (MATLAB_FILE and os.path.exists(MATLAB_FILE)), reason="SciPy's MATLAB -v7.3 sample absent")
def test_native_cell_reader_on_a_file_written_by_matlab():
    """csrc/h5cells.c on bytes MATLAB wrote (version-1 object header, version-2 contiguous layout)."""
    lib = io._load_h5cells()
    if lib is None:
        pytest.skip("no gcc / zlib on this box: the native cell reader is not built")
    with hdf5.File(MATLAB_FILE) as f:
        view = np.frombuffer(f._mm, dtype=np.uint8)
        addrs = np.array([f["testdouble"].addr], dtype=np.uint64)
        counts, sizes = np.empty(1, np.int64), np.empty(1, np.int32)
        lib.gpdla_h5cells_sizes(view.ctypes.data, view.size, f.userblock_size, addrs.ctypes.data, 1,
                                counts.ctypes.data, sizes.ctypes.data, 1)
        assert counts[0] == 9 and sizes[0] == 8
        out, status = np.empty(9), np.full(1, -1, np.int8)
        off = np.zeros(1, np.int64)
        failed = lib.gpdla_h5cells_read(view.ctypes.data, view.size, f.userblock_size, addrs.ctypes.data, 1, 8, 1,
                                        out.ctypes.data, off.ctypes.data, counts.ctypes.data, status.ctypes.data, 1)
        assert failed == 0 and status[0] == 0
        np.testing.assert_array_equal(out, f["testdouble"].read().ravel())
        # asked for as 8-byte INTEGERS the same cell is refused (bytes are copied, never converted)
        failed = lib.gpdla_h5cells_read(view.ctypes.data, view.size, f.userblock_size, addrs.ctypes.data, 1, 8, 0,
                                        out.ctypes.data, off.ctypes.data, counts.ctypes.data, status.ctypes.data, 1)
        assert failed == 1 and status[0] == -1
        del view
