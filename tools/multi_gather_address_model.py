#!/usr/bin/env python3
"""Host-side address model of the profile gathers of the multi-DLA sweeps (no GPU, no library).

k_sweep_multi_slim<ND> / k_sweep_multi / k_sweep_multi_split read the per-quasar Voigt profile table
that k_profiles wrote:

    prof[((ql * 2 + kind) * S + i) * stride + p]          csrc/multi_kernels.hpp:37
    stride = ceil16(4 * ceil(max_pix / 4) + 4)             csrc/gpdla.hip, multi_alloc
    allocation = nq_sub * 2 * S * stride doubles           (same place)
    steps(q) = ceil(n_u(q) / 4),  n_u <= the quasar's pixel count      csrc/sweep_kernels.hpp:285
    shipped gather index:  p = min(4 * (rn + kAhead) + jj, 4 * steps + jj),  rn < steps, jj < 4
                                                            csrc/sweep_multi_slim_kernel.hpp:123, 152, 210

This enumerates the largest byte offset any lane can request -- over quasars of a sub-batch, both
kinds, every sample row, every K-step and jj -- for the SHIPPED (clamped) index and for the index of
the experiment abandoned in round 4 (the clamp removed: p = 4 * (rn + kAhead) + jj, optionally with
`pad` extra entries per row), and compares it with the allocation.  It also says whether the
allocation ends on a 4-KiB page boundary (then the first byte behind it is not mapped by this
allocation: hipMalloc hands out whole pages).

    python tools/multi_gather_address_model.py            # the shapes of tests/test_gpu_multi.py and tools/fuzz_multi.py
"""
import sys

import numpy as np

K_AHEAD = 4


def stride_of(max_pix: int, pad: int = 0) -> int:
    """multi_alloc (csrc/gpdla.hip): doubles per profile row; `pad`: extra entries (the abandoned build's idea)."""
    return ((4 * ((max_pix + 3) // 4) + 4 + pad + 15) // 16) * 16


def model(pixel_counts, S: int, clamped: bool, pad: int = 0, n_u=None):
    """Largest requested byte offset (exclusive end) against the allocation, for one sub-batch whose
    quasars have `pixel_counts` input pixels (n_u: stored pixels per quasar, default = all of them,
    the worst case).  Returns a dict."""
    pixel_counts = np.asarray(pixel_counts, dtype=np.int64)
    n_u = pixel_counts if n_u is None else np.asarray(n_u, dtype=np.int64)
    stride = stride_of(int(pixel_counts.max()), pad)
    alloc_bytes = len(pixel_counts) * 2 * S * stride * 8
    worst_end, worst = 0, None
    overrun_rows = 0
    for ql, nu in enumerate(n_u):
        steps = (int(nu) + 3) // 4
        p_last = 4 * steps + 3                                  # jj = 3
        rn = np.arange(max(steps, 1))
        p = 4 * (rn + K_AHEAD) + 3                              # the request of K-step rn, jj = 3
        prime = 4 * np.arange(K_AHEAD) + 3                      # the priming requests
        p_all = np.concatenate([prime, p])
        if clamped:
            p_all = np.minimum(p_all, p_last)
        p_max = int(p_all.max())
        if p_max >= stride:
            overrun_rows += 2 * S                               # every row of this quasar is left by its own lanes
        row = ((ql * 2 + 1) * S + (S - 1)) * stride             # the quasar's last row (kind 1, sample S - 1)
        end = (row + p_max + 1) * 8
        if end > worst_end:
            worst_end, worst = end, dict(quasar=ql, steps=steps, p_max=p_max, row_start_bytes=row * 8)
    return dict(stride=stride, alloc_bytes=alloc_bytes, worst_end_bytes=worst_end, worst=worst,
                beyond_row=worst["p_max"] >= stride, beyond_allocation_bytes=max(0, worst_end - alloc_bytes),
                allocation_ends_on_page=(alloc_bytes % 4096 == 0), rows_left_by_their_lanes=overrun_rows)


def report(name, pixel_counts, S):
    a = model(pixel_counts, S, clamped=True)
    b = model(pixel_counts, S, clamped=False)
    c = model(pixel_counts, S, clamped=False, pad=12)
    print(f"{name}: pixels {list(pixel_counts)}, S = {S}, stride {a['stride']} doubles, table {a['alloc_bytes']} B"
          f"{' (ends on a 4-KiB page)' if a['allocation_ends_on_page'] else ''}")
    print(f"   shipped (clamped)      : largest index {a['worst']['p_max']:5d}  beyond the row: {a['beyond_row']}  "
          f"bytes beyond the table: {a['beyond_allocation_bytes']}")
    print(f"   unclamped, same stride : largest index {b['worst']['p_max']:5d}  beyond the row: {b['beyond_row']}  "
          f"bytes beyond the table: {b['beyond_allocation_bytes']}")
    print(f"   unclamped, rows + 12   : largest index {c['worst']['p_max']:5d}  beyond the row: {c['beyond_row']}  "
          f"bytes beyond the table: {c['beyond_allocation_bytes']}  (stride {c['stride']})")
    return a, b, c


def main():
    ok = True
    # tests/test_gpu_multi.py:56 test_golden_multi_spectrum_with_supplied_indices: one quasar, S = 256
    try:
        import os
        g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                 "spectrum_multi.npz"))
        a, b, c = report("golden multi spectrum (the test both aborted runs died in)", [g["wavelengths"].size], 256)
        ok &= a["beyond_allocation_bytes"] == 0 and not a["beyond_row"]
    except OSError as e:  # noqa: PERF203
        print("golden fixture not found:", e)
    a, b, c = report("test_gpu_resampling_then_oracle", [320, 211, 402], 160)
    ok &= a["beyond_allocation_bytes"] == 0 and not a["beyond_row"]
    # tools/fuzz_multi.py: its 48 shapes (same generator, same seed)
    rng = np.random.default_rng(5)
    bad_unclamped = bad_padded = 0
    for trial in range(48):
        rng.integers(1, 5)
        rng.integers(21, 41) if trial % 2 else rng.integers(1, 21)
        n = int(rng.integers(60, 700))
        S = int(rng.integers(16, 150))
        rng.uniform(0, 0.15)
        a = model([n], S, clamped=True)
        b = model([n], S, clamped=False)
        c = model([n], S, clamped=False, pad=12)
        ok &= a["beyond_allocation_bytes"] == 0 and not a["beyond_row"]
        bad_unclamped += b["beyond_allocation_bytes"] > 0
        bad_padded += c["beyond_allocation_bytes"] > 0 or c["beyond_row"]
    print(f"fuzz_multi shapes: shipped index inside its row and the table in all 48; unclamped at the same stride "
          f"leaves the TABLE in {bad_unclamped} of 48; unclamped with rows padded by 12 entries leaves row or table in {bad_padded}")
    # exhaustive over the pixel counts one quasar can have: the shipped index never leaves its row
    for n in range(1, 5000):
        a = model([n], 7, clamped=True)
        ok &= not a["beyond_row"] and a["beyond_allocation_bytes"] == 0
    print("shipped index stays inside its row for every pixel count 1..4999:", ok)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
