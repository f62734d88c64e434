#!/usr/bin/env python3
"""CPU-only numerical study for BASELINE config 5: what would accumulating the
DEVIATION from the null model, B - B_null and v - v_null, on the fp32 matrix cores buy over
accumulating B and v themselves (the shipped study variant)?

Emulates the fp32 contraction in NumPy -- operands rounded to fp32, products and the running sum
over K-steps of 4 pixels in fp32, as v_mfma_f32_16x16x4_f32 accumulates -- for one synthetic quasar
(n = 1500, k = 40) and 48 samples spread over log N_HI = 20..23, everything else (profile, weights,
quadratic form, log-determinant, Cholesky) in fp64.  Prints the error of the sample log-likelihoods
of (a) direct fp32 accumulation, (b) fp32 accumulation of the deviation + fp64 null-model B0, v0,
(c) as (b) with the B operand kept in fp64 (deviation weights alone rounded: the floor of (b)),
against the all-fp64 value.  Uses the oracle's Voigt profile (test infrastructure; nothing here is
product code)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_dla_detection_amd import synthetic  # noqa: E402
from gp_dla_detection_amd.parameters import Parameters  # noqa: E402
from oracle import oracle  # noqa: E402


def chol_ll(B, v, quad0, logd, n):
    L = np.linalg.cholesky(B)
    z = np.linalg.solve(L, v)
    return -0.5 * ((quad0 - z @ z) + logd + 2 * np.log(np.diag(L)).sum() + n * np.log(2 * np.pi))


def fp32_contract(A, P):
    """sum_p A[p] * P[p, :] with operands and accumulator in fp32, 4 pixels per step"""
    A32, P32 = A.astype(np.float32), P.astype(np.float32)
    acc = np.zeros(P.shape[1], dtype=np.float32)
    for t in range(0, A.size, 4):
        acc = acc + (A32[t:t + 4, None] * P32[t:t + 4]).sum(axis=0, dtype=np.float32)
    return acc.astype(np.float64)


def fp32_contract_split(A, P):
    """as fp32_contract with the B operand carried as an fp32 pair P_hi + P_lo (two MFMAs per
    tile into the same fp32 accumulator)"""
    A32 = A.astype(np.float32)
    hi = P.astype(np.float32)
    lo = (P - hi.astype(np.float64)).astype(np.float32)
    acc = np.zeros(P.shape[1], dtype=np.float32)
    for t in range(0, A.size, 4):
        acc = acc + (A32[t:t + 4, None] * hi[t:t + 4]).sum(axis=0, dtype=np.float32)
        acc = acc + (A32[t:t + 4, None] * lo[t:t + 4]).sum(axis=0, dtype=np.float32)
    return acc.astype(np.float64)


def main():
    k, n, S = 40, 1500, 48
    p = Parameters()
    model = synthetic.make_model(k)
    sp = synthetic.make_spectrum(4242, n, model)
    wl, z_qso = sp["wavelengths"], sp["z_qso"]
    rest = wl / (1 + z_qso)
    keep = (rest >= p.min_lambda) & (rest <= p.max_lambda)
    wl, rest, y, nv = wl[keep], rest[keep], sp["flux"][keep], sp["noise_variance"][keep]
    grid = model["rest_wavelengths"]
    mu = np.interp(rest, grid, model["mu"])
    M = np.stack([np.interp(rest, grid, model["M"][:, c]) for c in range(k)], 1)
    lya_z = (wl - p.lya_wavelength) / p.lya_wavelength
    sc = 1 - np.exp(-np.exp(model["log_tau_0"]) * (1 + lya_z) ** np.exp(model["log_beta"])) + np.exp(model["log_c_0"])
    om = np.exp(2 * np.interp(rest, grid, model["log_omega"])) * sc ** 2
    iu = np.tril_indices(k)
    P = M[:, iu[0]] * M[:, iu[1]]                        # vech(m m')
    zmin, zmax = p.min_z_dla(wl, z_qso), p.max_z_dla(wl, z_qso)
    ps = p.pixel_spacing
    lo, hi = np.log10(wl.min()), np.log10(wl.max())
    pad = np.concatenate([10 ** np.array([lo - 3 * ps, lo - 2 * ps, lo - ps]), wl,
                          10 ** np.array([hi + ps, hi + 2 * ps, hi + 3 * ps])])
    rng = np.random.default_rng(3)
    w0 = 1.0 / (om + nv)
    u0 = (y - mu) * w0
    B0 = np.zeros((k, k))
    B0[iu] = w0 @ P
    v0 = u0 @ M
    rows = []
    for i in range(S):
        zd = zmin + (zmax - zmin) * rng.uniform()
        logn = 20 + 3 * i / (S - 1)
        a = oracle.voigt(pad, zd, 10.0 ** logn, 3)
        r = y - a * mu
        d = om * a * a + nv
        w, u = a * a / d, a * r / d
        quad0, logd = (r * r / d).sum(), np.log(d).sum()

        def ll_of(bvec, v):
            B = np.zeros((k, k))
            B[iu] = bvec
            B = B + np.tril(B, -1).T + np.eye(k)
            return chol_ll(B, v, quad0, logd, wl.size)

        exact = ll_of(w @ P, u @ M)
        direct = ll_of(fp32_contract(w, P), fp32_contract(u, M))
        dev = ll_of(B0[iu] + fp32_contract(w - w0, P), v0 + fp32_contract(u - u0, M))
        dev_p64 = ll_of(B0[iu] + (w - w0).astype(np.float32).astype(np.float64) @ P,
                        v0 + (u - u0).astype(np.float32).astype(np.float64) @ M)
        frac = float(np.abs(w - w0).sum() / w0.sum())
        dev_split = ll_of(B0[iu] + fp32_contract_split(w - w0, P), v0 + fp32_contract_split(u - u0, M))
        dir_split = ll_of(fp32_contract_split(w, P), fp32_contract_split(u, M))
        rows.append((logn, exact, direct - exact, dev - exact, dev_p64 - exact, frac, dev_split - exact,
                     dir_split - exact))
    rows = np.array(rows)
    print("log N   log-likelihood   direct fp32   deviation fp32   deviation, fp64 B operand   sum|dw|/sum w0")
    for r in rows[::4]:
        print("%5.2f  %14.3f  %+11.3e  %+13.3e  %+13.3e  %10.3f" % tuple(r[:6]))
    print("max |error| (nat): direct %.3e, deviation %.3e, deviation with fp64 B operand %.3e"
          % (np.abs(rows[:, 2]).max(), np.abs(rows[:, 3]).max(), np.abs(rows[:, 4]).max()))
    print("with the B operand as an fp32 hi/lo pair (two MFMAs per tile, fp32 accumulator): deviation %.3e, "
          "direct %.3e" % (np.abs(rows[:, 6]).max(), np.abs(rows[:, 7]).max()))


if __name__ == "__main__":
    main()
