"""The identity behind the engine's coarse-grid path, checked in double precision with plain numpy (CPU only).

Per source point the field is E_s(q) = sum_{|k| <= h} A_s[k] w_N^(k q), h = pn/4, N = 2 pn, so the
intensity I(q) = sum_s |E_s(q)|^2 has Fourier coefficients C[kappa] only for |kappa| <= 2h = pn/2: pn samples per period
-- the coarse grid q = 2 v, where E_s(2v) = sum_k A_s[k] w_pn^(k v) is a pn-point transform -- determine it, except that
kappa = +pn/2 and -pn/2 alias there.  Those coefficients come only from products of opposite edges of the support box,
    Gx[kappa_y] = sum_s sum_ky A_s[ky, +h] conj(A_s[ky - kappa_y, -h])      (and Gy with rows),
and the part of I that the band-limited interpolation of the coarse samples misses is
    dI[qy, qx] = Re(2 i^qx G(qy)) for odd qx + Re(2 i^qy H(qx)) for odd qy,    G(q) = sum_kappa Gx[kappa] w_N^(kappa q).
The HIP engine implements exactly this (csrc/abbe_engine.hip: reconstruct_plane, k_nyquist_*); the GPU suite checks it
against the direct path and the reference's golden images."""
import numpy as np


def _cross(u, w):
    n = len(u)
    return {kk: sum(u[i] * np.conj(w[i - kk]) for i in range(n) if 0 <= i - kk < n) for kk in range(-(n - 1), n)}


def test_coarse_grid_reconstruction_is_exact():
    rng = np.random.default_rng(1)
    pn = 64; N = 2 * pn; c = pn // 2; h = pn // 4; S = 5
    k = np.arange(-h, h + 1)
    KY, KX = np.meshgrid(k, k, indexing="ij")
    support = (KX ** 2 + KY ** 2) <= h * h + 9            # a disk whose rim touches the box edges over several pixels
    support[0, 0] = support[0, -1] = support[-1, 0] = support[-1, -1] = False      # the engine requires empty corners
    A = (rng.standard_normal((S, 2 * h + 1, 2 * h + 1)) + 1j * rng.standard_normal((S, 2 * h + 1, 2 * h + 1))) * support
    q = np.arange(-c, c)
    Wf = np.exp(2j * np.pi * np.outer(k, q) / N)
    I_true = sum(np.abs(Wf.T @ A[s] @ Wf) ** 2 for s in range(S))                  # the reference's fine-grid sum
    v = np.arange(-pn // 2, pn // 2)
    Wc = np.exp(2j * np.pi * np.outer(k, v) / pn)
    I_c = sum(np.abs(Wc.T @ A[s] @ Wc) ** 2 for s in range(S))                     # pn-point transforms, grid q = 2 v
    kap = np.arange(-pn // 2, pn // 2)
    Fc = np.exp(-2j * np.pi * np.outer(kap, v) / pn)
    Chat = Fc @ I_c @ Fc.T / pn ** 2
    Wr = np.exp(2j * np.pi * np.outer(kap, q) / N)
    I_rec = (Wr.T @ Chat @ Wr).real                                                # band-limited interpolation
    assert np.abs(I_rec - I_true).max() / I_true.max() > 1e-4                      # ... is NOT enough on its own
    Gx, Gy = {}, {}
    for s in range(S):
        for dst, u, w in ((Gx, A[s][:, -1], A[s][:, 0]), (Gy, A[s][-1, :], A[s][0, :])):
            for kk, val in _cross(u, w).items():
                dst[kk] = dst.get(kk, 0) + val
    G = np.array([sum(g * np.exp(2j * np.pi * kk * qq / N) for kk, g in Gx.items()) for qq in q])
    H = np.array([sum(g * np.exp(2j * np.pi * kk * qq / N) for kk, g in Gy.items()) for qq in q])
    iq = 1j ** (q % 4)
    odd = (q % 2 != 0)
    dI = np.real(2 * G[:, None] * (iq * odd)[None, :]) + np.real(2 * H[None, :] * (iq * odd)[:, None])
    assert np.abs(I_rec + dI - I_true).max() / I_true.max() < 1e-12
