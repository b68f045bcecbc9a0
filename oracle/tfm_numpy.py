"""NumPy restatement of the travel-time-table consumers (TEST INFRASTRUCTURE — see oracle/__init__.py): transmit focal
laws and a total-focusing-method delay-and-sum over full-matrix-capture data.  The reference has neither (it stops at
the travel times, main_rt.py:497-504): these functions define what csrc/rtus_tfm.hip is checked against, together with
a synthetic point-scatterer FMC generator."""
import numpy as np


def focal_delays(tt):
    """delays[e, f] = nanmax_e' tt[e', f] - tt[e, f]; NaN stays NaN, an all-NaN column is all NaN."""
    tt = np.asarray(tt, dtype=np.float64)
    with np.errstate(all="ignore"):
        m = np.where(np.isnan(tt).all(axis=0), np.nan, np.nanmax(np.where(np.isnan(tt), -np.inf, tt), axis=0))
    return m[None, :] - tt


def tfm(fmc, fs, t0, tt_tx, tt_rx):
    """image[f] = sum_{tx, rx} fmc[tx, rx] interpolated linearly at sample (tt_tx[tx, f] + tt_rx[rx, f] - t0) * fs;
    samples outside the record are zero, NaN travel times contribute nothing.  float64 accumulation."""
    fmc = np.asarray(fmc, dtype=np.float32)
    n_tx, n_rx, n_t = fmc.shape
    tt_tx = np.asarray(tt_tx, dtype=np.float64)
    tt_rx = np.asarray(tt_rx, dtype=np.float64)
    n_f = tt_tx.shape[1]
    img = np.zeros(n_f)
    pad = np.concatenate([fmc.astype(np.float64), np.zeros((n_tx, n_rx, 1))], axis=2)   # sample n_t = 0
    for tx in range(n_tx):
        s = (tt_tx[tx][None, :] + tt_rx - t0) * fs                      # [n_rx, n_f]
        ok = np.isfinite(s) & (s >= 0) & (s < n_t)
        i = np.where(ok, np.floor(s), 0).astype(np.int64)
        w = np.where(ok, s - i, 0.0)
        rows = np.arange(n_rx)[:, None]
        v0 = pad[tx][rows, i]
        v1 = pad[tx][rows, i + 1]
        img += np.where(ok, v0 + w * (v1 - v0), 0.0).sum(axis=0)
    return img


def synth_fmc(x_el, z_el, scatterers, c, fs, n_t, t0=0.0, f0=5e6, cycles=2.5):
    """Straight-ray FMC of point scatterers in a homogeneous medium: Gaussian-windowed tone bursts at the two-way times.
    -> fmc float32 [n_el, n_el, n_t]"""
    x_el, z_el = np.asarray(x_el, dtype=np.float64), np.asarray(z_el, dtype=np.float64)
    t = t0 + np.arange(n_t) / fs
    fmc = np.zeros((x_el.size, x_el.size, n_t), dtype=np.float64)
    sig = cycles / f0 / 2.355
    for xs, zs, amp in scatterers:
        d = np.hypot(x_el - xs, z_el - zs) / c                          # one-way times
        tau = d[:, None] + d[None, :]                                   # [tx, rx]
        dt = t[None, None, :] - tau[:, :, None]
        fmc += amp * np.exp(-0.5 * (dt / sig) ** 2) * np.cos(2 * np.pi * f0 * dt)
    return fmc.astype(np.float32)
