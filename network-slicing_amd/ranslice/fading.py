"""Synthetic frequency-selective fading traces.

The reference reads three ns-3 style traces, `datasets/fading_trace_{EPA_3kmph,
ETU_3kmph,EVA_60kmph}.csv` (reference channel_models.py:29-33, 141-150): headerless CSV,
100 rows (PRBs) x T columns (time samples), values in dB, NaNs tolerated.  Those files
are absent from the reference checkout (.MISSING_LARGE_BLOBS), so this build ships a
seeded generator with the same layout.  Real traces can be loaded with `load_csv`.

Layout returned here is the reference's: float64 [rows=PRB][cols=time].  The C-ABI
(`rs_load_fading`) transposes to [time][PRB] for the device.
"""
import numpy as np

TRACE_NAMES = ("EPA_3kmph", "ETU_3kmph", "EVA_60kmph")
# (number of multipath taps, time-domain rate per sample, delay spread in PRB^-1 units)
_PROFILE = ((7, 0.004, 0.010), (9, 0.004, 0.045), (9, 0.080, 0.025))


def synth_fading(trace_id, n_cols, seed=20240, n_rows=100, nan_cols=()):
    """Sum-of-sinusoids Rayleigh-like gain in dB, quantised to 1e-4 dB."""
    taps, ft, fd = _PROFILE[trace_id]
    rng = np.random.default_rng([int(seed), int(trace_id)])
    t = np.arange(n_cols, dtype=np.float64)[None, :]
    f = np.arange(n_rows, dtype=np.float64)[:, None]
    re = np.zeros((n_rows, n_cols))
    im = np.zeros((n_rows, n_cols))
    for _ in range(taps):
        amp = rng.exponential(1.0)
        w_t = ft * rng.uniform(-1.0, 1.0)
        w_f = fd * rng.uniform(0.0, 1.0)
        ph = rng.uniform(0.0, 2.0 * np.pi)
        arg = 2.0 * np.pi * (w_t * t + w_f * f) + ph
        re += amp * np.cos(arg)
        im += amp * np.sin(arg)
    power = (re * re + im * im) / taps
    db = 10.0 * np.log10(np.maximum(power, 1e-6)) + 0.25 * rng.standard_normal((n_rows, n_cols))
    db = np.round(db, 4)
    for c in nan_cols:
        db[int(rng.integers(n_rows)), int(c)] = np.nan
    return np.ascontiguousarray(db)


# ---- second profile: tapped-delay-line traces in the style of the ns-3 LTE fading traces the reference was run on
# (fading_trace_{EPA,ETU,EVA}: 100 RBs x 10,000 samples of 1 ms, produced by ns-3's fading-trace-generator from the
# 3GPP TS 36.104 Annex B.2 delay profiles).  Each tap is an independent Rayleigh process with a Jakes Doppler spectrum
# (sum of sinusoids), the RB gain is |sum_l sqrt(P_l) g_l(t) exp(-j 2 pi f_RB tau_l)|^2 with unit mean power.  The
# first profile (synth_fading) is kept as it is: the golden fixtures depend on it.
_TDL = {
    0: ((0, 30, 70, 90, 110, 190, 410), (0.0, -1.0, -2.0, -3.0, -8.0, -17.2, -20.8), 3.0),            # EPA, 3 km/h
    1: ((0, 50, 120, 200, 230, 500, 1600, 2300, 5000), (-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -3.0, -5.0, -7.0), 3.0),  # ETU
    2: ((0, 30, 150, 310, 370, 710, 1090, 1730, 2510), (0.0, -1.5, -1.4, -3.6, -0.6, -9.1, -7.0, -12.0, -16.9), 60.0),  # EVA
}


def synth_fading_tdl(trace_id, n_cols, seed=20240, n_rows=100, carrier_ghz=2.0, n_osc=24, nan_cols=()):
    """Frequency-selective Rayleigh fading gains in dB, [rows = RB (180 kHz apart)][cols = 1-ms samples], quantised to
    1e-4 dB like the CSV exports.  Doppler f_d = v f_c / c (5.6 Hz at 3 km/h, 111 Hz at 60 km/h)."""
    delays_ns, powers_db, kmph = _TDL[trace_id]
    rng = np.random.default_rng([int(seed), 7, int(trace_id)])
    fd = kmph / 3.6 * carrier_ghz * 1e9 / 299792458.0
    t = np.arange(n_cols, dtype=np.float64) * 1e-3
    f = (np.arange(n_rows, dtype=np.float64) - (n_rows - 1) / 2.0) * 180e3
    p = 10.0 ** (np.asarray(powers_db) / 10.0)
    p = p / p.sum()
    h = np.zeros((n_rows, n_cols), dtype=np.complex128)
    for tau, pw in zip(delays_ns, p):
        # Jakes spectrum: n_osc arrivals with uniform angles, random phases (Zheng-Xiao style sum of sinusoids)
        alpha = (2.0 * np.pi * np.arange(1, n_osc + 1) - np.pi + rng.uniform(-np.pi, np.pi)) / (4.0 * n_osc)
        ph_i = rng.uniform(-np.pi, np.pi, n_osc)
        ph_q = rng.uniform(-np.pi, np.pi, n_osc)
        w = 2.0 * np.pi * fd * t[None, :]
        gi = np.cos(w * np.cos(alpha)[:, None] + ph_i[:, None]).sum(axis=0)
        gq = np.cos(w * np.sin(alpha)[:, None] + ph_q[:, None]).sum(axis=0)
        g = (gi + 1j * gq) / np.sqrt(n_osc)          # E|g|^2 = 1
        h += np.sqrt(pw) * np.exp(-2j * np.pi * f[:, None] * (tau * 1e-9)) * g[None, :]
    power = h.real ** 2 + h.imag ** 2
    db = np.round(10.0 * np.log10(np.maximum(power, 1e-12)), 4)
    for c in nan_cols:
        db[int(rng.integers(n_rows)), int(c)] = np.nan
    return np.ascontiguousarray(db)


PROFILES = {'sos': synth_fading, 'tdl': synth_fading_tdl}


def synth_traces(n_cols, profile='sos', seed=20240):
    """the three traces of SINRSelectiveFading from one of the seeded generators"""
    return [PROFILES[profile](t, n_cols, seed=seed) for t in range(3)]


def extend_rows(table, n_prbs):
    """Row extension used when the cell has more than 100 PRBs: wrap the first rows
    (reference channel_models.py:144-148)."""
    rows = table.shape[0]
    if n_prbs > rows:
        table = np.vstack((table, table[0:n_prbs - rows, :]))
    return np.ascontiguousarray(table)


def load_csv(path):
    """Load a real trace in the reference's CSV layout (channel_models.py:143: pd.read_csv(header=None)): headerless,
    comma separated, rows = PRB, columns = time samples, dB; empty / 'nan' fields become NaN (the walker skips columns
    that contain one, channel_models.py:188-189)."""
    t = np.genfromtxt(path, delimiter=",", dtype=np.float64)
    if t.ndim != 2:
        raise ValueError('%s: expected a 2-D table (rows = PRB, columns = time)' % path)
    return np.ascontiguousarray(t)


def load_fad(path, n_rows=100):
    """Load an ns-3 LTE fading trace (`fading_trace_*.fad`, the files the reference's CSVs were exported from):
    whitespace-separated dB values, RB-major -- all time samples of RB 0, then RB 1, ... (ns-3's
    TraceFadingLossModel::LoadTrace reads m_rbNum x m_samplesNum in that order; 100 RBs).  Returns the reference's
    layout, float64 [rows = PRB][cols = time]; 'nan' tokens are kept as NaN."""
    with open(path) as f:
        vals = np.array(f.read().split(), dtype=np.float64)
    if vals.size == 0 or vals.size % n_rows:
        raise ValueError('%s: %d values are not a multiple of %d RBs' % (path, vals.size, n_rows))
    return np.ascontiguousarray(vals.reshape(n_rows, vals.size // n_rows))


def load_traces(paths):
    """The three traces of SINRSelectiveFading (EPA_3kmph, ETU_3kmph, EVA_60kmph: channel_models.py:29-33) from
    files, .csv or .fad by extension, ready for VecRanSlice(fading=...) / scenario_creator.set_fading."""
    out = []
    for p in paths:
        out.append(load_fad(p) if str(p).lower().endswith('.fad') else load_csv(p))
    if len(out) != 3:
        raise ValueError('three traces are needed')
    return out
