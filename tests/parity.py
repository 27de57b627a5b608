"""Shared comparison helpers for the HIP-vs-oracle / HIP-vs-golden parity tests."""
import numpy as np

FIELDS = ("f", "mag", "ph", "realph", "binno")


def peak_rows_equal(a_binno, a_f, b_binno, b_f):
    """Boolean per frame: same emitted peak bins in the same slots."""
    return np.all((a_binno == b_binno) & ((a_f > 0) == (b_f > 0)), axis=1)


def compare_analysis(got, ref, nfft, hop, sr):
    """Error summary of an analysis result against a reference result (dicts with FIELDS, totalmag).

    Frames whose peak sets differ are counted, the rest are compared value by value.  Errors are
    reported (a) raw and (b) normalised by what a complex-bin perturbation eps relative to the
    frame's largest magnitude would cause: phase error ~ eps*max/|X_k|, so `ph_norm` etc. are in
    units of that eps."""
    same = peak_rows_equal(got["binno"], got["f"], ref["binno"], ref["f"])
    nref = int((ref["f"] > 0).sum())
    # peaks of the reference that the result misses or misplaces
    bad_peaks = int((ref["f"][~same] > 0).sum())
    out = dict(frames=len(same), frames_diff=int((~same).sum()), ref_peaks=nref, bad_peaks=bad_peaks)
    v = (ref["f"] > 0) & same[:, None]
    if v.any():
        rmag = ref["mag"][v]
        fmax = np.broadcast_to(ref["mag"].max(axis=1, keepdims=True), ref["mag"].shape)[v]
        w = np.maximum(fmax, 1e-300) / np.maximum(rmag, 1e-300)      # >= 1
        dt = hop / float(sr)
        df = np.abs(got["f"][v] - ref["f"][v])
        dph = np.abs(got["ph"][v] - ref["ph"][v])
        drp = np.abs(got["realph"][v] - ref["realph"][v])
        dm = np.abs(got["mag"][v] - rmag)
        out.update(f_abs=df.max(), ph_abs=dph.max(), realph_abs=drp.max(), mag_rel=(dm / rmag).max(),
                   f_norm=(df * (2 * np.pi * dt) / w).max(), ph_norm=(dph / w).max(),
                   realph_norm=(drp / w).max(), mag_norm=(dm / fmax).max())
    else:
        out.update(f_abs=0.0, ph_abs=0.0, realph_abs=0.0, mag_rel=0.0, f_norm=0.0, ph_norm=0.0,
                   realph_norm=0.0, mag_norm=0.0)
    tm_ref = np.asarray(ref["totalmag"], dtype=np.float64)
    tm = np.asarray(got["totalmag"], dtype=np.float64)
    out["totalmag_rel"] = float(np.max(np.abs(tm - tm_ref) / np.maximum(tm_ref, 1e-300))) if len(tm) else 0.0
    return out


def peak_error_shares(got, ref, hop, sr, f_tol=2e-5, realph_tol=2e-5):
    """Share of the reference's peaks (in frames with the same peak set) whose normalised frequency / unwrapped-phase errors
    are within the float32 tolerances.  f and realph hang on the phase DIFFERENCE to the previous frame's bin, so a peak
    over a weak previous bin (noise, recordings) is ill-conditioned in a way the frame-maximum normalisation does not see:
    on such material the maxima over 400 000 peaks are outliers and the shares are the meaningful figure."""
    same = peak_rows_equal(got["binno"], got["f"], ref["binno"], ref["f"])
    v = (ref["f"] > 0) & same[:, None]
    if not v.any():
        return dict(f=1.0, realph=1.0, peaks=0)
    rmag = ref["mag"][v]
    fmax = np.broadcast_to(ref["mag"].max(axis=1, keepdims=True), ref["mag"].shape)[v]
    w = np.maximum(fmax, 1e-300) / np.maximum(rmag, 1e-300)
    dt = hop / float(sr)
    fn = np.abs(got["f"][v] - ref["f"][v]) * (2 * np.pi * dt) / w
    rn = np.abs(got["realph"][v] - ref["realph"][v]) / w
    return dict(f=float((fn <= f_tol).mean()), realph=float((rn <= realph_tol).mean()), peaks=int(v.sum()))


def pv_result(p):
    return dict(f=p.f, mag=p.mag, ph=p.ph, realph=p.realph, binno=p.binno, t=p.t,
                totalmag=np.asarray(p.totalmag))
