"""ctypes binding of include/tdoa_mi355x.h.  No CPU fallback: loading fails loudly if the
HIP library is missing, and every compute call raises TdoaError on a non-zero status."""
import ctypes as C
import os

import numpy as np

from . import build as _build

OK = 0
KERNELS = ["STATS", "FWD_COL", "FWD_ROW", "INV_ROW", "INV_COL", "PEAK"]


class TdoaError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("tdoa_mi355x status %d: %s" % (status, msg))
        self.status = status


class Params(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("max_lag", C.c_int32), ("corr_block", C.c_int32),
                ("weak_threshold", C.c_double), ("window_len", C.c_int64),
                ("device", C.c_int32), ("windows_per_batch", C.c_int32),
                ("k1_smooth", C.c_int32), ("k1_gate", C.c_int32), ("lag_mode", C.c_int32), ("reserved", C.c_int32)]


LAGS_SIGNED, LAGS_GO = 0, 1


class Peak(C.Structure):
    _fields_ = [("lag", C.c_int32), ("abs_corr", C.c_float), ("corr", C.c_double)]


PEAK_DTYPE = np.dtype([("lag", np.int32), ("abs_corr", np.float32), ("corr", np.float64)])


QUALITY_DTYPE = np.dtype([("n_samples", np.int64), ("i_avg", np.float64), ("q_avg", np.float64), ("i_std", np.float64),
                          ("q_std", np.float64), ("power_level", np.float64), ("mean_power", np.float64),
                          ("i_min", np.int32), ("i_max", np.int32), ("q_min", np.int32), ("q_max", np.int32),
                          ("has_clipping", np.int32), ("has_overload", np.int32)])


class FinePeak(C.Structure):
    _fields_ = [("delay", C.c_double), ("frac", C.c_float), ("y", C.c_float * 3), ("plausible", C.c_int32),
                ("reserved", C.c_int32)]


FINE_DTYPE = np.dtype([("delay", np.float64), ("frac", np.float32), ("y", np.float32, (3,)), ("plausible", np.int32),
                       ("reserved", np.int32)])


class FastAnalysis(C.Structure):
    _fields_ = [("total_samples", C.c_int32), ("has_clipping", C.c_int32), ("has_overload", C.c_int32),
                ("reserved", C.c_int32), ("i_avg", C.c_double), ("q_avg", C.c_double), ("i_std", C.c_double),
                ("q_std", C.c_double), ("snr_estimate", C.c_double), ("power_level", C.c_double)]


class FmStats(C.Structure):
    _fields_ = [("s1", C.c_int64), ("s2_lo", C.c_uint64), ("s2_hi", C.c_uint64),
                ("mean", C.c_float), ("scale", C.c_float)]


# every symbol include/tdoa_mi355x.h declares
SYMBOLS = [
    "tdoa_default_params", "tdoa_create", "tdoa_destroy", "tdoa_strerror", "tdoa_last_error",
    "tdoa_abi_version", "tdoa_device_count",
    "tdoa_load_iq_u8", "tdoa_preprocess_c64", "tdoa_time_domain_correlation_c64",
    "tdoa_cross_correlate_c64", "tdoa_simple_correlate_c64", "tdoa_fast_snr_u8",
    "tdoa_fast_analyze_u8", "tdoa_fast_analyze_capture_u8",
    "tdoa_capture_upload", "tdoa_capture_upload_file", "tdoa_capture_attach_device", "tdoa_capture_clear",
    "tdoa_synth_capture", "tdoa_synth_weak_capture", "tdoa_capture_download", "tdoa_capture_upload_range",
    "tdoa_num_windows", "tdoa_num_pairs", "tdoa_process", "tdoa_process_u8",
    "tdoa_process_fine", "tdoa_fm_xcorr_fine_u8", "tdoa_window_quality_all", "tdoa_window_quality_u8",
    "tdoa_fm_xcorr_u8", "tdoa_fm_preprocess_u8", "tdoa_fm_xcorr_lags_u8", "tdoa_debug_force_generic",
    "tdoa_debug_flags", "tdoa_debug_last_k1", "tdoa_debug_graph_info", "tdoa_debug_segment_quads", "tdoa_debug_staged_groups", "tdoa_cross_correlate_batch_c64",
    "tdoa_latlon_to_ecef", "tdoa_ecef_to_latlon", "tdoa_solve_3station", "tdoa_solve_nstation", "tdoa_solve_surface",
    "tdoa_profile_enable", "tdoa_profile_select", "tdoa_profile_reset", "tdoa_profile_get", "tdoa_kernel_name",
    "tdoa_plan_info",
]

_lib = None


def library_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load the shared library (building it with hipcc if absent).  Raises if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and _build.needs_build():
        _build.build()
    if not os.path.exists(_build.LIB):
        raise RuntimeError("libtdoa_mi355x.so is missing; run tdoa_amd.build.build() (needs hipcc)")
    L = C.CDLL(_build.LIB)
    vp, sz = C.c_void_p, C.c_size_t
    fp, u8p, dp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
    i32p = C.POINTER(C.c_int32)
    L.tdoa_default_params.argtypes = [C.POINTER(Params)]
    L.tdoa_default_params.restype = None
    L.tdoa_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.tdoa_destroy.argtypes = [vp]
    L.tdoa_destroy.restype = None
    L.tdoa_strerror.argtypes = [C.c_int]
    L.tdoa_strerror.restype = C.c_char_p
    L.tdoa_last_error.argtypes = [vp]
    L.tdoa_last_error.restype = C.c_char_p
    L.tdoa_kernel_name.argtypes = [C.c_int]
    L.tdoa_kernel_name.restype = C.c_char_p
    L.tdoa_load_iq_u8.argtypes = [vp, u8p, sz, fp]
    L.tdoa_preprocess_c64.argtypes = [vp, fp, sz, fp, C.POINTER(C.c_int)]
    L.tdoa_time_domain_correlation_c64.argtypes = [vp, fp, sz, fp, sz, C.c_int, i32p, dp]
    L.tdoa_cross_correlate_c64.argtypes = [vp, fp, sz, fp, sz, i32p, dp]
    L.tdoa_cross_correlate_batch_c64.argtypes = [vp, C.POINTER(fp), C.POINTER(sz), C.c_int, i32p, dp]
    L.tdoa_simple_correlate_c64.argtypes = [vp, fp, sz, fp, sz, i32p, fp]
    L.tdoa_fast_snr_u8.argtypes = [vp, u8p, C.c_int, dp]
    L.tdoa_fast_analyze_u8.argtypes = [vp, u8p, C.c_int, C.POINTER(FastAnalysis)]
    L.tdoa_fast_analyze_capture_u8.argtypes = [vp, u8p, sz, C.POINTER(FastAnalysis), C.POINTER(FastAnalysis)]
    L.tdoa_capture_upload.argtypes = [vp, C.c_int, u8p, sz]
    L.tdoa_capture_upload_file.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(sz)]
    L.tdoa_capture_attach_device.argtypes = [vp, C.c_int, vp, sz]
    L.tdoa_capture_clear.argtypes = [vp]
    L.tdoa_synth_capture.argtypes = [vp, C.c_int, sz, C.c_double, C.c_double, C.c_double, dp, dp, C.c_double,
                                     C.c_uint64]
    L.tdoa_synth_weak_capture.argtypes = [vp, C.c_int, sz, C.c_double, C.c_double, dp, dp, C.c_double, C.c_double,
                                          C.c_uint64]
    L.tdoa_capture_upload_range.argtypes = [vp, C.c_int, sz, sz, u8p, sz]
    L.tdoa_capture_download.argtypes = [vp, C.c_int, sz, sz, u8p]
    L.tdoa_num_windows.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tdoa_num_pairs.argtypes = [vp]
    L.tdoa_process.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.tdoa_process_u8.argtypes = [vp, C.POINTER(u8p), C.POINTER(sz), C.c_int, vp]
    L.tdoa_fm_xcorr_u8.argtypes = [vp, u8p, sz, u8p, sz, C.c_int, C.POINTER(Peak)]
    L.tdoa_window_quality_all.argtypes = [vp, C.c_int, C.c_int, vp]
    L.tdoa_window_quality_u8.argtypes = [vp, u8p, sz, vp]
    L.tdoa_process_fine.argtypes = [vp, C.c_int, C.c_int, C.c_double, vp, vp]
    L.tdoa_fm_xcorr_fine_u8.argtypes = [vp, u8p, sz, u8p, sz, C.c_int, C.c_double, C.POINTER(Peak),
                                        C.POINTER(FinePeak)]
    L.tdoa_fm_preprocess_u8.argtypes = [vp, u8p, sz, fp, C.POINTER(FmStats)]
    L.tdoa_fm_xcorr_lags_u8.argtypes = [vp, u8p, sz, u8p, sz, C.c_int, dp]
    L.tdoa_debug_force_generic.argtypes = [vp, C.c_int]
    L.tdoa_debug_flags.argtypes = [vp, C.c_uint]
    L.tdoa_debug_last_k1.argtypes = [vp, C.c_int, C.POINTER(FmStats), C.POINTER(C.c_int32)]
    L.tdoa_debug_graph_info.argtypes = [vp, C.POINTER(C.c_int32), C.c_char_p]
    L.tdoa_debug_segment_quads.argtypes = [C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int]
    L.tdoa_debug_staged_groups.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_int]
    L.tdoa_latlon_to_ecef.argtypes = [C.c_double, C.c_double, C.c_double, dp]
    L.tdoa_latlon_to_ecef.restype = None
    L.tdoa_ecef_to_latlon.argtypes = [C.c_double, C.c_double, C.c_double, dp]
    L.tdoa_ecef_to_latlon.restype = None
    L.tdoa_solve_3station.argtypes = [dp, dp, dp, C.POINTER(C.c_int)]
    L.tdoa_solve_nstation.argtypes = [dp, C.c_int, dp, dp, C.c_int, dp, C.POINTER(C.c_int)]
    L.tdoa_solve_surface.argtypes = [dp, C.c_int, dp, dp, C.c_double, dp, C.POINTER(C.c_int)]
    L.tdoa_profile_enable.argtypes = [vp, C.c_int]
    L.tdoa_profile_select.argtypes = [vp, C.c_uint]
    L.tdoa_profile_reset.argtypes = [vp]
    L.tdoa_profile_get.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_int64), dp]
    L.tdoa_plan_info.argtypes = [vp, C.POINTER(C.c_int64), i32p, i32p]
    _lib = L
    return L


def default_params():
    p = Params()
    load().tdoa_default_params(C.byref(p))
    return p


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c64(x):
    a = np.ascontiguousarray(x, dtype=np.complex64)
    return a, a.view(np.float32)


class Context:
    """One tdoa_ctx (one GPU, single caller)."""

    def __init__(self, **kw):
        self._L = load()
        p = default_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise TypeError("unknown parameter %r" % k)
            setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        rc = self._L.tdoa_create(C.byref(p), C.byref(h))
        if rc != OK:
            raise TdoaError(rc, self._L.tdoa_strerror(rc).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.tdoa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            detail = self._L.tdoa_last_error(self._h).decode()
            raise TdoaError(rc, "%s (%s)" % (self._L.tdoa_strerror(rc).decode(), detail))

    # ---- mode A ------------------------------------------------------------
    def load_iq_u8(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        n = raw.size // 2
        out = np.empty(n, dtype=np.complex64)
        self._chk(self._L.tdoa_load_iq_u8(self._h, _u8(raw), n, _f(out.view(np.float32))))
        return out

    def preprocess(self, sig):
        a, v = _c64(sig)
        out = np.empty_like(a)
        weak = C.c_int()
        self._chk(self._L.tdoa_preprocess_c64(self._h, _f(v), a.size, _f(out.view(np.float32)), C.byref(weak)))
        return out, bool(weak.value)

    def time_domain_correlation(self, s1, s2, max_lag):
        a, va = _c64(s1)
        b, vb = _c64(s2)
        d, c = C.c_int32(), C.c_double()
        self._chk(self._L.tdoa_time_domain_correlation_c64(self._h, _f(va), a.size, _f(vb), b.size, int(max_lag),
                                                           C.byref(d), C.byref(c)))
        return d.value, c.value

    def cross_correlate(self, s1, s2):
        a, va = _c64(s1)
        b, vb = _c64(s2)
        d, c = C.c_int32(), C.c_double()
        self._chk(self._L.tdoa_cross_correlate_c64(self._h, _f(va), a.size, _f(vb), b.size, C.byref(d), C.byref(c)))
        return d.value, c.value

    def cross_correlate_batch(self, signals):
        """every signal preprocessed once, every pair i < j correlated (processor.go:816-850): [(delay, corr)] in pair order"""
        arrs = [_c64(x) for x in signals]
        k = len(arrs)
        ptrs = (C.POINTER(C.c_float) * k)(*[_f(v) for _, v in arrs])
        sizes = (C.c_size_t * k)(*[a.size for a, _ in arrs])
        npair = k * (k - 1) // 2
        d = (C.c_int32 * npair)()
        c = (C.c_double * npair)()
        self._chk(self._L.tdoa_cross_correlate_batch_c64(self._h, ptrs, sizes, k, d, c))
        return [(d[p], c[p]) for p in range(npair)]

    def simple_correlate(self, s1, s2):
        a, va = _c64(s1)
        b, vb = _c64(s2)
        d, c = C.c_int32(), C.c_float()
        self._chk(self._L.tdoa_simple_correlate_c64(self._h, _f(va), a.size, _f(vb), b.size, C.byref(d), C.byref(c)))
        return d.value, c.value

    def fast_snr(self, samples_u8, total_samples):
        s = np.ascontiguousarray(samples_u8, dtype=np.uint8)
        out = C.c_double()
        self._chk(self._L.tdoa_fast_snr_u8(self._h, _u8(s), int(total_samples), C.byref(out)))
        return out.value

    def fast_analyze(self, samples_u8, total_samples):
        s = np.ascontiguousarray(samples_u8, dtype=np.uint8)
        fa = FastAnalysis()
        self._chk(self._L.tdoa_fast_analyze_u8(self._h, _u8(s), int(total_samples), C.byref(fa)))
        return fa

    def fast_analyze_capture(self, raw_u8):
        """fast_analyzer's two output lines: (ref, tgt) analyses of one .dat capture."""
        s = np.ascontiguousarray(raw_u8, dtype=np.uint8)
        ref, tgt = FastAnalysis(), FastAnalysis()
        self._chk(self._L.tdoa_fast_analyze_capture_u8(self._h, _u8(s), s.size, C.byref(ref), C.byref(tgt)))
        return ref, tgt

    # ---- mode B ------------------------------------------------------------
    def capture_upload(self, station, iq_u8):
        s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
        self._chk(self._L.tdoa_capture_upload(self._h, int(station), _u8(s), s.size // 2))

    def capture_upload_range(self, station, total_samples, first_sample, iq_u8):
        """upload only samples [first_sample, first_sample + len) of a capture of total_samples (sharded ingest)"""
        s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
        self._chk(self._L.tdoa_capture_upload_range(self._h, int(station), int(total_samples), int(first_sample),
                                                    _u8(s), s.size // 2))

    def capture_upload_owned(self, station, iq_u8, rank, world, window_len):
        """upload the sample runs of `iq_u8` that tdoa_process(rank, world) reads; returns the bytes sent"""
        from . import sharding
        s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
        sent = 0
        for first, count in sharding.owned_sample_runs(rank, world, s.size // 2, window_len):
            self.capture_upload_range(station, s.size // 2, first, s[2 * first:2 * (first + count)])
            sent += 2 * count
        return sent

    def capture_upload_file(self, station, path):
        n = C.c_size_t()
        self._chk(self._L.tdoa_capture_upload_file(self._h, int(station), os.fsencode(path), C.byref(n)))
        return n.value

    def capture_attach_device(self, station, dev_ptr, n_samples):
        self._chk(self._L.tdoa_capture_attach_device(self._h, int(station), C.c_void_p(int(dev_ptr)), int(n_samples)))

    def synth_capture(self, station, block_samples, station_lle, tx_lle, seed, ref_freq=162.4e6,
                      tgt_freq=101.7e6, noise=0.01, tx_power=1000.0):
        st = np.ascontiguousarray(station_lle, dtype=np.float64)
        tx = np.ascontiguousarray(tx_lle, dtype=np.float64)
        self._chk(self._L.tdoa_synth_capture(self._h, int(station), int(block_samples), ref_freq, tgt_freq, noise,
                                             _d(st), _d(tx), tx_power, int(seed)))

    def synth_weak_capture(self, station, block_samples, station_lle, tx_lle, seed, ref_freq=162.4e6,
                           tgt_freq=92.3e6, ref_power=10.0, tgt_power=1000.0):
        """weak_signal_simulator.go capture (weak reference blocks, strong target block) generated in HBM"""
        st = np.ascontiguousarray(station_lle, dtype=np.float64)
        tx = np.ascontiguousarray(tx_lle, dtype=np.float64)
        self._chk(self._L.tdoa_synth_weak_capture(self._h, int(station), int(block_samples), ref_freq, tgt_freq,
                                                  _d(st), _d(tx), ref_power, tgt_power, int(seed)))

    def capture_download(self, station, first_sample, n_samples):
        out = np.empty(2 * int(n_samples), dtype=np.uint8)
        self._chk(self._L.tdoa_capture_download(self._h, int(station), int(first_sample), int(n_samples), _u8(out)))
        return out

    def capture_clear(self):
        self._chk(self._L.tdoa_capture_clear(self._h))

    def num_windows(self):
        a, b = C.c_int(), C.c_int()
        self._chk(self._L.tdoa_num_windows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def num_pairs(self):
        return self._L.tdoa_num_pairs(self._h)

    def num_stations(self):
        p = self.num_pairs()
        return (1 + int(round((1 + 8 * p) ** 0.5))) // 2

    def process(self, rank=0, world=1, out_dev_ptr=None, want_host=True):
        wpb, w = self.num_windows()
        p = self.num_pairs()
        out = np.zeros((w, p), dtype=PEAK_DTYPE) if want_host else None
        self._chk(self._L.tdoa_process(self._h, int(rank), int(world),
                                       out.ctypes.data_as(C.c_void_p) if want_host else None,
                                       C.c_void_p(int(out_dev_ptr)) if out_dev_ptr else None))
        return out

    def window_quality_all(self, rank=0, world=1):
        """tdoa_window_quality_all -> [W][S] QUALITY_DTYPE"""
        wpb, w = self.num_windows()
        out = np.zeros((w, self.num_stations()), dtype=QUALITY_DTYPE)
        self._chk(self._L.tdoa_window_quality_all(self._h, int(rank), int(world), out.ctypes.data_as(C.c_void_p)))
        return out

    def window_quality(self, iq_u8):
        s = np.ascontiguousarray(iq_u8, dtype=np.uint8)
        out = np.zeros(1, dtype=QUALITY_DTYPE)
        self._chk(self._L.tdoa_window_quality_u8(self._h, _u8(s), s.size // 2, out.ctypes.data_as(C.c_void_p)))
        return out[0]

    def process_fine(self, gate_samples, rank=0, world=1):
        """tdoa_process_fine -> (peaks [W][P], fine [W][P])"""
        wpb, w = self.num_windows()
        p = self.num_pairs()
        out = np.zeros((w, p), dtype=PEAK_DTYPE)
        fine = np.zeros((w, p), dtype=FINE_DTYPE)
        self._chk(self._L.tdoa_process_fine(self._h, int(rank), int(world), float(gate_samples),
                                            out.ctypes.data_as(C.c_void_p), fine.ctypes.data_as(C.c_void_p)))
        return out, fine

    def fm_xcorr_fine(self, iq1, iq2, max_lag, gate_samples):
        a = np.ascontiguousarray(iq1, dtype=np.uint8)
        b = np.ascontiguousarray(iq2, dtype=np.uint8)
        pk, fk = Peak(), FinePeak()
        self._chk(self._L.tdoa_fm_xcorr_fine_u8(self._h, _u8(a), a.size // 2, _u8(b), b.size // 2, int(max_lag),
                                                float(gate_samples), C.byref(pk), C.byref(fk)))
        return (pk.lag, pk.corr), dict(delay=fk.delay, frac=fk.frac, y=np.array(list(fk.y)),
                                       plausible=bool(fk.plausible))

    def process_u8(self, captures):
        caps = [np.ascontiguousarray(c, dtype=np.uint8) for c in captures]
        n = len(caps)
        ptrs = (C.POINTER(C.c_uint8) * n)(*[_u8(c) for c in caps])
        sizes = (C.c_size_t * n)(*[c.size // 2 for c in caps])
        blk = min(c.size // 2 for c in caps) // 3
        wl = min(self.params.window_len, blk)
        w = 3 * max(1, blk // wl) if blk >= 2 else 0
        out = np.zeros((w, n * (n - 1) // 2), dtype=PEAK_DTYPE)
        self._chk(self._L.tdoa_process_u8(self._h, ptrs, sizes, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def fm_xcorr(self, iq1, iq2, max_lag):
        a = np.ascontiguousarray(iq1, dtype=np.uint8)
        b = np.ascontiguousarray(iq2, dtype=np.uint8)
        pk = Peak()
        self._chk(self._L.tdoa_fm_xcorr_u8(self._h, _u8(a), a.size // 2, _u8(b), b.size // 2, int(max_lag), C.byref(pk)))
        return pk.lag, pk.corr

    def fm_xcorr_lags(self, iq1, iq2, max_lag):
        a = np.ascontiguousarray(iq1, dtype=np.uint8)
        b = np.ascontiguousarray(iq2, dtype=np.uint8)
        out = np.zeros(2 * max_lag - 1, dtype=np.float64)
        self._chk(self._L.tdoa_fm_xcorr_lags_u8(self._h, _u8(a), a.size // 2, _u8(b), b.size // 2, int(max_lag), _d(out)))
        return out

    def fm_preprocess(self, iq):
        a = np.ascontiguousarray(iq, dtype=np.uint8)
        out = np.empty(a.size // 2, dtype=np.float32)
        st = FmStats()
        self._chk(self._L.tdoa_fm_preprocess_u8(self._h, _u8(a), a.size // 2, _f(out), C.byref(st)))
        return out, st

    def graph_info(self, dot_path=None):
        """{nodes, edges, roots, memsets} of the step graph the last process() captured (tdoa_debug_graph_info)"""
        info = (C.c_int32 * 4)()
        self._chk(self._L.tdoa_debug_graph_info(self._h, info, dot_path.encode() if dot_path else None))
        return dict(nodes=info[0], edges=info[1], roots=info[2], memsets=info[3])

    def fm_stats(self, iq):
        """window statistics from the reduce-only pass of the fused path (tdoa_fm_preprocess_u8 with out_f32 = NULL)"""
        a = np.ascontiguousarray(iq, dtype=np.uint8)
        st = FmStats()
        self._chk(self._L.tdoa_fm_preprocess_u8(self._h, _u8(a), a.size // 2, None, C.byref(st)))
        return st

    def force_generic(self, on=True):
        self._chk(self._L.tdoa_debug_force_generic(self._h, 1 if on else 0))

    def debug_flags(self, generic=False, no_short_lag=False, no_fused_k1=False, no_segment_form=False, no_xcd_rows=False,
                    no_segment_quads=False, no_decimate=False, no_k1_once=False, no_seg_pack3=False, no_dec_cols=False, dec_cols_always=False, pow2_only=False, no_dec_staged=False, no_small_fused=False, small_fused_always=False):
        """pick kernel variants by hand (tests / measurements): include/tdoa_mi355x.h TDOA_DEBUG_*; no argument = the
        library's default path, every argument switches one specialised form off"""
        self._chk(self._L.tdoa_debug_flags(self._h, (1 if generic else 0) | (2 if no_short_lag else 0) |
                                           (4 if no_fused_k1 else 0) | (8 if no_segment_form else 0) |
                                           (16 if no_xcd_rows else 0) | (64 if no_segment_quads else 0) |
                                           (256 if no_decimate else 0) | (512 if no_k1_once else 0) |
                                           (1024 if no_seg_pack3 else 0) | (2048 if no_dec_cols else 0) |
                                           (4096 if dec_cols_always else 0) | (8192 if pow2_only else 0) | (16384 if no_dec_staged else 0) | (32768 if no_small_fused else 0) | (65536 if small_fused_always else 0)))

    def last_k1(self, sw_index=0):
        """(statistics of station-window `sw_index` of the last batch, True if that batch read every capture byte once:
        the single-look K1 of csrc/k1_single_look.hpp) -- tdoa_debug_last_k1"""
        st = FmStats()
        once = C.c_int32(0)
        self._chk(self._L.tdoa_debug_last_k1(self._h, int(sw_index), C.byref(st), C.byref(once)))
        return st, bool(once.value)

    # ---- measurement -------------------------------------------------------
    def profile_enable(self, on=True):
        """True / 1: launch by launch with events; 2: events as nodes of the replayed step graph; False / 0: off"""
        self._chk(self._L.tdoa_profile_enable(self._h, int(on)))

    def profile_select(self, names=None):
        """record events only for the named scopes (tdoa_kernel_name); None = all"""
        if names is None:
            mask = 0xffffffff
        else:
            all_names = [self._L.tdoa_kernel_name(k).decode() for k in range(len(KERNELS))]
            mask = 0
            for n in names:
                mask |= 1 << all_names.index(n)
        self._chk(self._L.tdoa_profile_select(self._h, mask))

    def profile_reset(self):
        self._chk(self._L.tdoa_profile_reset(self._h))

    def profile(self):
        out = {}
        for k, name in enumerate(KERNELS):
            ms, n, b = C.c_double(), C.c_int64(), C.c_double()
            self._chk(self._L.tdoa_profile_get(self._h, k, C.byref(ms), C.byref(n), C.byref(b)))
            out[self._L.tdoa_kernel_name(k).decode()] = {"ms": ms.value, "launches": n.value, "bytes": b.value}
        return out

    def plan_info(self):
        n, n1, n2 = C.c_int64(), C.c_int32(), C.c_int32()
        self._chk(self._L.tdoa_plan_info(self._h, C.byref(n), C.byref(n1), C.byref(n2)))
        return n.value, n1.value, n2.value


def latlon_to_ecef(lat, lon, elev):
    out = np.zeros(3)
    load().tdoa_latlon_to_ecef(lat, lon, elev, _d(out))
    return out


def ecef_to_latlon(x, y, z):
    out = np.zeros(3)
    load().tdoa_ecef_to_latlon(x, y, z, _d(out))
    return out


def solve_3station(stations_lle, range_diff):
    st = np.ascontiguousarray(stations_lle, dtype=np.float64).reshape(9)
    rd = np.ascontiguousarray(range_diff, dtype=np.float64)
    out = np.zeros(3)
    it = C.c_int()
    rc = load().tdoa_solve_3station(_d(st), _d(rd), _d(out), C.byref(it))
    return rc, out, it.value


def segment_quads(n_stations, pairs):
    """host only: the quad cover the segment form uses for a window's (template, signal) station pairs;
    returns rows [a, b, c, d, pair(a,c), pair(a,d), pair(b,c), pair(b,d)] (-1 = empty / not wanted)"""
    pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    out = np.full((max(len(pr), 1), 8), -2, dtype=np.int32)
    n = load().tdoa_debug_segment_quads(int(n_stations), pr.ctypes.data_as(C.POINTER(C.c_int32)), len(pr),
                                        out.ctypes.data_as(C.POINTER(C.c_int32)), len(out))
    if n < 0:
        raise ValueError("tdoa_debug_segment_quads: error %d" % -n)
    return out[:n]


def staged_groups(n_stations, max_pairs=15):
    """host only: the staged column walk's share-out of a window's pairs to workgroups: a list of (station mask, [pair numbers])"""
    masks = np.zeros(128, dtype=np.uint32)
    counts = np.zeros(128, dtype=np.int32)
    pairs = np.zeros((128, 16), dtype=np.uint8)
    n = load().tdoa_debug_staged_groups(int(n_stations), int(max_pairs), masks.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        counts.ctypes.data_as(C.POINTER(C.c_int32)), pairs.ctypes.data_as(C.POINTER(C.c_uint8)), 128)
    if n < 0:
        raise ValueError("tdoa_debug_staged_groups: error %d" % -n)
    return [(int(masks[g]), [int(x) for x in pairs[g, :counts[g]]]) for g in range(n)]


def solve_nstation(stations_lle, range_diff, weights=None, solve_z=False):
    st = np.ascontiguousarray(stations_lle, dtype=np.float64)
    n = st.size // 3
    rd = np.ascontiguousarray(range_diff, dtype=np.float64)
    wt = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    out = np.zeros(3)
    it = C.c_int()
    rc = load().tdoa_solve_nstation(_d(st.reshape(-1)), n, _d(rd), _d(wt) if wt is not None else None,
                                    1 if solve_z else 0, _d(out), C.byref(it))
    return rc, out, it.value


def solve_surface(stations_lle, range_diff, weights=None, height_m=0.0):
    """tdoa_solve_surface: least-squares fix on the ellipsoid at `height_m` -> (status, lle, iterations)"""
    st = np.ascontiguousarray(stations_lle, dtype=np.float64)
    n = st.size // 3
    rd = np.ascontiguousarray(range_diff, dtype=np.float64)
    wt = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    out = np.zeros(3)
    it = C.c_int()
    rc = load().tdoa_solve_surface(_d(st.reshape(-1)), n, _d(rd), _d(wt) if wt is not None else None, float(height_m),
                                   _d(out), C.byref(it))
    return rc, out, it.value
