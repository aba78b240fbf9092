"""ctypes binding of the C ABI in include/sqg.h (squigulator_amd/csrc/libsqg_hip.so).

This is plumbing for tests and bench.py; the product is the shared library.  There is NO CPU
fallback: if the HIP library is missing or no GPU is usable, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from . import profiles as P

ABI_VERSION = 2
MODE_EXACT = 0
MODE_CERTIFIED = 1

_ERRORS = {-1: "SQG_EINVAL", -2: "SQG_ENOMEM", -3: "SQG_EDEVICE", -4: "SQG_ESEQUENCE",
           -5: "SQG_ENODEVICE", -6: "SQG_EOVERFLOW", -7: "SQG_EIO"}


class SqgError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {_ERRORS.get(code, code)} {detail}".strip())


class CProfile(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std",
        "median_before_mean", "median_before_std", "dwell_mean", "dwell_std")]


class CKmer(C.Structure):
    _fields_ = [("level_mean", C.c_float), ("level_stdv", C.c_float)]


class CCfg(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("profile", CProfile), ("flags", C.c_uint32),
                ("amp_noise", C.c_float), ("kmer_size", C.c_uint32), ("model", C.POINTER(CKmer)),
                ("seed", C.c_int64), ("num_workers", C.c_int32), ("worker_lo", C.c_int32),
                ("worker_hi", C.c_int32), ("device", C.c_int32), ("mode", C.c_uint32)]


class CResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_events", C.c_int64), ("n_samples", C.c_int64),
                ("n_bases", C.c_int64), ("sig_off", C.POINTER(C.c_int64)), ("ev_off", C.POINTER(C.c_int64)),
                ("offset", C.POINTER(C.c_double)), ("median_before", C.POINTER(C.c_double)),
                ("d_signal", C.c_void_p), ("d_dwell", C.c_void_p)]


class CTiming(C.Structure):
    _fields_ = [("dwell_ms", C.c_float), ("events_ms", C.c_float), ("samples_ms", C.c_float),
                ("lean_ms", C.c_float), ("total_ms", C.c_float), ("fallback_samples", C.c_int64),
                ("carried_first_pass", C.c_int32), ("first_pass_ran_ahead", C.c_int32)]


class CSvb(C.Structure):
    _fields_ = [("n_bytes", C.c_int64), ("svb_off", C.POINTER(C.c_int64)), ("d_svb", C.c_void_p)]


class CGenome(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("seqs", C.c_void_p), ("contig_off", C.POINTER(C.c_int64)),
                ("rlen", C.c_int32), ("mode", C.c_uint32), ("n_trans", C.c_int32),
                ("trans_csum", C.POINTER(C.c_float)), ("trans_idx", C.POINTER(C.c_int32))]


class CSample(C.Structure):
    _fields_ = [("ref_idx", C.POINTER(C.c_int32)), ("ref_len", C.POINTER(C.c_int32)), ("ref_pos", C.POINTER(C.c_int32)),
                ("rlen", C.POINTER(C.c_int32)), ("strand", C.POINTER(C.c_char)), ("seq_off", C.POINTER(C.c_int64))]


SAMPLE_DNA, SAMPLE_RNA, SAMPLE_CDNA, SAMPLE_TRUNC, SAMPLE_FULL = 0, 1, 2, 4, 8
BLOW5_STORED = 0x10000          # include/sqg.h SQG_BLOW5_STORED (flags of sqg_blow5_open)

EXPORTS = ("sqg_create", "sqg_destroy", "sqg_last_error", "sqg_strerror", "sqg_device_count",
           "sqg_batch_stage", "sqg_batch_run", "sqg_batch_wait", "sqg_fetch_signal", "sqg_fetch_dwell",
           "sqg_batch_free", "sqg_get_timing", "sqg_set_phase_timing", "sqg_submit", "sqg_worker_of", "sqg_probe_store_bandwidth", "sqg_probe_lds_order",
           "sqg_batch_compress", "sqg_fetch_svb", "sqg_genome_load", "sqg_batch_sample", "sqg_fetch_reads",
           "sqg_host_alloc", "sqg_host_free", "sqg_set_range_mode", "sqg_skip_reads", "sqg_batch_sample_range",
           "sqg_batch_run_begin", "sqg_batch_run_end", "sqg_genome_load_device",
           "sqg_blow5_open", "sqg_blow5_write", "sqg_blow5_write_batch", "sqg_blow5_close", "sqg_blow5_last_error", "sqg_batch_blow5_records",
           "sqg_genome_set_meth", "sqg_build_info", "sqg_set_stage_threads")

# Environment knobs only the development build of the library reads (csrc/h_common.h: SQG_DEV_ENV; tools/README.md).  The release
# library ignores them, so a process that sets one -- a test forcing a code path, an A/B script -- gets libsqg_hip_dev.so.
DEV_KNOBS = ("SQG_SEPARATE_DWELL", "SQG_EVENTS_WIDE_MAX", "SQG_MID_SPLIT", "SQG_SCAN_G4", "SQG_TEST_ORDER_FAULT", "SQG_LEAN_GRID",
             "SQG_LEAN_DYNLDS", "SQG_FIX_INLINE", "SQG_ABL_NOFIX", "SQG_SAMPLER_SERIAL", "SQG_OVERLAP", "SQG_PART_CLAIMS",
             "SQG_TEST_DELTA_X", "SQG_LEAN_EPL", "SQG_TEST_ROW_TURNS", "SQG_PART_WG_EVENTS", "SQG_SPLIT_CHAINS", "SQG_NO_PART",
             "SQG_PART_SLICE", "SQG_TEST_NO_LEAN", "SQG_STAGE_THREADS", "SQG_NO_PRECOUNT", "SQG_PHC_ABL", "SQG_PHC_GRID", "SQG_NO_DRAW_AHEAD",
             "SQG_CU_SPLIT", "SQG_NO_FOLD", "SQG_NO_WHOLE_LINKS", "SQG_NO_PLACE")

_libs = {}                  # absolute path -> loaded library
LOADED_PATH = None          # the library the last load_library() call opened (bench.py prints it with its hash)


def dev_knobs_set() -> list:
    return [k for k in DEV_KNOBS if k in os.environ]


def default_library_path() -> str:
    """SQG_LIB (A/B testing of kernel build variants), else the development build if the environment holds one of its knobs,
    else the release library"""
    return os.environ.get("SQG_LIB") or (_build.LIB_DEV if dev_knobs_set() else _build.LIB)


def build_info(L) -> dict:
    """sqg_build_info() as a dict: {"source_hash": ..., "dev": "0"|"1"}"""
    return dict(kv.split("=", 1) for kv in L.sqg_build_info().decode().split(";") if "=" in kv)


def load_library(path: str | None = None):
    """dlopen the HIP library; raises if it has not been built (no fallback)."""
    path = os.path.abspath(path or default_library_path())
    global LOADED_PATH
    if path in _libs:
        LOADED_PATH = path
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m squigulator_amd.build` "
                           "(there is no CPU fallback for the signal path)")
    L = C.CDLL(path)
    LOADED_PATH = path
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.sqg_create.restype = C.c_int
    L.sqg_create.argtypes = [C.POINTER(CCfg), C.POINTER(vp)]
    L.sqg_destroy.restype = None
    L.sqg_destroy.argtypes = [vp]
    L.sqg_last_error.restype = C.c_char_p
    L.sqg_last_error.argtypes = [vp]
    L.sqg_strerror.restype = C.c_char_p
    L.sqg_strerror.argtypes = [C.c_int]
    L.sqg_device_count.restype = C.c_int
    L.sqg_batch_stage.restype = C.c_int
    L.sqg_batch_stage.argtypes = [vp, i32, C.c_char_p, C.POINTER(i64), C.POINTER(i32), C.POINTER(vp)]
    L.sqg_batch_run.restype = C.c_int
    L.sqg_batch_run.argtypes = [vp, vp]
    L.sqg_batch_wait.restype = C.c_int
    L.sqg_batch_wait.argtypes = [vp, vp, C.POINTER(CResult)]
    L.sqg_fetch_signal.restype = C.c_int
    L.sqg_fetch_signal.argtypes = [vp, vp, C.c_void_p]
    L.sqg_fetch_dwell.restype = C.c_int
    L.sqg_fetch_dwell.argtypes = [vp, vp, C.c_void_p]
    L.sqg_batch_free.restype = None
    L.sqg_batch_free.argtypes = [vp, vp]
    L.sqg_get_timing.restype = C.c_int
    L.sqg_get_timing.argtypes = [vp, C.POINTER(CTiming)]
    L.sqg_set_phase_timing.restype = C.c_int
    L.sqg_set_phase_timing.argtypes = [vp, C.c_int]
    L.sqg_submit.restype = C.c_int
    L.sqg_submit.argtypes = [vp, i32, C.c_char_p, C.POINTER(i64), C.POINTER(i32), C.POINTER(vp), C.POINTER(CResult)]
    L.sqg_worker_of.restype = i32
    L.sqg_worker_of.argtypes = [i32, i32, i32]
    L.sqg_probe_store_bandwidth.restype = C.c_int
    L.sqg_probe_store_bandwidth.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
    L.sqg_probe_lds_order.restype = C.c_int
    L.sqg_probe_lds_order.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    L.sqg_batch_compress.restype = C.c_int
    L.sqg_batch_compress.argtypes = [vp, vp, C.POINTER(CSvb)]
    L.sqg_fetch_svb.restype = C.c_int
    L.sqg_fetch_svb.argtypes = [vp, vp, vp]
    L.sqg_genome_load.restype = C.c_int
    L.sqg_genome_load.argtypes = [vp, C.POINTER(CGenome)]
    L.sqg_genome_load_device.restype = C.c_int
    L.sqg_genome_load_device.argtypes = [vp, C.POINTER(CGenome)]
    L.sqg_genome_set_meth.restype = C.c_int
    L.sqg_genome_set_meth.argtypes = [vp, vp, vp]
    L.sqg_batch_sample.restype = C.c_int
    L.sqg_batch_sample.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(vp), C.POINTER(CSample)]
    L.sqg_fetch_reads.restype = C.c_int
    L.sqg_fetch_reads.argtypes = [vp, vp, vp]
    L.sqg_host_alloc.restype = vp
    L.sqg_host_alloc.argtypes = [C.c_size_t]
    L.sqg_host_free.restype = None
    L.sqg_host_free.argtypes = [vp]
    L.sqg_set_range_mode.restype = C.c_int
    L.sqg_set_range_mode.argtypes = [vp, C.c_int]
    L.sqg_skip_reads.restype = C.c_int
    L.sqg_skip_reads.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i32)]
    L.sqg_batch_sample_range.restype = C.c_int
    L.sqg_batch_sample_range.argtypes = [vp, i32, C.POINTER(i32), i32, i32, C.POINTER(vp), C.POINTER(CSample)]
    L.sqg_batch_run_begin.restype = C.c_int
    L.sqg_batch_run_begin.argtypes = [vp, vp, C.POINTER(vp)]
    L.sqg_batch_run_end.restype = C.c_int
    L.sqg_batch_run_end.argtypes = [vp, vp, vp, vp]
    L.sqg_blow5_open.restype = C.c_int
    L.sqg_blow5_open.argtypes = [C.c_char_p, C.POINTER(CProfile), C.c_uint32, i32, C.POINTER(vp)]
    L.sqg_blow5_write.restype = C.c_int
    L.sqg_blow5_write.argtypes = [vp, i32, C.c_char_p, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(i64), vp, C.POINTER(i64)]
    L.sqg_blow5_write_batch.restype = C.c_int
    L.sqg_blow5_write_batch.argtypes = [vp, vp, vp, C.c_char_p, C.POINTER(i64)]
    L.sqg_batch_blow5_records.restype = C.c_int
    L.sqg_batch_blow5_records.argtypes = [vp, vp, C.POINTER(CProfile), C.c_uint32, C.c_char_p, C.POINTER(i64), i64, C.c_uint64,
                                          C.POINTER(C.c_void_p), C.POINTER(i64), C.POINTER(C.POINTER(i64))]
    L.sqg_blow5_close.restype = C.c_int
    L.sqg_blow5_close.argtypes = [vp, C.POINTER(i64)]
    L.sqg_blow5_last_error.restype = C.c_char_p
    L.sqg_blow5_last_error.argtypes = [vp]
    L.sqg_set_stage_threads.restype = C.c_int
    L.sqg_set_stage_threads.argtypes = [vp, C.c_int]
    L.sqg_build_info.restype = C.c_char_p
    L.sqg_build_info.argtypes = []
    _libs[path] = L
    return L


def _atoi(t: str) -> int:
    """C atoi: optional blanks and sign, then the leading digits (0 when there are none)"""
    import re
    m = re.match(r"\s*([+-]?\d+)", t)
    return int(m.group(1)) if m else 0


def _atof(t: str) -> float:
    """C atof: the longest leading decimal floating-point prefix (0.0 when there is none)"""
    import re
    m = re.match(r"\s*([+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|inf(inity)?|nan))", t, re.I)
    return float(m.group(1)) if m else 0.0


def load_meth_freq(contigs, names, meth_freq_path: str):
    """load_meth_freq, src/ref.c:291-361: the --meth-freq table (tab separated: contig, 0-based position of a C, frequency in [0, 1])
    as one frequency byte per base of the concatenated contigs + a flag per contig; the reference's checks, as ValueError"""
    idx = {n: i for i, n in enumerate(names)}
    arrs = [np.zeros(len(c), np.uint8) for c in contigs]
    has = np.zeros(len(contigs), np.uint8)
    with open(meth_freq_path) as f:
        for line, ln in enumerate(f):
            if ln.startswith("#"):
                continue
            # the checks of load_meth_freq, src/ref.c:314-345 (the reference exits; here: ValueError naming the line)
            cols = ln.rstrip("\n").split("\t")
            if len(cols) < 3 or not all(cols[:3]):
                raise ValueError(f"{meth_freq_path}: malformed line {line} (need contig, position, frequency; src/ref.c:279-289)")
            name, pos, fr = cols[:3]
            if name not in idx:
                raise ValueError(f"There was no such chromosome in the reference. Check line {line} value {name} of the input methy-freq file.")
            i, p = idx[name], _atoi(pos)
            if p < 0:
                raise ValueError(f"Chromosome position cannot be negative. Check line {line} value {p} of the input methy-freq file.")
            if p >= len(contigs[i]):
                raise ValueError(f"Chromosome {name} position must be less than the length {len(contigs[i])}. Check line {line} value {p} "
                                 "of the input methy-freq file.")
            if contigs[i][p:p + 1] not in (b"C", b"c"):
                raise ValueError(f"The chromosome {name} position {p} in the reference was a {contigs[i][p:p + 1].decode()}. How can it be "
                                 f"methylated C? Check line {line} of the input methy-freq file.")
            fq = float(np.float32(_atof(fr)))
            if fq < 0 or fq > 1:
                raise ValueError(f"Methylation frequency must be between 0 to 1. Check line {line} value {fq:f} of the input methy-freq file.")
            has[i] = 1
            arrs[i][p] = int(np.floor(float(np.float32(fq) * np.float32(255)) + 0.5))     # (uint8_t)roundf(freq*255), float freq
    blob = np.concatenate(arrs) if arrs else np.zeros(1, np.uint8)
    return blob, has


class Blow5Writer:
    """The library's native BLOW5 writer (sqg_blow5_*): header, record framing and zlib on host threads; the signal field is
    the svb-zd encoding made on the device.  Pure host code: write() works without a GPU."""

    def __init__(self, path: str, profile: P.Profile, flags: int, threads: int = 0, lib_path: str | None = None, stored: bool = False, shards: int = 1):
        """stored: SQG_BLOW5_STORED -- the records in zlib streams of stored blocks, framed on the device by write_batch (include/sqg.h);
        shards > 1: SQG_BLOW5_SHARDS -- that many files (self.paths), each with a contiguous range of every batch's reads"""
        self.L = load_library(lib_path)
        self.h = C.c_void_p()
        cp = CProfile(*profile.as_tuple())
        self.paths = [path] if shards <= 1 else [(path[:-6] + f".{i}.blow5") if path.endswith(".blow5") else f"{path}.{i}" for i in range(shards)]
        rc = self.L.sqg_blow5_open(os.fsencode(path), C.byref(cp), (flags & (P.SQ_RNA | P.SQ_R10 | P.SQ_ONT)) | (BLOW5_STORED if stored else 0)
                                   | ((shards & 0xff) << 24 if shards > 1 else 0), threads, C.byref(self.h))
        if rc != 0:
            raise SqgError(rc, "sqg_blow5_open", path)

    def _chk(self, rc, where):
        if rc != 0:
            raise SqgError(rc, where, self.L.sqg_blow5_last_error(self.h).decode())

    @staticmethod
    def _ids(read_ids):
        blob = b"".join(read_ids)
        off = np.zeros(len(read_ids) + 1, np.int64)
        off[1:] = np.cumsum([len(r) for r in read_ids])
        return blob, off

    def write(self, read_ids, offset, median_before, sig_off, svb, svb_off):
        """read_ids: list of bytes; svb: uint8 array of the concatenated svb-zd encodings, svb_off their offsets"""
        blob, ioff = self._ids(read_ids)
        offset = np.ascontiguousarray(offset, np.float64); median_before = np.ascontiguousarray(median_before, np.float64)
        sig_off = np.ascontiguousarray(sig_off, np.int64); svb_off = np.ascontiguousarray(svb_off, np.int64)
        svb = np.ascontiguousarray(svb, np.uint8)
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int64)
        self._chk(self.L.sqg_blow5_write(self.h, len(read_ids), blob, ioff.ctypes.data_as(ip), offset.ctypes.data_as(dp),
                                         median_before.ctypes.data_as(dp), sig_off.ctypes.data_as(ip), svb.ctypes.data,
                                         svb_off.ctypes.data_as(ip)), "sqg_blow5_write")

    def write_batch(self, batch: "Batch", read_ids):
        blob, ioff = self._ids(read_ids)
        self._chk(self.L.sqg_blow5_write_batch(self.h, batch.gen.ctx, batch.handle, blob, ioff.ctypes.data_as(C.POINTER(C.c_int64))),
                  "sqg_blow5_write_batch")

    def close(self) -> int:
        n = C.c_int64()
        if self.h:
            rc = self.L.sqg_blow5_close(self.h, C.byref(n))
            self.h = None
            if rc != 0:
                raise SqgError(rc, "sqg_blow5_close")
        return n.value


class Batch:
    """A staged batch (one process_db() worth of reads)."""

    def __init__(self, gen, handle, n_reads):
        self.gen, self.handle, self.n_reads = gen, handle, n_reads
        self.res = None

    def run(self):
        self.gen._chk(self.gen.L.sqg_batch_run(self.gen.ctx, self.handle), "sqg_batch_run")
        return self

    def run_begin(self) -> int:
        """Range sharding: first phase of the run; returns the DEVICE address of this range's per-stream sample counts
        (uint32 [num_workers][4^k], valid until the next run_begin)."""
        p = C.c_void_p()
        self.gen._chk(self.gen.L.sqg_batch_run_begin(self.gen.ctx, self.handle, C.byref(p)), "sqg_batch_run_begin")
        return p.value

    def run_end(self, d_before: int | None = None, d_after: int | None = None):
        """second phase; d_before / d_after: device addresses of the counts summed over the earlier / later ranges"""
        self.gen._chk(self.gen.L.sqg_batch_run_end(self.gen.ctx, self.handle, C.c_void_p(d_before), C.c_void_p(d_after)),
                      "sqg_batch_run_end")
        return self

    def wait(self):
        r = CResult()
        self.gen._chk(self.gen.L.sqg_batch_wait(self.gen.ctx, self.handle, C.byref(r)), "sqg_batch_wait")
        self.res = r
        n = r.n_reads
        self.n_samples, self.n_events, self.n_bases = r.n_samples, r.n_events, r.n_bases
        self.sig_off = np.ctypeslib.as_array(r.sig_off, shape=(n + 1,)).copy()
        self.ev_off = np.ctypeslib.as_array(r.ev_off, shape=(n + 1,)).copy()
        self.offset = np.ctypeslib.as_array(r.offset, shape=(n,)).copy() if n else np.zeros(0)
        self.median_before = np.ctypeslib.as_array(r.median_before, shape=(n,)).copy() if n else np.zeros(0)
        return self

    def signal(self, out: np.ndarray | None = None) -> np.ndarray:
        """int16 samples of the batch; `out` may be (a view of) a pinned buffer from SignalGenerator.pinned()."""
        if out is None:
            out = np.empty(self.n_samples, np.int16)
        elif out.dtype != np.int16 or len(out) < self.n_samples:
            raise ValueError("destination too small for the batch's samples")
        self.gen._chk(self.gen.L.sqg_fetch_signal(self.gen.ctx, self.handle, out.ctypes.data), "sqg_fetch_signal")
        return out[:self.n_samples]

    def dwell(self) -> np.ndarray:
        out = np.empty(self.n_events, np.int32)
        self.gen._chk(self.gen.L.sqg_fetch_dwell(self.gen.ctx, self.handle, out.ctypes.data), "sqg_fetch_dwell")
        return out

    def reads(self):
        """The sampled reads as gen_read returned them (list of bytes); only for batches made by sample()."""
        off = self.sampled["seq_off"]
        buf = np.empty(max(int(off[-1]), 1), np.uint8)
        self.gen._chk(self.gen.L.sqg_fetch_reads(self.gen.ctx, self.handle, buf.ctypes.data), "sqg_fetch_reads")
        raw = buf.tobytes()
        return [raw[off[i]:off[i + 1]] for i in range(self.n_reads)]

    def compress(self, fetch=True, out: np.ndarray | None = None):
        """svb-zd encodings of the batch's signals (slow5lib's signal compression), made on the device.
        Returns (bytes as uint8 array, offsets[n_reads+1]); fetch=False leaves the bytes on the device."""
        r = CSvb()
        self.gen._chk(self.gen.L.sqg_batch_compress(self.gen.ctx, self.handle, C.byref(r)), "sqg_batch_compress")
        off = np.ctypeslib.as_array(r.svb_off, shape=(self.n_reads + 1,)).copy()
        self._svb_bytes = int(r.n_bytes)
        if not fetch:
            return None, off
        return self.fetch_svb(out), off

    def fetch_svb(self, out: np.ndarray | None = None):
        """the encodings compress(fetch=False) left on the device (sqg_fetch_svb): a host that queues its next batch between the two
        calls has that batch's kernels running while these bytes cross PCIe"""
        n = getattr(self, "_svb_bytes", None)
        if n is None:
            raise ValueError("fetch_svb needs a compress() of this batch first")
        if out is not None and len(out) < n:
            raise ValueError("destination too small for the batch's encodings")
        out = np.empty(n, np.uint8) if out is None else out[:n]
        self.gen._chk(self.gen.L.sqg_fetch_svb(self.gen.ctx, self.handle, out.ctypes.data), "sqg_fetch_svb")
        return out

    def blow5_records(self, profile, flags: int, read_ids, read_number0: int = 0, start_time0: int = 0):
        """the batch's BLOW5 records in stored-block zlib streams, framed on the device (sqg_batch_blow5_records) -> (bytes copy, offsets)"""
        blob = b"".join(read_ids)
        ioff = np.zeros(len(read_ids) + 1, np.int64)
        ioff[1:] = np.cumsum([len(r) for r in read_ids])
        cp = CProfile(*profile.as_tuple())
        recs, nb, ro = C.c_void_p(), C.c_int64(), C.POINTER(C.c_int64)()
        self.gen._chk(self.gen.L.sqg_batch_blow5_records(self.gen.ctx, self.handle, C.byref(cp), flags, blob, ioff.ctypes.data_as(C.POINTER(C.c_int64)),
                                                       read_number0, start_time0, C.byref(recs), C.byref(nb), C.byref(ro)), "sqg_batch_blow5_records")
        data = C.string_at(recs, nb.value) if nb.value else b""
        return data, np.array([ro[i] for i in range(len(read_ids) + 1)], np.int64)

    def free(self):
        if self.handle:
            self.gen.L.sqg_batch_free(self.gen.ctx, self.handle)
            self.handle = None

    def __del__(self):
        try:
            if self.gen.ctx:
                self.free()
        except Exception:
            pass


class SignalGenerator:
    """One simulation context: the reference's core_t for (profile, flags, model, seed, -t)."""

    def __init__(self, profile: P.Profile, flags: int, kmer_size: int, level_mean, level_stdv, seed: int,
                 num_workers: int = 1, amp_noise: float = 1.0, device: int = 0, mode: int = MODE_EXACT,
                 worker_lo: int = 0, worker_hi: int | None = None, lib_path: str | None = None):
        self.L = load_library(lib_path)
        self.ctx = None
        n = 5 ** kmer_size if (flags & P.SQ_METH) else 1 << (2 * kmer_size)
        if len(level_mean) != n or len(level_stdv) != n:
            raise ValueError("pore model must have 4^k rows (5^k for the methylation tables)")
        self._model = (CKmer * n)()
        a = np.frombuffer(self._model, dtype=np.float32).reshape(n, 2)
        a[:, 0] = level_mean
        a[:, 1] = level_stdv
        cfg = CCfg(ABI_VERSION, CProfile(*profile.as_tuple()), flags & 0x303d, amp_noise, kmer_size,
                   self._model, seed, num_workers, worker_lo,
                   num_workers if worker_hi is None else worker_hi, device, mode)
        h = C.c_void_p()
        rc = self.L.sqg_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise SqgError(rc, "sqg_create", self.L.sqg_strerror(rc).decode())
        self.ctx = h
        self.num_workers, self.kmer_size, self.flags, self.profile = num_workers, kmer_size, flags, profile

    def _chk(self, rc, where):
        if rc != 0:
            raise SqgError(rc, where, self.L.sqg_last_error(self.ctx).decode())

    def stage(self, seqs, workers=None) -> Batch:
        """seqs: list of bytes (reads as gen_read returns them)."""
        n = len(seqs)
        off = np.zeros(n + 1, np.int64)
        if n:
            off[1:] = np.cumsum([len(s) for s in seqs])
        blob = b"".join(seqs)
        wk = None
        if workers is not None:
            wk = np.ascontiguousarray(workers, np.int32)
        h = C.c_void_p()
        rc = self.L.sqg_batch_stage(self.ctx, n, blob, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                    wk.ctypes.data_as(C.POINTER(C.c_int32)) if wk is not None else None,
                                    C.byref(h))
        self._chk(rc, "sqg_batch_stage")
        return Batch(self, h, n)

    def stage_packed(self, blob: bytes, off: np.ndarray, workers=None) -> Batch:
        n = len(off) - 1
        off = np.ascontiguousarray(off, np.int64)
        wk = np.ascontiguousarray(workers, np.int32) if workers is not None else None
        h = C.c_void_p()
        rc = self.L.sqg_batch_stage(self.ctx, n, blob, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                    wk.ctypes.data_as(C.POINTER(C.c_int32)) if wk is not None else None,
                                    C.byref(h))
        self._chk(rc, "sqg_batch_stage")
        return Batch(self, h, n)

    def submit(self, seqs, workers=None) -> Batch:
        return self.stage(seqs, workers).run().wait()

    def load_genome(self, contigs, rlen: int, mode: int = SAMPLE_DNA, trans=None):
        """Keep the reference (list of bytes, as load_ref returned it) on the device for sample()."""
        blob = b"".join(contigs)
        off = np.zeros(len(contigs) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in contigs])
        g = CGenome(len(contigs), C.cast(C.c_char_p(blob), C.c_void_p), off.ctypes.data_as(C.POINTER(C.c_int64)), rlen, mode, 0, None, None)
        if trans is not None:
            csum = np.ascontiguousarray(trans[0], np.float32)
            idx = np.ascontiguousarray(trans[1], np.int32)
            g.n_trans = len(csum)
            g.trans_csum = csum.ctypes.data_as(C.POINTER(C.c_float))
            g.trans_idx = idx.ctypes.data_as(C.POINTER(C.c_int32))
        self._chk(self.L.sqg_genome_load(self.ctx, C.byref(g)), "sqg_genome_load")

    def set_meth(self, contigs, names, meth_freq_path: str):
        """--meth-freq FILE (tab separated: contig, 0-based position of a C, frequency) for the genome loaded with
        load_genome(contigs): the per-base frequency bytes load_meth_freq builds (src/ref.c:291-361)"""
        blob, has = load_meth_freq(contigs, names, meth_freq_path)
        self._chk(self.L.sqg_genome_set_meth(self.ctx, blob.ctypes.data, has.ctypes.data), "sqg_genome_set_meth")

    def load_genome_device(self, d_seqs: int, contig_lens, rlen: int, mode: int = SAMPLE_DNA):
        """The same for a reference that already sits in device memory: d_seqs is the device address of the concatenated
        contigs (e.g. torch_tensor.data_ptr()); it is copied, the caller may free it afterwards."""
        off = np.zeros(len(contig_lens) + 1, np.int64)
        off[1:] = np.cumsum(np.asarray(contig_lens, np.int64))
        g = CGenome(len(contig_lens), C.c_void_p(d_seqs), off.ctypes.data_as(C.POINTER(C.c_int64)), rlen, mode, 0, None, None)
        self._chk(self.L.sqg_genome_load_device(self.ctx, C.byref(g)), "sqg_genome_load_device")

    def set_range_mode(self, on: bool = True):
        """range sharding (include/sqg.h): this context owns all workers and generates a range of each batch's reads"""
        self._chk(self.L.sqg_set_range_mode(self.ctx, 1 if on else 0), "sqg_set_range_mode")

    def skip_reads(self, seq_lens, workers):
        ln = np.ascontiguousarray(seq_lens, np.int64)
        wk = np.ascontiguousarray(workers, np.int32)
        assert ln.shape == wk.shape
        self._chk(self.L.sqg_skip_reads(self.ctx, len(ln), ln.ctypes.data_as(C.POINTER(C.c_int64)),
                                        wk.ctypes.data_as(C.POINTER(C.c_int32))), "sqg_skip_reads")

    def sample(self, n: int, workers=None, lo: int | None = None, hi: int | None = None) -> Batch:
        """gen_read for n reads on the device + staging; the returned batch carries .sampled (per-read arrays).
        lo/hi (range sharding): all n reads are sampled, reads [lo, hi) are staged."""
        wk = np.ascontiguousarray(workers, np.int32) if workers is not None else None
        wkp = wk.ctypes.data_as(C.POINTER(C.c_int32)) if wk is not None else None
        h = C.c_void_p()
        info = CSample()
        if lo is None and hi is None:
            rc = self.L.sqg_batch_sample(self.ctx, n, wkp, C.byref(h), C.byref(info))
        else:
            lo, hi = (0 if lo is None else lo), (n if hi is None else hi)
            rc = self.L.sqg_batch_sample_range(self.ctx, n, wkp, lo, hi, C.byref(h), C.byref(info))
            n = hi - lo
        self._chk(rc, "sqg_batch_sample")
        b = Batch(self, h, n)
        arr = lambda p, shape, dt: (np.ctypeslib.as_array(p, shape=shape).copy() if n else np.zeros(0, dt))  # noqa: E731
        b.sampled = dict(ref_idx=arr(info.ref_idx, (n,), np.int32), ref_len=arr(info.ref_len, (n,), np.int32),
                         ref_pos=arr(info.ref_pos, (n,), np.int32), rlen=arr(info.rlen, (n,), np.int32),
                         strand=bytes(info.strand[:n]) if n else b"",
                         seq_off=np.ctypeslib.as_array(info.seq_off, shape=(n + 1,)).copy())
        return b

    def pinned(self, nbytes: int, dtype=np.uint8) -> np.ndarray:
        """A page-locked host array (sqg_host_alloc) for fast sqg_fetch_* destinations; freed with the generator."""
        p = self.L.sqg_host_alloc(nbytes)
        if not p:
            raise MemoryError("sqg_host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_uint8 * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dtype)

    def set_stage_threads(self, n: int) -> int:
        """Host threads that share a batch's per-read libm draws at staging (0: automatic); returns how many the last staging used."""
        rc = self.L.sqg_set_stage_threads(self.ctx, int(n))
        if rc < 0:
            self._chk(rc, "sqg_set_stage_threads")
        return rc

    def set_phase_timing(self, every: int):
        """Phase events (timing()'s milliseconds) on the batches whose run index is a multiple of `every`; 1: all (default), 0: none."""
        self._chk(self.L.sqg_set_phase_timing(self.ctx, int(every)), "sqg_set_phase_timing")

    def timing(self):
        t = CTiming()
        self._chk(self.L.sqg_get_timing(self.ctx, C.byref(t)), "sqg_get_timing")
        return {"dwell_ms": t.dwell_ms, "events_ms": t.events_ms, "samples_ms": t.samples_ms,
                "lean_ms": t.lean_ms, "total_ms": t.total_ms, "fallback_samples": t.fallback_samples,
                "carried_first_pass": bool(t.carried_first_pass), "first_pass_ran_ahead": bool(t.first_pass_ran_ahead)}

    def probe_store_bandwidth(self, nbytes=1 << 30, iters=10) -> float:
        ms = C.c_float()
        self._chk(self.L.sqg_probe_store_bandwidth(self.ctx, nbytes, iters, C.byref(ms)), "sqg_probe_store_bandwidth")
        return nbytes / (ms.value * 1e-3)

    def probe_lds_order(self, workgroups=1024, rounds=16):
        """(mismatches, in_use): the device's LDS atomics against their serial result; whether this context relies on them"""
        bad, used = C.c_uint(), C.c_int()
        self._chk(self.L.sqg_probe_lds_order(self.ctx, workgroups, rounds, C.byref(bad), C.byref(used)), "sqg_probe_lds_order")
        return bad.value, bool(used.value)

    def close(self):
        if self.ctx:
            for p in getattr(self, "_pinned", []):
                self.L.sqg_host_free(p)
            self._pinned = []
            self.L.sqg_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
