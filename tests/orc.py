"""ctypes binding of the CPU oracle (oracle/libsqg_oracle.so).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libsqg_oracle.so")


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std",
        "median_before_mean", "median_before_std", "dwell_mean", "dwell_std")]


class Kmer(C.Structure):
    _fields_ = [("level_mean", C.c_float), ("level_stdv", C.c_float)]


class Norm(C.Structure):
    _fields_ = [("m", C.c_double), ("s", C.c_double), ("x", C.c_int64)]


class Gamma(C.Structure):
    _fields_ = [("a", C.c_double), ("b", C.c_double), ("x", C.c_int64)]


class Worker(C.Structure):
    _fields_ = [("pos_x", C.c_int64), ("strand_x", C.c_int64), ("meth_x", C.c_int64),
                ("dwell", Norm), ("rlen", Gamma), ("offset", Norm), ("median", Norm),
                ("kmer_x", C.POINTER(C.c_int64))]


class Ref(C.Structure):
    _fields_ = [("num_ref", C.c_int32), ("sum", C.c_int64), ("names", C.POINTER(C.c_char_p)),
                ("seqs", C.POINTER(C.c_char_p)), ("lengths", C.POINTER(C.c_int32)),
                ("trans_n", C.c_int32), ("trans_csum", C.POINTER(C.c_float)),
                ("trans_idx", C.POINTER(C.c_int32)), ("meth", C.POINTER(C.POINTER(C.c_uint8)))]


class Core(C.Structure):
    _fields_ = [("prof", Profile), ("flags", C.c_uint32), ("amp_noise", C.c_float),
                ("kmer_size", C.c_uint32), ("num_kmer", C.c_uint32), ("model", C.POINTER(Kmer)),
                ("kmer_s", C.POINTER(C.c_double)), ("seed", C.c_int64), ("num_workers", C.c_int32),
                ("rlen", C.c_int32), ("workers", C.POINTER(Worker)), ("n_samples", C.c_int64),
                ("total_reads", C.c_int64)]


class Read(C.Structure):
    _fields_ = [("tid", C.c_int32), ("ref_idx", C.c_int32), ("ref_len", C.c_int32),
                ("ref_pos_st", C.c_int32), ("ref_pos_end", C.c_int32), ("rlen", C.c_int32),
                ("strand", C.c_char), ("seq", C.POINTER(C.c_char)), ("offset", C.c_double),
                ("median_before", C.c_double), ("len_raw_signal", C.c_int64),
                ("raw_signal", C.POINTER(C.c_int16)), ("start_time", C.c_int64),
                ("read_number", C.c_int64), ("ss_n", C.c_int64), ("ss", C.POINTER(C.c_int32))]


class Batch(C.Structure):
    _fields_ = [("n", C.c_int32), ("reads", C.POINTER(Read))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsqg_oracle.so"])
        L = C.CDLL(_SO)
        L.orc_rng.restype = C.c_double
        L.orc_rng.argtypes = [C.POINTER(C.c_int64)]
        L.orc_nrng.restype = C.c_double
        L.orc_nrng.argtypes = [C.POINTER(Norm)]
        L.orc_grng.restype = C.c_double
        L.orc_grng.argtypes = [C.POINTER(Gamma)]
        L.orc_kmer_rank.restype = C.c_uint32
        L.orc_kmer_rank.argtypes = [C.c_char_p, C.c_uint32]
        L.orc_read_model.restype = C.c_uint32
        L.orc_read_model.argtypes = [C.c_char_p, C.POINTER(C.POINTER(Kmer))]
        L.orc_ref_load.restype = C.POINTER(Ref)
        L.orc_ref_load.argtypes = [C.c_char_p]
        L.orc_ref_load_trans_count.restype = C.c_int
        L.orc_ref_load_trans_count.argtypes = [C.POINTER(Ref), C.c_char_p]
        L.orc_ref_load_meth_freq.restype = C.c_int
        L.orc_ref_load_meth_freq.argtypes = [C.POINTER(Ref), C.c_char_p]
        L.orc_meth_kmer_rank.restype = C.c_uint32
        L.orc_meth_kmer_rank.argtypes = [C.c_char_p, C.c_uint32]
        L.orc_ref_free.argtypes = [C.POINTER(Ref)]
        L.orc_core_new.restype = C.POINTER(Core)
        L.orc_core_new.argtypes = [C.POINTER(Profile), C.c_uint32, C.c_float, C.c_uint32,
                                   C.POINTER(Kmer), C.c_int64, C.c_int32, C.c_int32]
        L.orc_core_free.argtypes = [C.POINTER(Core)]
        L.orc_batch_run.restype = C.POINTER(Batch)
        L.orc_batch_run.argtypes = [C.POINTER(Core), C.POINTER(Ref), C.c_int32, C.c_int, C.c_int]
        L.orc_batch_run_seqs.restype = C.POINTER(Batch)
        L.orc_batch_run_seqs.argtypes = [C.POINTER(Core), C.c_int32, C.POINTER(C.c_char_p),
                                         C.POINTER(C.c_int32), C.c_int, C.c_int]
        L.orc_batch_run_assigned.restype = C.POINTER(Batch)
        L.orc_batch_run_assigned.argtypes = [C.POINTER(Core), C.c_int32, C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
        L.orc_batch_free.argtypes = [C.POINTER(Batch)]
        L.orc_worker_of.restype = C.c_int32
        L.orc_worker_of.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.orc_svb_zd_bound.restype = C.c_size_t
        L.orc_svb_zd_bound.argtypes = [C.c_int64]
        L.orc_svb_zd.restype = C.c_size_t
        L.orc_svb_zd.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        _lib = L
    return _lib


def make_profile(p) -> Profile:
    return Profile(*p.as_tuple())


def make_model(mean, stdv):
    n = len(mean)
    arr = (Kmer * n)()
    a = np.frombuffer(arr, dtype=np.float32).reshape(n, 2)
    a[:, 0] = mean
    a[:, 1] = stdv
    return arr


class OracleRead:
    __slots__ = ("tid", "ref_idx", "ref_len", "ref_pos_st", "ref_pos_end", "rlen", "strand", "seq",
                 "offset", "median_before", "sig", "start_time", "read_number", "ss")


class Oracle:
    """One simulation context = the reference's core_t for (profile, flags, model, seed, -t)."""

    def __init__(self, profile, flags, k, mean, stdv, seed, num_workers=1, rlen=10000, amp_noise=1.0):
        self.L = lib()
        self._prof = make_profile(profile)
        self._model = make_model(mean, stdv)
        self.k = k
        self.core = self.L.orc_core_new(C.byref(self._prof), flags, amp_noise, k, self._model,
                                        seed, num_workers, rlen)
        self.ref = None

    def load_ref(self, fasta, trans_count=None, meth_freq=None):
        self.ref = self.L.orc_ref_load(os.fsencode(fasta))
        if not self.ref:
            raise FileNotFoundError(fasta)
        if trans_count:
            rc = self.L.orc_ref_load_trans_count(self.ref, os.fsencode(trans_count))
            assert rc == 0, rc
        if meth_freq:
            rc = self.L.orc_ref_load_meth_freq(self.ref, os.fsencode(meth_freq))
            assert rc == 0, rc
        return self.ref.contents

    def ref_name(self, i):
        return self.ref.contents.names[i].decode()

    def _collect(self, b, want_ss):
        out = []
        for i in range(b.contents.n):
            r = b.contents.reads[i]
            o = OracleRead()
            o.tid, o.ref_idx, o.ref_len = r.tid, r.ref_idx, r.ref_len
            o.ref_pos_st, o.ref_pos_end, o.rlen = r.ref_pos_st, r.ref_pos_end, r.rlen
            o.strand = r.strand.decode()
            o.seq = C.string_at(r.seq, r.rlen)
            o.offset, o.median_before = r.offset, r.median_before
            n = r.len_raw_signal
            o.sig = np.ctypeslib.as_array(r.raw_signal, shape=(n,)).copy() if n else np.zeros(0, np.int16)
            o.start_time, o.read_number = r.start_time, r.read_number
            o.ss = (np.ctypeslib.as_array(r.ss, shape=(r.ss_n,)).copy() if want_ss and r.ss_n
                    else np.zeros(0, np.int32))
            out.append(o)
        self.L.orc_batch_free(b)
        return out

    def run_batch(self, n_rec, want_ss=True, nthreads=1):
        b = self.L.orc_batch_run(self.core, self.ref, n_rec, int(want_ss), nthreads)
        return self._collect(b, want_ss)

    def run_batch_seqs(self, seqs, want_ss=True, nthreads=1):
        n = len(seqs)
        arr = (C.c_char_p * n)(*seqs)
        lens = (C.c_int32 * n)(*[len(s) for s in seqs])
        b = self.L.orc_batch_run_seqs(self.core, n, arr, lens, int(want_ss), nthreads)
        return self._collect(b, want_ss)

    def run_batch_assigned(self, seqs, workers, want_ss=True):
        n = len(seqs)
        arr = (C.c_char_p * n)(*seqs)
        lens = (C.c_int32 * n)(*[len(s) for s in seqs])
        wk = (C.c_int32 * n)(*[int(w) for w in workers])
        b = self.L.orc_batch_run_assigned(self.core, n, arr, lens, wk, int(want_ss))
        return self._collect(b, want_ss)

    def simulate(self, n, batch=1000, want_ss=True, nthreads=1):
        """sim_main's batch loop, src/sim.c:1065-1075."""
        out, done = [], 0
        while done < n:
            nb = min(batch, n - done)
            out += self.run_batch(nb, want_ss, nthreads)
            done += nb
        return out

    def close(self):
        if self.core:
            self.L.orc_core_free(self.core)
            self.core = None
        if self.ref:
            self.L.orc_ref_free(self.ref)
            self.ref = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def svb_zd(sig):
    """slow5lib's svb-zd encoding of an int16 array (oracle restatement, oracle/sqg_oracle.c)."""
    import numpy as np
    sig = np.ascontiguousarray(sig, np.int16)
    L = lib()
    out = np.zeros(L.orc_svb_zd_bound(len(sig)), np.uint8)
    n = L.orc_svb_zd(sig.ctypes.data, len(sig), out.ctypes.data)
    return out[:n].copy()


def svb_zd_decode(enc):
    """Decoder for the round-trip tests (format: uint32 count | ceil(count/4) key bytes | data)."""
    import numpy as np
    enc = np.ascontiguousarray(enc, np.uint8)
    count = int(enc[:4].view(np.uint32)[0])
    nkey = (count + 3) // 4
    keys = enc[4:4 + nkey]
    codes = ((keys[:, None] >> (2 * np.arange(4, dtype=np.uint8))) & 3).reshape(-1)[:count].astype(np.int64)
    lens = codes + 1
    off = np.concatenate(([0], np.cumsum(lens)))[:-1] + 4 + nkey
    data = np.concatenate((enc, np.zeros(4, np.uint8))).astype(np.uint32)
    z = np.zeros(count, np.uint32)
    for b in range(4):
        z |= np.where(lens > b, data[off + b] << np.uint32(8 * b), np.uint32(0)).astype(np.uint32)
    d = (z >> np.uint32(1)).astype(np.int64) ^ -(z & np.uint32(1)).astype(np.int64)
    return np.cumsum(d).astype(np.int16), int(off[-1] + lens[-1]) if count else 4
