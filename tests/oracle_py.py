"""ctypes binding of the CPU oracle (oracle/libkmc_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libkmc_oracle.so")
ORACLE_CLI = os.path.join(ORACLE_DIR, "kmc_oracle_cli")


class _Reads(C.Structure):
    _fields_ = [("bases", C.POINTER(C.c_uint8)), ("offsets", C.POINTER(C.c_uint64)), ("n_reads", C.c_uint64), ("n_bases", C.c_uint64)]


class _Table(C.Structure):
    _fields_ = [("key_hi", C.POINTER(C.c_uint64)), ("key_lo", C.POINTER(C.c_uint64)), ("count", C.POINTER(C.c_uint64)),
                ("n_distinct", C.c_uint64), ("n_total", C.c_uint64), ("klen", C.c_int)]


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


_lib = None


def build():
    r = subprocess.run(["make", "-C", ORACLE_DIR, "all"], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB) or not os.path.exists(ORACLE_CLI):
            build()
        L = C.CDLL(ORACLE_LIB)
        L.kmo_parse_fasta.argtypes = [C.c_char_p, C.POINTER(_Reads)]
        L.kmo_free_reads.argtypes = [C.POINTER(_Reads)]
        L.kmo_free_table.argtypes = [C.POINTER(_Table)]
        L.kmo_count_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(_Table)]
        L.kmo_count_lr.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(_Table)]
        L.kmo_count_kmers_strings.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(_Table)]
        L.kmo_strerror.restype = C.c_char_p
        L.kmo_strerror.argtypes = [C.c_int]
        _lib = L
    return _lib


def _kmc():
    return importlib.import_module("k-mer-count_amd")


def _table_out(t):
    L = lib()
    try:
        n = int(t.n_distinct)
        def arr(p):
            return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
        return _kmc().Table(arr(t.key_hi), arr(t.key_lo), arr(t.count), int(t.klen))
    finally:
        L.kmo_free_table(C.byref(t))


def parse_fasta(path):
    L = lib()
    rd = _Reads()
    rc = L.kmo_parse_fasta(os.fsencode(path), C.byref(rd))
    if rc:
        raise OracleError(rc, L.kmo_strerror(rc).decode())
    try:
        nb, nr = int(rd.n_bases), int(rd.n_reads)
        # (np.ctypeslib.as_array on a POINTER is very slow for GB-sized buffers)
        bases = (np.frombuffer((C.c_uint8 * nb).from_address(C.addressof(rd.bases.contents)), dtype=np.uint8).copy()
                 if nb else np.zeros(0, np.uint8))
        offsets = np.frombuffer((C.c_uint64 * (nr + 1)).from_address(C.addressof(rd.offsets.contents)), dtype=np.uint64).copy()
    finally:
        L.kmo_free_reads(C.byref(rd))
    return bases, offsets


def _prep(bases, offsets):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    if bases.size == 0:
        bases = np.zeros(1, np.uint8)
    return bases, offsets, int(offsets.shape[0]) - 1


def count_kmers(bases, offsets, k, canonical=True, method=0):
    L = lib()
    b, o, n = _prep(bases, offsets)
    t = _Table()
    rc = L.kmo_count_kmers(b.ctypes.data, o.ctypes.data, n, k, 1 if canonical else 0, method, C.byref(t))
    if rc:
        raise OracleError(rc, L.kmo_strerror(rc).decode())
    return _table_out(t)


def count_kmers_strings(bases, offsets, k, canonical=True):
    L = lib()
    b, o, n = _prep(bases, offsets)
    t = _Table()
    rc = L.kmo_count_kmers_strings(b.ctypes.data, o.ctypes.data, n, k, 1 if canonical else 0, C.byref(t))
    if rc:
        raise OracleError(rc, L.kmo_strerror(rc).decode())
    return _table_out(t)


def count_lr(bases, offsets):
    L = lib()
    b, o, n = _prep(bases, offsets)
    t = _Table()
    rc = L.kmo_count_lr(b.ctypes.data, o.ctypes.data, n, C.byref(t))
    if rc:
        raise OracleError(rc, L.kmo_strerror(rc).decode())
    return _table_out(t)
