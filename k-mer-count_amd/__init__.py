"""k-mer-count_amd -- Python host side of the MI355X k-mer counter.

A thin ctypes binding over the C ABI of ``libkmc.so`` (``include/kmc.h``).  Python is the
host language here because the reference's own toolchain (Rust) is absent from this image;
the names mirror the reference's pipeline, ``k-mer-count/src/main.rs:43-91`` and
``test.py:14-40``: a FASTA path goes in, a table sorted like ``lr_chunk.sort()``
(main.rs:87) comes out, printed one ``println!`` line per occurrence (main.rs:88-90) in
reference mode or ``KMER<TAB>COUNT`` in ``-k`` mode.

Import with ``importlib.import_module("k-mer-count_amd")`` (the directory name is the one the
project layout prescribes; it is not a Python identifier).

There is no CPU fallback anywhere in this package: every counting call goes through the HIP
kernels in ``libkmc.so`` and raises if the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from dataclasses import dataclass
from typing import Iterable, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMC_LIB_PATH: load another build of the same ABI (same-box A/B runs of kernel variants; tools/ab_bench.sh)
LIB_PATH = os.environ.get("KMC_LIB_PATH") or os.path.join(_HERE, "libkmc.so")

MODE_CONTIG, MODE_LR = 0, 1
ALGO_AUTO, ALGO_STREAM, ALGO_WALK, ALGO_SORT = 0, 1, 2, 3

OK = 0
ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_IO, ERR_FORMAT, ERR_ALPHABET, ERR_CAPACITY, ERR_STATE = range(-1, -10, -1)


class KmcError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libkmc status {status}: {message}")
        self.status = status
        self.message = message


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("k", C.c_int32), ("mode", C.c_int32), ("canonical", C.c_int32),
                ("device", C.c_int32), ("algo", C.c_int32), ("capacity_hint", C.c_uint64), ("stream", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_kmers", C.c_uint64), ("n_distinct", C.c_uint64),
                ("table_capacity", C.c_uint64), ("n_spilled", C.c_uint64), ("n_batches", C.c_uint64),
                ("kernel_ms_last", C.c_double), ("kernel_ms_total", C.c_double), ("algo_last", C.c_int32),
                ("launches_last", C.c_int32), ("n_slabs_skipped", C.c_uint64), ("n_direct", C.c_uint64),
                ("kernel_ms_lifetime", C.c_double), ("launches_lifetime", C.c_uint64),
                ("n_async_ok", C.c_uint64), ("n_async_slabs_skipped", C.c_uint64), ("n_planner_stale", C.c_uint64)]


class _Reads(C.Structure):
    _fields_ = [("bases", C.POINTER(C.c_uint8)), ("offsets", C.POINTER(C.c_uint64)), ("n_reads", C.c_uint64),
                ("n_bases", C.c_uint64), ("max_read_len", C.c_uint64)]


class Synth(C.Structure):
    """Parameters of the synthetic-input generator (random_fasta_generator.py:5-15 distribution)."""
    _fields_ = [("seed", C.c_uint64), ("pool", C.c_uint32), ("line_len", C.c_uint32),
                ("lines_per_record", C.c_uint32), ("reserved", C.c_uint32)]

    def __init__(self, seed=1, pool=10, line_len=80, lines_per_record=5):
        super().__init__(seed, pool, line_len, lines_per_record, 0)

    @property
    def read_len(self) -> int:
        return self.line_len * self.lines_per_record


# every symbol include/kmc.h declares
ABI_SYMBOLS = [
    "kmc_version", "kmc_status_string", "kmc_create", "kmc_destroy", "kmc_last_error", "kmc_reset",
    "kmc_add_batch", "kmc_add_batch_device", "kmc_merge_pairs_device", "kmc_finalize", "kmc_export",
    "kmc_export_device", "kmc_partition_device", "kmc_owner_of", "kmc_get_stats", "kmc_count_file",
    "kmc_parse_fasta", "kmc_free_reads", "kmc_decode_key", "kmc_synth_records_for_bytes",
    "kmc_synth_reads_host", "kmc_synth_reads_device", "kmc_synth_write_fasta",
    "kmc_slab_words", "kmc_pack_slab_device", "kmc_merge_slabs_device", "kmc_forget_source",
    "kmc_fasta_stream_open", "kmc_fasta_stream_next", "kmc_fasta_stream_close", "kmc_poll",
    "kmc_count_file_multi", "kmc_read_pieces", "kmc_sync", "kmc_read_peak_device", "kmc_finalize_async",
]

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libkmc.so and the CLI tools for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building libkmc.so failed")
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded libkmc.so.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `make -C {_HERE}` (or __graft_entry__.build()); "
                          "there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64/libhsa with the
    # same SONAMEs as /opt/rocm.  Importing torch first makes libkmc.so bind to the runtime torch
    # uses, so device pointers of torch tensors are valid inside libkmc (and torch still sees the
    # GPU); loading libkmc first would pin the system runtime and torch then finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    pu64 = C.POINTER(C.c_uint64)
    L.kmc_version.restype = C.c_char_p
    L.kmc_status_string.restype = C.c_char_p
    L.kmc_status_string.argtypes = [i32]
    L.kmc_create.argtypes = [C.POINTER(vp), C.POINTER(_Config)]
    L.kmc_destroy.argtypes = [vp]
    L.kmc_destroy.restype = None
    L.kmc_last_error.argtypes = [vp]
    L.kmc_last_error.restype = C.c_char_p
    L.kmc_reset.argtypes = [vp]
    L.kmc_add_batch.argtypes = [vp, vp, vp, u64]
    L.kmc_add_batch_device.argtypes = [vp, vp, vp, u64, u64, u64]
    L.kmc_merge_pairs_device.argtypes = [vp, vp, vp, vp, u64]
    L.kmc_finalize.argtypes = [vp, pu64, pu64]
    L.kmc_export.argtypes = [vp, vp, vp, vp, u64]
    L.kmc_export_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), pu64]
    L.kmc_partition_device.argtypes = [vp, u32, pu64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.kmc_owner_of.argtypes = [u64, u64, u32]
    L.kmc_owner_of.restype = u32
    L.kmc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.kmc_slab_words.argtypes = [vp, u64]
    L.kmc_slab_words.restype = u64
    L.kmc_pack_slab_device.argtypes = [vp, vp, u64]
    L.kmc_merge_slabs_device.argtypes = [vp, vp, u32, u64, u32, u32]
    L.kmc_forget_source.argtypes = [vp, i32]
    L.kmc_poll.argtypes = [vp]
    L.kmc_sync.argtypes = [vp]
    L.kmc_finalize_async.argtypes = [vp]
    L.kmc_read_peak_device.argtypes = [vp, u64, i32, vp, i32, i32, C.POINTER(C.c_double), pu64]
    L.kmc_read_pieces.argtypes = [u64, i32, vp, vp, u64]
    L.kmc_read_pieces.restype = u64
    L.kmc_count_file.argtypes = [vp, C.c_char_p, pu64, pu64]
    L.kmc_count_file_multi.argtypes = [C.POINTER(vp), u32, C.c_char_p, pu64, pu64]
    L.kmc_parse_fasta.argtypes = [C.c_char_p, C.POINTER(_Reads), C.c_char_p, C.c_size_t]
    L.kmc_free_reads.argtypes = [C.POINTER(_Reads)]
    L.kmc_free_reads.restype = None
    L.kmc_fasta_stream_open.argtypes = [C.c_char_p, u64, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.kmc_fasta_stream_next.argtypes = [vp, C.POINTER(_Reads), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    L.kmc_fasta_stream_close.argtypes = [vp]
    L.kmc_fasta_stream_close.restype = None
    L.kmc_decode_key.argtypes = [u64, u64, i32, C.c_char_p]
    L.kmc_decode_key.restype = None
    L.kmc_synth_records_for_bytes.argtypes = [C.POINTER(Synth), u64, pu64]
    L.kmc_synth_records_for_bytes.restype = u64
    L.kmc_synth_reads_host.argtypes = [C.POINTER(Synth), u64, u64, vp, vp]
    L.kmc_synth_reads_device.argtypes = [C.POINTER(Synth), u64, u64, vp, vp, i32, vp]
    L.kmc_synth_write_fasta.argtypes = [C.POINTER(Synth), u64, u64, vp]
    _lib = L
    return L


# ---------------------------------------------------------------------------------------------
# tables
# ---------------------------------------------------------------------------------------------
_CODE = np.frombuffer(b"ACGT", dtype=np.uint8)


@dataclass
class Table:
    """Sorted count table: ascending by (key_hi, key_lo) == string order (main.rs:87)."""
    key_hi: np.ndarray
    key_lo: np.ndarray
    count: np.ndarray
    klen: int

    @property
    def n_distinct(self) -> int:
        return int(self.key_lo.shape[0])

    @property
    def n_total(self) -> int:
        return int(self.count.sum(dtype=np.uint64)) if self.n_distinct else 0

    def kmers(self) -> np.ndarray:
        """(n_distinct, klen) uint8 ASCII matrix."""
        n, k = self.n_distinct, self.klen
        out = np.empty((n, k), dtype=np.uint8)
        lo = self.key_lo.astype(np.uint64).copy()
        hi = self.key_hi.astype(np.uint64).copy()
        for i in range(k - 1, -1, -1):
            out[:, i] = _CODE[(lo & np.uint64(3)).astype(np.intp)]
            lo = (lo >> np.uint64(2)) | (hi << np.uint64(62))
            hi = hi >> np.uint64(2)
        return out

    def to_bytes(self, expand: bool = False) -> bytes:
        """``KMER\\tCOUNT\\n`` lines, or with expand=True every key repeated COUNT times, one per
        line: byte-identical to the reference's output loop, main.rs:88-90."""
        n, k = self.n_distinct, self.klen
        if n == 0:
            return b""
        km = self.kmers()
        if expand:
            lines = np.empty((n, k + 1), dtype=np.uint8)
            lines[:, :k] = km
            lines[:, k] = 10
            return np.repeat(lines, self.count.astype(np.intp), axis=0).tobytes()
        parts = []
        for i in range(n):
            parts.append(km[i].tobytes() + b"\t%d\n" % int(self.count[i]))
        return b"".join(parts)

    def digest(self, expand: bool = False) -> str:
        """sha256 of to_bytes(), computed in slices (the expanded LR output is ~195 MB)."""
        h = hashlib.sha256()
        n = self.n_distinct
        step = 1 << 16
        for s in range(0, n, step):
            h.update(Table(self.key_hi[s:s + step], self.key_lo[s:s + step], self.count[s:s + step], self.klen).to_bytes(expand))
        return h.hexdigest()

    def equals(self, other: "Table") -> bool:
        return (self.klen == other.klen and self.n_distinct == other.n_distinct
                and np.array_equal(self.key_hi, other.key_hi) and np.array_equal(self.key_lo, other.key_lo)
                and np.array_equal(self.count, other.count))


def parse_fasta(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Host FASTA reader of libkmc (the reader the reference uses, main.rs:45-46,59-62).
    Returns (bases uint8[n_bases], offsets uint64[n_reads+1])."""
    L = lib()
    rd = _Reads()
    eb = C.create_string_buffer(256)
    rc = L.kmc_parse_fasta(os.fsencode(path), C.byref(rd), eb, 256)
    if rc:
        raise KmcError(rc, eb.value.decode() or L.kmc_status_string(rc).decode())
    try:
        nb, nr = int(rd.n_bases), int(rd.n_reads)
        # (np.ctypeslib.as_array on a POINTER is very slow for GB-sized buffers)
        bases = (np.frombuffer((C.c_uint8 * nb).from_address(C.addressof(rd.bases.contents)), dtype=np.uint8).copy()
                 if nb else np.zeros(0, np.uint8))
        offsets = np.frombuffer((C.c_uint64 * (nr + 1)).from_address(C.addressof(rd.offsets.contents)), dtype=np.uint64).copy()
    finally:
        L.kmc_free_reads(C.byref(rd))
    return bases, offsets


def stream_fasta(path: str, chunk_bytes: int = 0):
    """Streaming form of the host reader: yields (bases, offsets) per chunk of about chunk_bytes of
    FASTA text, each ending at a record boundary (copies: the stream reuses its buffers)."""
    L = lib()
    h = C.c_void_p()
    eb = C.create_string_buffer(256)
    rc = L.kmc_fasta_stream_open(os.fsencode(path), int(chunk_bytes), C.byref(h), eb, 256)
    if rc:
        raise KmcError(rc, eb.value.decode() or L.kmc_status_string(rc).decode())
    try:
        while True:
            rd = _Reads()
            eof = C.c_int(0)
            rc = L.kmc_fasta_stream_next(h, C.byref(rd), C.byref(eof), eb, 256)
            if rc:
                raise KmcError(rc, eb.value.decode() or L.kmc_status_string(rc).decode())
            nb, nr = int(rd.n_bases), int(rd.n_reads)
            bases = (np.frombuffer((C.c_uint8 * nb).from_address(C.addressof(rd.bases.contents)), dtype=np.uint8).copy()
                     if nb else np.zeros(0, np.uint8))
            offsets = np.frombuffer((C.c_uint64 * (nr + 1)).from_address(C.addressof(rd.offsets.contents)), dtype=np.uint64).copy()
            yield bases, offsets
            if eof.value:
                break
    finally:
        L.kmc_fasta_stream_close(h)


# ---------------------------------------------------------------------------------------------
# the counter
# ---------------------------------------------------------------------------------------------
class KmerCounter:
    """One counting context on one GPU (``kmc_ctx``).

    mode=MODE_LR reproduces the reference's computation (27+gap+27, chunk sizes 80..=140,
    main.rs:48-49,63); mode=MODE_CONTIG counts contiguous k-mers.
    """

    def __init__(self, k: int = 31, canonical: bool = True, mode: int = MODE_CONTIG, device: int = 0,
                 algo: int = ALGO_AUTO, capacity_hint: int = 0, stream: Optional[int] = None):
        L = lib()
        self._L = L
        cfg = _Config(C.sizeof(_Config), int(k), int(mode), 1 if canonical else 0, int(device), int(algo),
                      int(capacity_hint), C.c_void_p(stream) if stream else None)
        h = C.c_void_p()
        rc = L.kmc_create(C.byref(h), C.byref(cfg))
        if rc:
            raise KmcError(rc, L.kmc_last_error(None).decode())
        self._h = h
        self.stream = int(stream) if stream else 0  # hipStream_t the ctx runs on (0: its own stream)
        self.k = 54 if mode == MODE_LR else int(k)
        self.mode = mode
        self.device = int(device)

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None):
            self._L.kmc_destroy(self._h)
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

    def _chk(self, rc: int):
        if rc:
            raise KmcError(rc, self._L.kmc_last_error(self._h).decode())

    # -- feeding --
    def reset(self):
        self._chk(self._L.kmc_reset(self._h))

    def add_batch(self, bases: np.ndarray, offsets: np.ndarray):
        """Host buffers: ASCII bases of all reads concatenated + offsets[n_reads+1]."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = int(offsets.shape[0]) - 1
        self._chk(self._L.kmc_add_batch(self._h, bases.ctypes.data, offsets.ctypes.data, max(n_reads, 0)))

    def add_batch_device(self, d_bases: int, d_offsets: int, n_reads: int, n_bases: int, max_read_len: int = 0):
        """Device-resident buffers given as raw addresses (e.g. torch ``tensor.data_ptr()``)."""
        self._chk(self._L.kmc_add_batch_device(self._h, d_bases, d_offsets, int(n_reads), int(n_bases), int(max_read_len)))

    def add_batch_tensors(self, bases, offsets, max_read_len: int = 0):
        """torch tensors on this ctx's GPU: bases uint8[n_bases], offsets int64[n_reads+1]."""
        assert bases.is_cuda and offsets.is_cuda and bases.is_contiguous() and offsets.is_contiguous()
        self.add_batch_device(bases.data_ptr(), offsets.data_ptr(), offsets.numel() - 1, bases.numel(), max_read_len)

    def merge_pairs_device(self, d_key_hi: int, d_key_lo: int, d_count: int, n: int):
        self._chk(self._L.kmc_merge_pairs_device(self._h, d_key_hi or None, d_key_lo, d_count, int(n)))

    def count_file(self, path: str) -> Tuple[int, int]:
        nd, nt = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kmc_count_file(self._h, os.fsencode(path), C.byref(nd), C.byref(nt)))
        return nd.value, nt.value

    # -- results --
    def finalize(self) -> Tuple[int, int]:
        nd, nt = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kmc_finalize(self._h, C.byref(nd), C.byref(nt)))
        return nd.value, nt.value

    def finalize_async(self):
        """Queue the finalize of a small table and return without waiting (kmc_finalize_async)."""
        self._chk(self._L.kmc_finalize_async(self._h))

    def export(self) -> Table:
        nd, _ = self.finalize()
        hi = np.zeros(nd, np.uint64)
        lo = np.zeros(nd, np.uint64)
        cnt = np.zeros(nd, np.uint64)
        self._chk(self._L.kmc_export(self._h, hi.ctypes.data, lo.ctypes.data, cnt.ctypes.data, nd))
        return Table(hi, lo, cnt, self.k)

    def export_device(self) -> Tuple[int, int, int, int]:
        """(d_key_hi or 0, d_key_lo, d_count, n) of the sorted table of the last finalize."""
        a, b, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._chk(self._L.kmc_export_device(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value or 0, b.value or 0, c.value or 0, n.value

    def partition_device(self, n_parts: int):
        """Owner-partitioned view for the all-to-all: (part_begin[n_parts+1], d_hi, d_lo, d_cnt)."""
        pb = (C.c_uint64 * (n_parts + 1))()
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self._L.kmc_partition_device(self._h, n_parts, pb, C.byref(a), C.byref(b), C.byref(c)))
        return list(pb), a.value or 0, b.value or 0, c.value or 0

    # -- multi-GPU reduce, small tables (one fixed-size all-gather; distributed.py) --
    def slab_words(self, slab_entries: int) -> int:
        return int(self._L.kmc_slab_words(self._h, int(slab_entries)))

    def pack_slab_device(self, d_slab: int, slab_entries: int):
        """After finalize(): this table as one fixed-size slab (or an 'oversize' marker)."""
        self._chk(self._L.kmc_pack_slab_device(self._h, d_slab, int(slab_entries)))

    def merge_slabs_device(self, d_slabs: int, n_slabs: int, slab_entries: int, my_part: int, n_parts: int):
        """Add every pair of the gathered slabs that this rank owns; oversize slabs are skipped
        and show up in stats().n_slabs_skipped after the next finalize()."""
        self._chk(self._L.kmc_merge_slabs_device(self._h, d_slabs, int(n_slabs), int(slab_entries), int(my_part), int(n_parts)))

    def poll(self):
        """Synchronise and read the device counters (stats, launch-planner history) without finalizing."""
        self._chk(self._L.kmc_poll(self._h))

    def sync(self):
        """Wait for everything queued on the ctx's stream (no counters are read)."""
        self._chk(self._L.kmc_sync(self._h))

    def forget_source(self, memo: bool = True, history: bool = False):
        self._chk(self._L.kmc_forget_source(self._h, (1 if memo else 0) | (2 if history else 0)))

    def stats(self) -> Stats:
        s = Stats()
        self._chk(self._L.kmc_get_stats(self._h, C.byref(s)))
        return s


def count_file_multi(counters, path: str) -> Tuple[int, int]:
    """One FASTA file on several ctxs of this process (normally one per GPU); counters[0] holds the
    reduced table afterwards."""
    L = lib()
    arr = (C.c_void_p * len(counters))(*[c._h for c in counters])
    nd, nt = C.c_uint64(), C.c_uint64()
    rc = L.kmc_count_file_multi(arr, len(counters), os.fsencode(path), C.byref(nd), C.byref(nt))
    if rc:
        raise KmcError(rc, L.kmc_last_error(counters[0]._h).decode())
    return nd.value, nt.value


def owner_of(key_hi: int, key_lo: int, n_parts: int) -> int:
    return int(lib().kmc_owner_of(int(key_hi), int(key_lo), int(n_parts)))


def count_file(path: str, k: Optional[int] = None, canonical: bool = True, device: int = 0, algo: int = ALGO_AUTO) -> Table:
    """FASTA path in, sorted table out.  k=None is the reference's own computation (main.rs:58-90)."""
    mode = MODE_LR if k is None else MODE_CONTIG
    with KmerCounter(k=k or 54, canonical=canonical, mode=mode, device=device, algo=algo) as kc:
        kc.count_file(path)
        return kc.export()


# ---------------------------------------------------------------------------------------------
# synthetic input
# ---------------------------------------------------------------------------------------------
def synth_records_for_bytes(s: Synth, file_bytes: int) -> Tuple[int, int]:
    exact = C.c_uint64()
    n = lib().kmc_synth_records_for_bytes(C.byref(s), int(file_bytes), C.byref(exact))
    return int(n), int(exact.value)


def synth_reads_host(s: Synth, first_record: int, n_records: int) -> Tuple[np.ndarray, np.ndarray]:
    bases = np.empty(n_records * s.read_len, np.uint8)
    offsets = np.empty(n_records + 1, np.uint64)
    rc = lib().kmc_synth_reads_host(C.byref(s), first_record, n_records, bases.ctypes.data, offsets.ctypes.data)
    if rc:
        raise KmcError(rc, lib().kmc_status_string(rc).decode())
    return bases, offsets


def synth_reads_device(s: Synth, first_record: int, n_records: int, d_bases: int, d_offsets: int, device: int = 0, stream: int = 0):
    rc = lib().kmc_synth_reads_device(C.byref(s), first_record, n_records, d_bases, d_offsets, device, stream or None)
    if rc:
        raise KmcError(rc, lib().kmc_status_string(rc).decode())


def read_peak_device(d_buf: int, n_bytes: int, device: int = 0, stream: int = 0, shape: int = 0, iters: int = 5) -> Tuple[float, int]:
    """Measured streaming-read rate (kmc_read_peak_device): (ms per launch, xor checksum)."""
    ms, x = C.c_double(), C.c_uint64()
    rc = lib().kmc_read_peak_device(C.c_void_p(d_buf), int(n_bytes), int(device), C.c_void_p(stream), int(shape), int(iters), C.byref(ms), C.byref(x))
    if rc:
        raise KmcError(rc, "kmc_read_peak_device failed")
    return ms.value, x.value
