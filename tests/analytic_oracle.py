"""Exact count tables for the synthetic generator at ANY size, in seconds -- TEST INFRASTRUCTURE ONLY
(a checker, like oracle/; never imported by the product path).

The benchmark input re-creates the distribution of the reference's generator
(/root/reference/random_fasta_generator.py:5-15): a pool of P random lines of LL bases (:5-8) and
records of LPR lines, each drawn from the pool (:13-15).  With k <= LL + 1 a window lies inside one
line or spans exactly two adjacent lines of one record, so

    count(kmer) =   sum over (p, x)      U[p]    * [kmer == pool[p][x : x+k]]                 x + k <= LL
                  + sum over (p, q, x)   A[p][q] * [kmer == (pool[p] + pool[q])[x : x+k]]     LL - k < x < LL

where U[p] = how many lines of the record range are pool line p, and A[p][q] = how many adjacent
line pairs INSIDE a record are (p, q).  U and A are histograms over all records (numpy, chunked:
22 M records take about two seconds, 114 M about ten); the expansion touches at most
P*(LL-k+1) + P*P*(k-1) windows (3,500 at k=31).  The result is the exact canonical (or forward)
table of the whole range -- what `count >= count-of-a-prefix` could only bound.

This file restates the generator's arithmetic (csrc/kmc_synth.hip.h: counter-based splitmix64) in
numpy / pure Python on purpose: it shares no code with libkmc, the HIP kernels or oracle/kmc_oracle.c.
tests/test_properties_cpu.py validates it against the C oracle counting the bytes that libkmc's host
generator produces (200 k records), which pins both the restated generator and the expansion.
"""
import importlib

import numpy as np

_M64 = (1 << 64) - 1
_GOLD = 0x9E3779B97F4A7C15
_STREAM = 0xD6E8FEB86659FD93
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _mix_int(seed, stream, ctr):
    z = (seed + stream * _STREAM) & _M64
    z = (z + (ctr + 1) * _GOLD) & _M64
    z = ((z ^ (z >> 30)) * _C1) & _M64
    z = ((z ^ (z >> 27)) * _C2) & _M64
    return z ^ (z >> 31)


def _mix_np(seed, stream, ctr):
    """ctr: uint64 array."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed + stream * _STREAM) & _M64)
        z = base + (ctr + np.uint64(1)) * np.uint64(_GOLD)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        return z ^ (z >> np.uint64(31))


def pool_lines(seed, pool, line_len):
    """The P pool lines as str."""
    return ["".join("ACGT"[_mix_int(seed, 0, p * line_len + x) >> 62] for x in range(line_len)) for p in range(pool)]


def choices(seed, pool, lines_per_record, first_record, n_records, _buf=None):
    """(n_records, lines_per_record) int64 array of pool indices (a view of _buf[0] when given)."""
    n = n_records * lines_per_record
    z = _buf[0][:n] if _buf is not None else np.empty(n, np.uint64)
    t = _buf[1][:n] if _buf is not None else np.empty(n, np.uint64)
    with np.errstate(over="ignore"):
        # the mix of kmc_synth_choice, in place over cache-sized chunks (fresh temporaries of hundreds of
        # MB per operation made the 10 GB configuration take 26 s)
        z[:] = np.arange(first_record * lines_per_record + 1, first_record * lines_per_record + 1 + n, dtype=np.uint64)
        z *= np.uint64(_GOLD)
        z += np.uint64((seed + 1 * _STREAM) & _M64)
        np.right_shift(z, np.uint64(30), out=t); z ^= t; z *= np.uint64(_C1)
        np.right_shift(z, np.uint64(27), out=t); z ^= t; z *= np.uint64(_C2)
        np.right_shift(z, np.uint64(31), out=t); z ^= t
        z >>= np.uint64(32)
        z *= np.uint64(pool)
        z >>= np.uint64(32)
    return z.view(np.int64).reshape(n_records, lines_per_record)


def _line_histograms_range(seed, pool, lines_per_record, first_record, n_records, chunk):
    U = np.zeros(pool, np.int64)
    A = np.zeros(pool * pool, np.int64)
    buf = (np.empty(chunk * lines_per_record, np.uint64), np.empty(chunk * lines_per_record, np.uint64))
    done = 0
    while done < n_records:
        m = min(chunk, n_records - done)
        c = choices(seed, pool, lines_per_record, first_record + done, m, buf)
        U += np.bincount(c.ravel(), minlength=pool)
        if lines_per_record > 1:
            pair = c[:, :-1] * pool
            pair += c[:, 1:]
            A += np.bincount(pair.ravel(), minlength=pool * pool)
        done += m
    return U, A


def line_histograms(seed, pool, lines_per_record, first_record, n_records, chunk=400_000, threads=None):
    """U[p] (lines that are pool line p) and A[p][q] (adjacent pairs inside a record) over the range.
    Record sub-ranges are histogrammed by a few threads (numpy releases the GIL) and added up."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    T = threads or max(1, min(8, os.cpu_count() or 1, n_records // chunk + 1))
    cuts = [first_record + n_records * i // T for i in range(T + 1)]
    with ThreadPoolExecutor(T) as ex:
        parts = list(ex.map(lambda i: _line_histograms_range(seed, pool, lines_per_record, cuts[i], cuts[i + 1] - cuts[i], chunk), range(T)))
    U = sum(p[0] for p in parts)
    A = sum(p[1] for p in parts)
    return U, A.reshape(pool, pool)


_COMP = str.maketrans("ACGT", "TGCA")
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def exact_table(seed, k, first_record, n_records, canonical=True, pool=10, line_len=80, lines_per_record=5):
    """The exact sorted count table (k-mer-count_amd.Table) of records [first_record, first_record+n_records)."""
    assert pool > 0, "pool == 0 (every line fresh random) has no small closed form"
    assert 1 <= k <= line_len + 1 and k <= 63
    U, A = line_histograms(seed, pool, lines_per_record, first_record, n_records)
    lines = pool_lines(seed, pool, line_len)
    table = {}

    def add(w, c):
        if c == 0:
            return
        if canonical:
            rc = w.translate(_COMP)[::-1]
            if rc < w:
                w = rc
        table[w] = table.get(w, 0) + int(c)

    for p in range(pool):
        for x in range(0, line_len - k + 1):
            add(lines[p][x:x + k], U[p])
    for p in range(pool):
        for q in range(pool):
            two = lines[p] + lines[q]
            for x in range(max(line_len - k + 1, 0), line_len):
                add(two[x:x + k], A[p][q])
    keys = sorted(table)
    n = len(keys)
    hi = np.zeros(n, np.uint64)
    lo = np.zeros(n, np.uint64)
    cnt = np.zeros(n, np.uint64)
    for i, w in enumerate(keys):
        v = 0
        for ch in w:
            v = (v << 2) | _CODE[ch]
        hi[i] = v >> 64
        lo[i] = v & _M64
        cnt[i] = table[w]
    assert int(cnt.sum()) == n_records * (lines_per_record * line_len - k + 1) or n_records * lines_per_record * line_len < k
    return importlib.import_module("k-mer-count_amd").Table(hi, lo, cnt, k)
