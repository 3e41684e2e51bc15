"""Property tests (hypothesis) of the CPU oracle and the host logic of libkmc -- no GPU needed."""
import numpy as np
from hypothesis import given, settings, strategies as st

ALPH = b"ACGTNacgt-"


@st.composite
def read_sets(draw):
    n = draw(st.integers(0, 12))
    reads = [draw(st.binary(min_size=0, max_size=90).map(lambda b: bytes(ALPH[x % len(ALPH)] for x in b))) for _ in range(n)]
    return reads


def _pack(reads):
    bases = np.frombuffer(b"".join(reads), np.uint8)
    offs = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    return bases, offs


def _naive(reads, k, canonical):
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    out = {}
    for r in reads:
        for i in range(len(r) - k + 1):
            w = r[i:i + k]
            if any(c not in b"ACGT" for c in w):
                continue
            if canonical:
                rc = bytes(comp[c] for c in reversed(w))
                w = min(w, rc)
            out[w] = out.get(w, 0) + 1
    return out


@settings(max_examples=120, deadline=None)
@given(read_sets(), st.integers(1, 40), st.booleans())
def test_oracle_equals_naive_definition(oracle, reads, k, canonical):
    """The oracle's three methods against a direct transcription of SURVEY.md 8a-def."""
    bases, offs = _pack(reads)
    want = _naive(reads, k, canonical)
    for t in (oracle.count_kmers(bases, offs, k, canonical, 0), oracle.count_kmers(bases, offs, k, canonical, 1),
              oracle.count_kmers_strings(bases, offs, k, canonical)):
        got = {km.tobytes(): int(c) for km, c in zip(t.kmers(), t.count)}
        assert got == want
        keys = [km.tobytes() for km in t.kmers()]
        assert keys == sorted(keys)                       # main.rs:87 order
        assert t.n_total == sum(want.values())


@settings(max_examples=60, deadline=None)
@given(read_sets(), read_sets(), st.integers(1, 33))
def test_counting_is_additive_over_record_shards(oracle, a, b, k):
    """Records are independent units (main.rs:58-62,73-75): table(a + b) == table(a) + table(b)."""
    ta = oracle.count_kmers(*_pack(a), k, True)
    tb = oracle.count_kmers(*_pack(b), k, True)
    tab = oracle.count_kmers(*_pack(a + b), k, True)
    merged = {}
    for t in (ta, tb):
        for h, l, c in zip(t.key_hi, t.key_lo, t.count):
            merged[(int(h), int(l))] = merged.get((int(h), int(l)), 0) + int(c)
    assert merged == {(int(h), int(l)): int(c) for h, l, c in zip(tab.key_hi, tab.key_lo, tab.count)}


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.binary(min_size=0, max_size=20), st.lists(st.binary(min_size=0, max_size=30), max_size=4)), max_size=6),
       st.sampled_from([b"\n", b"\r\n"]), st.booleans())
def test_host_reader_equals_oracle_reader(kmc, oracle, tmp_path_factory, records, eol, final_eol):
    """Random FASTA text: the product's (multi-threaded) reader and the oracle's restatement agree."""
    clean = lambda b: bytes(c for c in b if c not in b">\r\n")
    text = b""
    for hdr, lines in records:
        text += b">" + clean(hdr) + eol + b"".join(clean(l) + eol for l in lines)
    if not final_eol and text.endswith(eol):
        text = text[:-len(eol)]
    p = tmp_path_factory.mktemp("fa") / "x.fasta"
    p.write_bytes(text)
    try:
        b2, o2 = oracle.parse_fasta(str(p))
    except oracle.OracleError as e:
        try:
            kmc.parse_fasta(str(p))
            assert False, "product accepted what the oracle rejects"
        except kmc.KmcError as e2:
            assert (e.code, e2.status) in ((-2, kmc.ERR_FORMAT), (-1, kmc.ERR_IO))
        return
    b1, o1 = kmc.parse_fasta(str(p))
    assert np.array_equal(o1, o2) and np.array_equal(b1, b2)
    # the streaming reader, at a chunk size that cuts this text in several places
    for cb in (1, 9, 0):
        bs, os_, base = [], [np.zeros(1, np.uint64)], 0
        for b, o in kmc.stream_fasta(str(p), cb):
            bs.append(b)
            os_.append(o[1:] + np.uint64(base))
            base += int(b.shape[0])
        b3 = np.concatenate(bs) if bs else np.zeros(0, np.uint8)
        assert np.array_equal(np.concatenate(os_), o2) and np.array_equal(b3, b2), cb


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**62), st.integers(0, 10**7), st.integers(1, 300))
def test_synth_ranges_are_consistent(kmc, seed, first, n):
    s = kmc.Synth(seed=seed)
    b, o = kmc.synth_reads_host(s, first, n)
    b2, _ = kmc.synth_reads_host(s, first + n // 2, n - n // 2)
    assert np.array_equal(b[(n // 2) * 400:], b2)
    assert len(np.unique(b.reshape(-1, 80), axis=0)) <= 10


@settings(max_examples=300, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 63))
def test_walk_pieces_partition_the_windows(kmc, read_len, k):
    """KMC_ALGO_WALK walks a read longer than 416 bases as pieces that overlap by k-1 bases
    (kmc_read_pieces = the host copy of the device arithmetic).  Every window [i, i+k) of the read
    must lie completely inside exactly one piece, pieces are at most 416 bases, in order, inside
    the read; a short read is its own single piece."""
    L = kmc.lib()
    cap = 64
    starts = np.zeros(cap, np.uint64)
    ends = np.zeros(cap, np.uint64)
    n = int(L.kmc_read_pieces(read_len, k, starts.ctypes.data, ends.ctypes.data, cap))
    assert 1 <= n <= cap
    sp, ep = starts[:n].astype(np.int64), ends[:n].astype(np.int64)
    assert np.all(ep - sp <= 416) and np.all(sp >= 0) and np.all(ep <= read_len) and np.all(np.diff(sp) > 0 if n > 1 else True)
    if read_len <= 416:
        assert n == 1 and sp[0] == 0 and ep[0] == read_len
    owners = np.zeros(max(read_len - k + 1, 0), np.int64)
    for s_, e_ in zip(sp, ep):
        if e_ - s_ >= k:
            owners[s_:e_ - k + 1] += 1      # windows starting at s_ .. e_-k lie inside [s_, e_)
    assert np.all(owners == 1)


def test_analytic_oracle_equals_c_oracle_on_generated_bytes(kmc, oracle):
    """tests/analytic_oracle.py (exact tables of the synthetic generator at any size, from line and
    adjacent-pair histograms) against the C oracle counting the bytes libkmc's host generator makes:
    pins the numpy restatement of the generator AND the expansion.  This is what lets the full-size
    GPU tests (1 / 10 / 50 GB) assert exact table equality instead of bounds."""
    import analytic_oracle as ao
    for seed, k, first, n, canon, shape in (
            (1, 21, 0, 200_000, True, (10, 80, 5)), (2, 31, 1234, 200_000, True, (10, 80, 5)),
            (2, 63, 77, 100_000, True, (10, 80, 5)), (3, 31, 5, 50_000, False, (10, 80, 5)),
            (9, 5, 0, 20_000, True, (10, 80, 5)), (4, 1, 3, 5_000, True, (10, 80, 5)),
            (5, 31, 11, 30_000, True, (26, 80, 5)), (6, 63, 0, 20_000, True, (16, 64, 3)),
            (7, 17, 2, 20_000, False, (3, 16, 1)), (8, 33, 9, 10_000, True, (7, 32, 9))):
        pool, ll, lpr = shape
        s = kmc.Synth(seed=seed, pool=pool, line_len=ll, lines_per_record=lpr)
        hb, ho = kmc.synth_reads_host(s, first, n)
        want = oracle.count_kmers(hb, ho, k, canon, method=1)
        got = ao.exact_table(seed, k, first, n, canon, pool=pool, line_len=ll, lines_per_record=lpr)
        assert got.equals(want), (seed, k, shape)
    # histograms are additive over record ranges and independent of the chunking / threading
    U1, A1 = ao.line_histograms(2, 10, 5, 0, 1_000_003, chunk=50_000, threads=3)
    U2, A2 = ao.line_histograms(2, 10, 5, 0, 400_000)
    U3, A3 = ao.line_histograms(2, 10, 5, 400_000, 600_003, chunk=7_777, threads=1)
    assert np.array_equal(U1, U2 + U3) and np.array_equal(A1, A2 + A3)
    assert int(U1.sum()) == 5 * 1_000_003 and int(A1.sum()) == 4 * 1_000_003
