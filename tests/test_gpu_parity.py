"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle (bit-exact --
integer work), the committed golden fixtures, and size-independent properties."""
import json
import importlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, SAMPLE

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))["cases"]
LR = json.load(open(os.path.join(GOLDEN, "lr_goldens.json")))["cases"]


def _algos(kmc, k, max_read_len=None):
    """Every algorithm that can take this input (WALK walks reads longer than 416 bases as
    overlapping pieces; it needs at least one non-empty read)."""
    a = [kmc.ALGO_STREAM, kmc.ALGO_AUTO, kmc.ALGO_SORT]
    if k <= 63 and (max_read_len is None or max_read_len >= 1):
        a.append(kmc.ALGO_WALK)
    return a


def _maxlen(offs):
    return int(np.diff(offs.astype(np.int64)).max()) if len(offs) > 1 else 0


def _count(kmc, bases, offs, k, canonical=True, algo=0, **kw):
    with kmc.KmerCounter(k=k, canonical=canonical, algo=algo, **kw) as kc:
        kc.add_batch(bases, offs)
        t = kc.export()
        st = kc.stats()
    return t, st


def _random_reads(rng, n_reads, lo, hi, alphabet=b"ACGT", p_bad=0.0):
    lens = rng.integers(lo, hi + 1, n_reads)
    offs = np.zeros(n_reads + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    n = int(offs[-1])
    bases = np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)].copy()
    if p_bad > 0 and n:
        bad = rng.random(n) < p_bad
        bases[bad] = np.frombuffer(b"NnacgtRY-*", np.uint8)[rng.integers(0, 10, int(bad.sum()))]
    return bases, offs


@pytest.mark.parametrize("k", ["5", "21", "31", "63"])
def test_sample_fasta_kats(kmc, oracle, k):
    """configs[0] plus the KAT table: reference fixture, k = 5/21/31/63, both strands."""
    kat = KAT[k]
    bases, offs = kmc.parse_fasta(SAMPLE)
    for canonical, tag in ((True, "canon"), (False, "fwd")):
        want = oracle.count_kmers(bases, offs, int(k), canonical)
        for algo in _algos(kmc, int(k), 400):
            t, st = _count(kmc, bases, offs, int(k), canonical, algo)
            assert t.n_total == kat["total"] and t.n_distinct == kat[f"distinct_{tag}"]
            assert int(t.count.max()) == kat[f"max_{tag}"]
            assert t.digest()[:16] == kat[f"digest_{tag}"]
            assert t.equals(want)
            assert st.n_kmers == kat["total"] and st.n_reads == 200 and st.n_bases == 80000


@pytest.mark.parametrize("k", [1, 2, 3, 4, 7, 11, 15, 16, 17, 24, 30, 31, 32, 33, 40, 47, 48, 49, 62, 63])
def test_every_key_width_boundary(kmc, oracle, k):
    rng = np.random.default_rng(100 + k)
    bases, offs = _random_reads(rng, 300, 0, 300)
    for canonical in (True, False):
        want = oracle.count_kmers(bases, offs, k, canonical)
        for algo in _algos(kmc, k, _maxlen(offs)):
            t, _ = _count(kmc, bases, offs, k, canonical, algo)
            assert t.equals(want), (k, canonical, algo)


@pytest.mark.parametrize("seed", range(6))
def test_ragged_reads_and_invalid_bytes(kmc, oracle, seed):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([5, 13, 21, 31, 33, 63]))
    bases, offs = _random_reads(rng, int(rng.integers(1, 2000)), 0, int(rng.choice([40, 150, 700, 3000])), p_bad=float(rng.choice([0, 0.001, 0.02])))
    for canonical in (True, False):
        want = oracle.count_kmers(bases, offs, k, canonical)
        for algo in _algos(kmc, k, _maxlen(offs)):
            t, _ = _count(kmc, bases, offs, k, canonical, algo)
            assert t.equals(want), (seed, k, canonical, algo)


@pytest.mark.parametrize("seed", range(4))
def test_walk_short_reads(kmc, oracle, seed):
    """KMC_ALGO_WALK on its own ground: ragged short reads (0..416), non-ACGT bytes (diverted to the
    scalar kernel), pool-like low cardinality and full-random high cardinality (memo overflow ->
    direct counting), every step-tail length."""
    rng = np.random.default_rng(200 + seed)
    for k in (3, 8, 9, 21, 31, 32, 40, 63):
        hi = int(rng.choice([20, 100, 250, 416]))
        bases, offs = _random_reads(rng, int(rng.integers(1, 6000)), 0, hi, p_bad=float(rng.choice([0, 0.002])))
        if seed % 2 == 0:
            # low cardinality: rebuild the reads from a pool of 7 lines of 40 bases
            pool = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (7, 40))]
            n = int(offs[-1])
            picks = rng.integers(0, 7, n // 40 + 2)
            bases = pool[picks].reshape(-1)[:n].copy()
        for canonical in (True, False):
            want = oracle.count_kmers(bases, offs, k, canonical)
            t, st = _count(kmc, bases, offs, k, canonical, kmc.ALGO_WALK)
            assert st.algo_last == kmc.ALGO_WALK
            assert t.equals(want), (seed, k, hi, canonical)


def test_walk_round_boundaries(kmc, oracle):
    """The walk kernel's load rounds are 320 sixteen-byte pieces: tiles whose byte range is exactly
    1, 2, 3, 4, 5 rounds (64 reads of 80, 160, 240, 320, 400 bases), one piece more or less, shifted
    off 16-byte alignment by a leading odd read, plus whole tiles of empty reads in the middle and at
    the end of the batch (a tile with no pieces at all)."""
    rng = np.random.default_rng(31)
    pool = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (5, 16))]
    for read_len in (80, 160, 240, 320, 400, 79, 81, 319, 321, 399, 401, 16, 15, 1):
        for lead in (0, 7):
            lens = ([lead] if lead else []) + [read_len] * 200 + [0] * 130 + [read_len] * 70 + [0] * 200
            offs = np.zeros(len(lens) + 1, np.uint64)
            offs[1:] = np.cumsum(lens)
            n = int(offs[-1])
            bases = pool[rng.integers(0, 5, n // 16 + 2)].reshape(-1)[:n].copy()
            for k in (5, 31):
                want = oracle.count_kmers(bases, offs, k, True)
                t, st = _count(kmc, bases, offs, k, True, kmc.ALGO_WALK)
                assert st.algo_last == kmc.ALGO_WALK and t.equals(want), (read_len, lead, k)


@pytest.mark.parametrize("seed", range(4))
def test_walk_long_reads_as_pieces(kmc, oracle, seed):
    """Reads longer than 416 bases go through the walk kernel as pieces that overlap by k-1 bases
    (kmc_vreads_*): every window lies in exactly one piece, so the table is the oracle's.  Lengths
    around every piece boundary (416, 416 + S, ...), one read of 150 k bases, short and empty reads in
    between, non-ACGT bytes (pieces diverted to the scalar kernel), low and high cardinality."""
    rng = np.random.default_rng(700 + seed)
    for k in (1, 5, 31, 32, 63):
        S = 416 - (k - 1)
        lens = [417, 416, 0, 416 + S, 416 + S + 1, 415, 2 * S + k - 1, 2 * S + k, 3, 150_000, 1000, k - 1, k, 5000]
        lens += [int(x) for x in rng.integers(0, 3000, 40)]
        offs = np.zeros(len(lens) + 1, np.uint64)
        offs[1:] = np.cumsum(lens)
        n = int(offs[-1])
        if seed % 2 == 0:  # low cardinality: lines of 50 bases from a pool of 6
            pool = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (6, 50))]
            bases = pool[rng.integers(0, 6, n // 50 + 2)].reshape(-1)[:n].copy()
        else:
            bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        if seed >= 2:
            bad = rng.random(n) < 0.0005
            bases[bad] = ord("N")
        for canonical in (True, False):
            want = oracle.count_kmers(bases, offs, k, canonical)
            t, st = _count(kmc, bases, offs, k, canonical, kmc.ALGO_WALK)
            assert st.algo_last == kmc.ALGO_WALK
            assert t.equals(want), (seed, k, canonical)
            t, st = _count(kmc, bases, offs, k, canonical, kmc.ALGO_AUTO)
            assert t.equals(want), (seed, k, canonical, "auto", st.algo_last)
    # device-resident batch with max_read_len unknown (computed on the device), twice (additivity)
    torch = pytest.importorskip("torch")
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    with kmc.KmerCounter(k=63, algo=kmc.ALGO_WALK) as kc:
        want = oracle.count_kmers(bases, offs, 63, True)
        kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), len(lens), n, 0)
        kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), len(lens), n, 0)
        t = kc.export()
        assert np.array_equal(t.key_lo, want.key_lo) and np.array_equal(t.key_hi, want.key_hi) and np.array_equal(t.count, want.count * 2)


def test_low_complexity_and_palindromes(kmc, oracle):
    # homopolymers, dinucleotide repeats (revcomp-palindromic k-mers, canonical ties), all-T reads
    reads = [b"A" * 500, b"T" * 500, b"AT" * 300, b"ACGT" * 200, b"G" * 31, b"C" * 30, b"TTTTTTTTTT" * 7]
    bases = np.frombuffer(b"".join(reads), np.uint8)
    offs = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    for k in (4, 16, 31, 32, 63):
        for canonical in (True, False):
            want = oracle.count_kmers(bases, offs, k, canonical)
            for algo in _algos(kmc, k, _maxlen(offs)):
                t, _ = _count(kmc, bases, offs, k, canonical, algo)
                assert t.equals(want), (k, canonical, algo)


def test_edge_inputs(kmc, oracle):
    z = np.zeros(0, np.uint8)
    for k in (5, 31, 63):
        # no reads at all; only empty reads; reads shorter than k; exactly k
        for bases, offs in [(z, np.array([0], np.uint64)), (z, np.array([0, 0, 0], np.uint64)),
                            (np.frombuffer(b"ACGT", np.uint8), np.array([0, 2, 4], np.uint64)),
                            (np.frombuffer(b"ACGTTGCAAC" * 7, np.uint8)[:k], np.array([0, k], np.uint64))]:
            want = oracle.count_kmers(bases, offs, k, True)
            for algo in _algos(kmc, k):
                t, _ = _count(kmc, bases, offs, k, True, algo)
                assert t.equals(want)
    # one long read crossing many 1024-base chunks, read starts at every alignment of a chunk edge
    rng = np.random.default_rng(9)
    for cut in (1023, 1024, 1025, 1024 * 3 - 30, 1024 * 3 + 31):
        bases, _ = _random_reads(rng, 1, 5000, 5000)
        offs = np.array([0, cut, 5000], np.uint64)
        for k in (31, 63):
            want = oracle.count_kmers(bases, offs, k, True)
            t, _ = _count(kmc, bases, offs, k, True, kmc.ALGO_STREAM)
            assert t.equals(want), (cut, k)


def test_many_batches_and_reset(kmc, oracle):
    rng = np.random.default_rng(3)
    bases, offs = _random_reads(rng, 3000, 50, 400)
    want = oracle.count_kmers(bases, offs, 21, True)
    with kmc.KmerCounter(k=21) as kc:
        # feed in 7 batches cut at read boundaries: counting is additive (shard invariance)
        cuts = np.linspace(0, 3000, 8).astype(int)
        for a, b in zip(cuts[:-1], cuts[1:]):
            o = offs[a:b + 1] - offs[a]
            kc.add_batch(bases[int(offs[a]):int(offs[b])], o)
        t = kc.export()
        assert t.equals(want)
        assert kc.stats().n_batches == 7
        kc.reset()
        assert kc.export().n_distinct == 0
        kc.add_batch(bases, offs)
        assert kc.export().equals(want)
    # sorted runs (SORT batches) and hash-table contents (merged pairs) are combined by finalize
    for k in (21, 63):
        want = oracle.count_kmers(bases, offs, k, True)
        with kmc.KmerCounter(k=k, algo=kmc.ALGO_SORT) as kc, kmc.KmerCounter(k=k, algo=kmc.ALGO_STREAM) as other:
            cuts = np.linspace(0, 3000, 4).astype(int)
            for a, b in zip(cuts[:-1], cuts[1:]):
                kc.add_batch(bases[int(offs[a]):int(offs[b])], offs[a:b + 1] - offs[a])
            assert kc.export().equals(want)           # three runs merged
            assert kc.export().equals(want)           # finalize is repeatable
            other.add_batch(bases, offs)
            other.finalize()
            dhi, dlo, dcnt, n = other.export_device()
            kc.merge_pairs_device(dhi, dlo, dcnt, n)  # table part: everything once more
            t2 = kc.export()
            assert np.array_equal(t2.key_lo, want.key_lo) and np.array_equal(t2.count, 2 * want.count)
            kc.reset()
            assert kc.export().n_distinct == 0


def test_high_cardinality_growth_and_spill(kmc, oracle):
    """More distinct keys than the initial table: growth + spill path stays exact."""
    rng = np.random.default_rng(11)
    bases, offs = _random_reads(rng, 20000, 600, 600)   # 12 M bases, ~11.4 M distinct 31-mers
    want = oracle.count_kmers(bases, offs, 31, True, method=1)
    t, st = _count(kmc, bases, offs, 31, True, kmc.ALGO_STREAM)
    assert t.equals(want)
    assert st.table_capacity >= 2 * want.n_distinct
    # long reads, AUTO: starts on the stream kernel, sees > 1 new key per 5 k-mers, finishes by sorting
    t, st = _count(kmc, bases, offs, 31, True, kmc.ALGO_AUTO)
    assert t.equals(want) and st.algo_last == kmc.ALGO_SORT
    t, st = _count(kmc, bases, offs, 31, False, kmc.ALGO_SORT)
    assert t.equals(oracle.count_kmers(bases, offs, 31, False, method=1))
    # short reads, every k-mer new: AUTO starts with the walk kernel, sees its memo overflow and hands
    # the rest of the batch to the sort path mid-batch; explicit WALK counts directly, in
    # sub-batches.  Both stay exact.
    bases, offs = _random_reads(rng, 40000, 300, 400)
    for k in (31, 47):
        want = oracle.count_kmers(bases, offs, k, True, method=1)
        t, st = _count(kmc, bases, offs, k, True, kmc.ALGO_AUTO)
        assert t.equals(want) and st.algo_last == kmc.ALGO_SORT
        t, st = _count(kmc, bases, offs, k, True, kmc.ALGO_WALK)
        assert t.equals(want) and st.algo_last == kmc.ALGO_WALK
        # a second batch on the same ctx goes straight to the sort path (two sorted runs + table merged)
        with kmc.KmerCounter(k=k) as kc:
            kc.add_batch(bases, offs)
            kc.add_batch(bases, offs)
            t2 = kc.export()
            assert np.array_equal(t2.key_lo, want.key_lo) and np.array_equal(t2.count, 2 * want.count)


def test_device_resident_batches_and_synth(kmc, oracle):
    torch = pytest.importorskip("torch")
    s = kmc.Synth(seed=5)
    n = 5000
    hb, ho = kmc.synth_reads_host(s, 100, n)
    d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda")
    d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    kmc.synth_reads_device(s, 100, n, d_b.data_ptr(), d_o.data_ptr())
    assert np.array_equal(d_b[:n * 400].cpu().numpy(), hb) and np.array_equal(d_o.cpu().numpy().astype(np.uint64), ho)
    for k in (21, 31, 63):
        want = oracle.count_kmers(hb, ho, k, True)
        for algo in _algos(kmc, k, 400):
            with kmc.KmerCounter(k=k, algo=algo) as kc:
                kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400)
                assert kc.export().equals(want), (k, algo)
                kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 0)  # max_read_len unknown
                t2 = kc.export()
                assert np.array_equal(t2.count, want.count * 2) and np.array_equal(t2.key_lo, want.key_lo)


@pytest.mark.parametrize("small_max", [None, "131072"])
def test_small_table_finalize_sizes(kmc, monkeypatch, small_max):
    """The rank-sort finalize of small tables at its size boundaries (16 keys per workgroup of 1024 threads, the
    table through LDS in tiles of 4096 / 2048 keys, at most 131072 keys; 131073 takes the weighted radix sort), one-
    and two-word keys, repeated on the same ctx (the kernel drains the table and must leave its ticket clean; a
    table that outgrew the speculative grid is finalized again with the full one).  Once with the library's own
    cut-over to the radix sort (40 k keys: the rank sort is quadratic) and once with the kernel run up to its limit."""
    torch = pytest.importorskip("torch")
    if small_max: monkeypatch.setenv("KMC_FIN_SMALL_MAX", small_max)
    rng = np.random.default_rng(55)
    for k in (31, 63):
        with kmc.KmerCounter(k=k) as kc:
            for n in (1, 2, 15, 16, 17, 63, 64, 65, 127, 1023, 1024, 1025, 2047, 2048, 2049, 3350, 4096, 4097, 8191, 8192, 8193, 20000, 32767, 32768, 32769, 39999, 40000, 40001, 70000, 131071, 131072, 131073, 200000, 64):
                lo = rng.integers(0, 2**62, n, dtype=np.uint64)
                hi = rng.integers(0, 2**60, n, dtype=np.uint64) if k > 31 else np.zeros(n, np.uint64)
                if k > 31:
                    hi[: n // 2] = hi[0]  # equal high words: the order is decided by the low word
                cnt = rng.integers(1, 1000, n, dtype=np.uint64)
                key = (hi.astype(object) << 64) | lo.astype(object)
                assert len(set(key.tolist())) == n
                d_lo = torch.from_numpy(lo.astype(np.int64)).cuda()
                d_hi = torch.from_numpy(hi.astype(np.int64)).cuda()
                d_c = torch.from_numpy(cnt.astype(np.int64)).cuda()
                torch.cuda.synchronize()
                kc.reset()
                kc.merge_pairs_device(d_hi.data_ptr() if k > 31 else 0, d_lo.data_ptr(), d_c.data_ptr(), n)
                t = kc.export()
                order = np.lexsort((lo, hi))
                assert t.n_distinct == n and t.n_total == int(cnt.sum())
                assert np.array_equal(t.key_hi, hi[order]) and np.array_equal(t.key_lo, lo[order]) and np.array_equal(t.count, cnt[order]), (k, n)


def test_poll_and_forget_source(kmc, oracle):
    """kmc_poll brings the stats up to date without a finalize (the multi-GPU step packs the live
    table and never finalizes the counting ctx); kmc_forget_source drops the walk memo and the
    planner history without touching counts."""
    bases, offs = oracle.parse_fasta(SAMPLE)
    want = oracle.count_kmers(bases, offs, 31, True)
    with kmc.KmerCounter(k=31, algo=kmc.ALGO_WALK) as kc:
        kc.add_batch(bases, offs)
        kc.poll()
        st = kc.stats()
        assert st.n_kmers == want.n_total and st.kernel_ms_last > 0 and st.launches_last >= 1
        kc.forget_source(memo=True, history=True)
        kc.add_batch(bases, offs)          # counts survive; the second batch re-learns the memo
        t = kc.export()
        assert np.array_equal(t.key_lo, want.key_lo) and np.array_equal(t.count, want.count * 2)
        kc.reset()
        kc.forget_source(memo=True, history=False)
        kc.add_batch(bases, offs)
        assert kc.export().equals(want)


def test_merge_and_partition(kmc, oracle):
    """Multi-GPU reduce building blocks on one GPU: owner partition + merge == counting everything."""
    rng = np.random.default_rng(21)
    for k in (21, 63):
        bases, offs = _random_reads(rng, 2000, 100, 300)
        half = 1000
        want = oracle.count_kmers(bases, offs, k, True)
        with kmc.KmerCounter(k=k) as a, kmc.KmerCounter(k=k) as b, kmc.KmerCounter(k=k) as c:
            a.add_batch(bases[:int(offs[half])], offs[:half + 1])
            b.add_batch(bases[int(offs[half]):], offs[half:] - offs[half])
            a.finalize(); b.finalize()
            for src in (a, b):
                pb, dhi, dlo, dcnt = src.partition_device(4)
                assert pb[0] == 0 and pb[4] == src.export_device()[3]
                # every pair sits in its owner's range
                t = src.export()
                own = np.array([kmc.owner_of(int(h), int(l), 4) for h, l in zip(t.key_hi[:200], t.key_lo[:200])])
                assert own.min() >= 0 and own.max() < 4
                for p in range(4):
                    n = pb[p + 1] - pb[p]
                    if n:
                        c.merge_pairs_device((dhi + 8 * pb[p]) if dhi else 0, dlo + 8 * pb[p], dcnt + 8 * pb[p], n)
            assert c.export().equals(want)


def test_full_size_properties(kmc):
    """Size-independent checks at a size the oracle would not finish quickly: totals are
    analytic, sortedness, shard invariance, doubling the input doubles every count."""
    torch = pytest.importorskip("torch")
    s = kmc.Synth(seed=2)
    n = 2_000_000  # 0.8 G bases
    d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda")
    d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr())
    for k in (31, 63):
        tabs = []
        for algo in _algos(kmc, k, 400):
            with kmc.KmerCounter(k=k, algo=algo) as kc:
                kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400)
                t = kc.export()
                assert t.n_total == n * (400 - k + 1)
                keys = (t.key_hi.astype(object) << 64) | t.key_lo.astype(object)
                assert all(keys[i] < keys[i + 1] for i in range(len(keys) - 1))
                assert t.n_distinct <= 10 * (81 - k) + 100 * (k - 1)
                tabs.append(t)
                # two shards == one batch
                kc.reset()
                h = n // 2
                kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), h, h * 400, 400)
                d_o2 = (d_o[h:] - d_o[h]).contiguous()
                kc.add_batch_device(d_b.data_ptr() + h * 400, d_o2.data_ptr(), n - h, (n - h) * 400, 400)
                assert kc.export().equals(t)
        assert all(tabs[0].equals(x) for x in tabs[1:])


def test_cli_matches_oracle(kmc, oracle, tmp_path):
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "bin", "k-mer-count")
    for args in (["-k", "5"], ["-k", "31", "--forward"], ["-k", "21", "--expand"]):
        out = subprocess.run([exe, SAMPLE] + args, capture_output=True, check=True).stdout
        oargs = ["count", SAMPLE, args[1]] + [a for a in args[2:]]
        ref = subprocess.run([oracle.ORACLE_CLI] + oargs, capture_output=True, check=True).stdout
        assert out == ref, args
    r = subprocess.run([exe, str(tmp_path / "missing.fasta"), "-k", "5"], capture_output=True)
    assert r.returncode == 101 and r.stdout == b"" and b"Error during opening the file" in r.stderr


def test_rccl_reduce_single_rank(kmc, oracle):
    """The multi-GPU reduce (partition -> RCCL all-to-all -> merge) with a one-rank nccl group:
    exercises the device plumbing bench.py --gpus N uses (zero-copy views of library memory,
    all_to_all_single, kmc_merge_pairs_device).  More ranks need more GPUs (driver's 8-GPU run);
    the routing logic itself is covered by the world_size-2 gloo tests."""
    torch = pytest.importorskip("torch")
    import importlib
    import torch.distributed as dist
    kd = importlib.import_module("k-mer-count_amd.distributed")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 1000))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(77)
        for k in (31, 63):
            bases, offs = _random_reads(rng, 3000, 100, 400)
            want = oracle.count_kmers(bases, offs, k, True)
            with kmc.KmerCounter(k=k) as local, kmc.KmerCounter(k=k) as owner:
                local.add_batch(bases, offs)
                sent, got = kd.reduce_tables(local, owner)
                assert sent == got == want.n_distinct
                assert owner.export().equals(want)
        # ctxs on their OWN streams (no stream= argument) with a long batch queued right in front of the
        # slab pack: the collective must wait for the ctx stream (kmc_sync), not only for torch's stream --
        # otherwise it reads the cached slab buffer before the pack kernel has written it (stale / zero slab)
        s = kmc.Synth(seed=4)
        n = 1_500_000
        d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda")
        d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr())
        import analytic_oracle as ao
        want = ao.exact_table(4, 31, 0, n, True)
        with kmc.KmerCounter(k=31, algo=kmc.ALGO_STREAM) as local, kmc.KmerCounter(k=31) as owner:
            for rep in range(3):
                local.reset()
                owner.reset()
                local.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400)  # several ms of kernel time, asynchronous
                sent, got = kd.reduce_tables(local, owner)
                assert sent == got == want.n_distinct, rep
                assert owner.export().equals(want), rep
    finally:
        dist.destroy_process_group()


def test_slab_pack_and_merge_kernels(kmc, oracle):
    """kmc_pack_slab_device / kmc_merge_slabs_device against the numpy restatement of the slab
    layout (tests/slab_np.py): three shard tables -> three slabs -> every owner merges its share;
    the union is the table of everything.  One table is made too large for its slab: it must be
    marked oversize, skipped by the merge and counted."""
    torch = pytest.importorskip("torch")
    import slab_np
    bases, offs = oracle.parse_fasta(SAMPLE)
    n_reads = len(offs) - 1
    cuts = [0, 3, 70, n_reads]
    E = 8192  # (sample.fasta has 6,140 distinct 63-mers)
    for k in (31, 63):
        kw = 1 if k <= 31 else 2
        shards, ctxs = [], []
        for i in range(3):
            sb = bases[int(offs[cuts[i]]):int(offs[cuts[i + 1]])]
            so = offs[cuts[i]:cuts[i + 1] + 1] - offs[cuts[i]]
            shards.append(oracle.count_kmers(sb, so, k, True))
            kc = kmc.KmerCounter(k=k)
            kc.add_batch(sb, so)
            kc.finalize()
            ctxs.append(kc)
        words = ctxs[0].slab_words(E)
        assert words == slab_np.slab_words(kw, E)
        # garbage first: the pack kernel must write every word it owns (header + n pairs)
        gathered = torch.full((3 * words,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda")
        for i, kc in enumerate(ctxs):
            kc.pack_slab_device(gathered.data_ptr() + 8 * words * i, E)
            kc.finalize()  # (synchronises the ctx stream)
        host = gathered.cpu().numpy().view(np.uint64)
        for i, t in enumerate(shards):
            got = slab_np.unpack(host[i * words:(i + 1) * words], kw, E)
            assert got is not None and int(host[i * words]) == t.n_distinct and int(host[i * words + 1]) == t.n_total
            assert np.array_equal(got[0], t.key_hi) and np.array_equal(got[1], t.key_lo) and np.array_equal(got[2], t.count)
        # the same slabs straight from the live (not finalized, unsorted) tables
        live = torch.full((3 * words,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        fresh = []
        for i in range(3):
            sb = bases[int(offs[cuts[i]]):int(offs[cuts[i + 1]])]
            so = offs[cuts[i]:cuts[i + 1] + 1] - offs[cuts[i]]
            kc = kmc.KmerCounter(k=k)
            kc.add_batch(sb, so)
            kc.pack_slab_device(live.data_ptr() + 8 * words * i, E)  # no finalize first
            fresh.append(kc)
        for kc in fresh:
            kc.finalize()
            kc.close()
        hl = live.cpu().numpy().view(np.uint64)
        for i, t in enumerate(shards):
            got = slab_np.unpack(hl[i * words:(i + 1) * words], kw, E)
            assert got is not None and int(hl[i * words]) == t.n_distinct and int(hl[i * words + 1]) == t.n_total
            h2, l2, c2 = slab_np.merge_sorted([got[0]], [got[1]], [got[2]])
            assert np.array_equal(h2, t.key_hi) and np.array_equal(l2, t.key_lo) and np.array_equal(c2, t.count)
        want = oracle.count_kmers(bases, offs, k, True)
        owned = []
        for part in range(3):
            with kmc.KmerCounter(k=k) as ow:
                ow.merge_slabs_device(gathered.data_ptr(), 3, E, part, 3)
                t = ow.export()
                assert ow.stats().n_slabs_skipped == 0
                assert np.all(importlib.import_module("k-mer-count_amd.distributed").owner_np(t.key_hi, t.key_lo, 3) == part)
                owned.append(t)
        hi, lo, cnt = slab_np.merge_sorted([t.key_hi for t in owned], [t.key_lo for t in owned], [t.count for t in owned])
        assert np.array_equal(hi, want.key_hi) and np.array_equal(lo, want.key_lo) and np.array_equal(cnt, want.count)
        assert sum(t.n_distinct for t in owned) == want.n_distinct
        # oversize: shard 2 has more than 64 keys
        small = torch.zeros(3 * ctxs[0].slab_words(64), dtype=torch.int64, device="cuda")
        w64 = ctxs[0].slab_words(64)
        for i, kc in enumerate(ctxs):
            kc.pack_slab_device(small.data_ptr() + 8 * w64 * i, 64)
            kc.finalize()
        hs = small.cpu().numpy().view(np.uint64)
        n_over = sum(1 for t in shards if t.n_distinct > 64)
        assert n_over >= 1 and sum(1 for i in range(3) if hs[i * w64] == slab_np.OVERSIZE) == n_over
        with kmc.KmerCounter(k=k) as ow:
            ow.merge_slabs_device(small.data_ptr(), 3, 64, 0, 1)
            t = ow.export()
            assert ow.stats().n_slabs_skipped == n_over
            inline = [sh for sh in shards if sh.n_distinct <= 64]
            assert t.n_total == sum(sh.n_total for sh in inline)
        for kc in ctxs:
            kc.close()
    # a live table with more keys than the slab (or than the claimed-slot list) is oversize
    rng = np.random.default_rng(9)
    rb, ro = _random_reads(rng, 400, 100, 300)
    with kmc.KmerCounter(k=31) as kc:
        kc.add_batch(rb, ro)
        sl = torch.zeros(kc.slab_words(1 << 16), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        kc.pack_slab_device(sl.data_ptr(), 1 << 16)
        kc.finalize()
        assert int(sl[0].item()) == -1


def test_rccl_reduce_small_tables_on_torch_stream(kmc, oracle):
    """The path bench.py --gpus N takes for generator-style input: both ctxs run on a torch stream,
    the tables travel in one all-gather of slabs, no host synchronisation in between (one-rank nccl
    group here; the driver's 8-GPU run is the real thing)."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    kd = importlib.import_module("k-mer-count_amd.distributed")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(30500 + os.getpid() % 1000)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        bases, offs = oracle.parse_fasta(SAMPLE)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for k in (31, 63):
                want = oracle.count_kmers(bases, offs, k, True)
                with kmc.KmerCounter(k=k, stream=st.cuda_stream) as local, kmc.KmerCounter(k=k, stream=st.cuda_stream) as owner:
                    assert kd._same_stream(local, torch.device("cuda", 0))
                    for rep in range(3):  # repeated steps reuse the slab buffers
                        local.reset(); owner.reset()
                        local.add_batch(bases, offs)
                        sent, got = kd.reduce_tables(local, owner)
                        assert sent == got == want.n_distinct and owner.stats().n_slabs_skipped == 0
                        assert owner.export().equals(want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["G-1", "G-3", "G-full"])
def test_reference_mode_matches_reference_goldens(kmc, oracle, tmp_path, case):
    """KMC_MODE_LR on the GPU reproduces the digests of the reference's OWN output
    (test.py / main.rs:87-90 on sample.fasta), and the oracle's table."""
    g = LR[case]
    path = SAMPLE
    if g["head_lines"] is not None:
        path = str(tmp_path / "head.fasta")
        with open(SAMPLE, "rb") as f:
            open(path, "wb").write(b"".join(f.readlines()[:g["head_lines"]]))
    t = kmc.count_file(path, k=None)
    assert t.klen == 54 and t.n_distinct == g["distinct"] and t.n_total == g["lines"] and int(t.count.max()) == g["max_count"]
    assert t.digest(expand=True) == g["sha256"]
    bases, offs = kmc.parse_fasta(path)
    assert t.equals(oracle.count_lr(bases, offs))
    if case == "G-1":
        # one fixture is the reference's stdout itself (test.py executed in place by make_goldens.py)
        import gzip
        assert t.to_bytes(expand=True) == gzip.decompress(open(os.path.join(GOLDEN, "g1_expected.txt.gz"), "rb").read())


def test_reference_mode_cli_and_errors(kmc, oracle, tmp_path):
    import hashlib
    import shutil
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "bin", "k-mer-count")
    # no arguments: opens ./sample.fasta in the cwd like main.rs:44 and prints main.rs:88-90's lines
    shutil.copy(SAMPLE, tmp_path / "sample.fasta")
    out = subprocess.run([exe], cwd=tmp_path, capture_output=True, check=True).stdout
    assert hashlib.sha256(out).hexdigest() == LR["G-full"]["sha256"]
    # a non-ACGT character aborts the reference (main.rs:23): exit 101, nothing on stdout
    bad = tmp_path / "bad.fasta"
    bad.write_bytes(b">r\n" + b"ACGT" * 30 + b"N" + b"ACGT" * 30 + b"\n")
    r = subprocess.run([exe, str(bad)], capture_output=True)
    assert r.returncode == 101 and r.stdout == b"" and b"Unexpected charactor" in r.stderr
    with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
        b = np.frombuffer(b"ACGT" * 30 + b"N" + b"ACGT" * 30, np.uint8)
        kc.add_batch(b, np.array([0, b.size], np.uint64))
        with pytest.raises(kmc.KmcError) as e:
            kc.finalize()
        assert e.value.status == kmc.ERR_ALPHABET
    # ... but only where the reference looks (main.rs:17-23: the characters of the emitted chunks): a record of
    # exactly 80 bases has ONE chunk, L = [0, 27) and R = [53, 80); an N in the gap between them is never seen
    with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
        b = np.frombuffer(b"ACGT" * 10 + b"N" + b"CGT" + b"ACGT" * 9, np.uint8)
        assert b.size == 80 and b[40] == ord("N")
        o = np.array([0, b.size], np.uint64)
        kc.add_batch(b, o)
        t = kc.export()
        assert t.n_distinct == 1 and t.n_total == 1 and t.equals(oracle.count_lr(b, o))
    # header-only / short records: empty result (G-empty)
    with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
        b = np.frombuffer(b"ACGT" * 19, np.uint8)  # 76 < 80
        kc.add_batch(b, np.array([0, 0, b.size], np.uint64))
        assert kc.export().n_distinct == 0
    # ragged random records vs the oracle
    rng = np.random.default_rng(5)
    bases, offs = _random_reads(rng, 60, 0, 300)
    with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
        kc.add_batch(bases, offs)
        assert kc.export().equals(oracle.count_lr(bases, offs))


def test_count_file_pipeline_chunks(kmc, oracle, tmp_path, monkeypatch):
    """kmc_count_file = streaming reader + pinned double buffers + upload of the reader's pieces to
    their dense place + one batch per chunk.  Every chunking gives the oracle's table: sample.fasta
    cut into ~90 chunks (k = 5, 31, 63 and the reference's LR mode against its golden digest), and a
    60 MB generator-style file in 7 multi-threaded chunks."""
    import subprocess
    from conftest import ROOT
    bases, offs = oracle.parse_fasta(SAMPLE)
    for cb in ("1000", "30000", ""):
        if cb:
            monkeypatch.setenv("KMC_INGEST_CHUNK_BYTES", cb)
        else:
            monkeypatch.delenv("KMC_INGEST_CHUNK_BYTES", raising=False)
        for k in (5, 31, 63):
            want = oracle.count_kmers(bases, offs, k, True)
            with kmc.KmerCounter(k=k) as kc:
                nd, nt = kc.count_file(SAMPLE)
                assert (nd, nt) == (want.n_distinct, want.n_total)
                assert kc.export().equals(want), (cb, k)
                assert kc.stats().n_batches == (1 if not cb else -(-87492 // int(cb))) or kc.stats().n_batches > 1
        with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
            kc.count_file(SAMPLE)
            t = kc.export()
            assert t.n_distinct == LR["G-full"]["distinct"] and t.digest(expand=True) == LR["G-full"]["sha256"], cb
    # the reference aborts on a non-ACGT character (main.rs:23), whichever chunk holds it
    monkeypatch.setenv("KMC_INGEST_CHUNK_BYTES", "200")
    bad = tmp_path / "bad.fasta"
    bad.write_bytes(b">r0\n" + b"ACGT" * 40 + b"\n>r1\n" + b"ACGT" * 30 + b"N" + b"ACGT" * 30 + b"\n")
    with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
        with pytest.raises(kmc.KmcError) as e:
            kc.count_file(str(bad))
        assert e.value.status == kmc.ERR_ALPHABET
    with kmc.KmerCounter(k=31) as kc:  # count-table mode skips windows with the N instead
        b2, o2 = oracle.parse_fasta(str(bad))
        kc.count_file(str(bad))
        assert kc.export().equals(oracle.count_kmers(b2, o2, 31, True))
    with kmc.KmerCounter(k=31) as kc:
        with pytest.raises(kmc.KmcError) as e:
            kc.count_file(str(tmp_path / "missing.fasta"))
        assert e.value.status == kmc.ERR_IO
    # multi-threaded chunks
    exe = os.path.join(ROOT, "bin", "kmc-genfasta")
    p = tmp_path / "big.fasta"
    with open(p, "wb") as f:
        subprocess.run([exe, "--bytes", "60000000", "--seed", "13"], stdout=f, check=True)
    b3, o3 = oracle.parse_fasta(str(p))
    want = oracle.count_kmers(b3, o3, 31, True)
    monkeypatch.setenv("KMC_INGEST_CHUNK_BYTES", "9000000")
    with kmc.KmerCounter(k=31) as kc:
        kc.count_file(str(p))
        assert kc.export().equals(want) and kc.stats().n_batches == 7


def test_walk_repeatability_stress(kmc, oracle):
    """Same-size batches with different data every time through fresh contexts (freed device
    buffers get reused) -- guards against races in the LDS memo and against stale reads."""
    rng = np.random.default_rng(7)
    for it in range(40):
        k = int(rng.choice([17, 24, 31, 47]))
        lens = rng.integers(0, 301, 300) if it % 2 else np.full(300, 150)
        offs = np.zeros(301, np.uint64)
        offs[1:] = np.cumsum(lens)
        bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
        want = oracle.count_kmers(bases, offs, k, True)
        for algo in (kmc.ALGO_WALK, kmc.ALGO_STREAM, kmc.ALGO_WALK):
            t, _ = _count(kmc, bases, offs, k, True, algo)
            assert t.equals(want), (it, k, algo)


def test_garbage_after_the_batch_is_ignored(kmc, oracle):
    """Device-resident batches whose last 16-byte piece is followed by arbitrary bytes (the ABI only
    promises readability up to the next 16-byte boundary).  Regression test: a 3-bit LUT selector
    in the walk kernel's encoder let such bytes corrupt valid bases 1..4 of the last piece."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(99)
    for tail_fill in (0xFF, 0x7E, 0x0D, 0x5C, 0x00):
        for n_extra in (1, 5, 8, 13):
            lens = np.full(257, 150)
            lens[-1] = 150 + n_extra                  # make n_bases % 16 vary
            offs = np.zeros(258, np.uint64)
            offs[1:] = np.cumsum(lens)
            n = int(offs[-1])
            bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
            d_b = torch.full((n + 256,), tail_fill, dtype=torch.uint8, device="cuda")
            d_b[:n] = torch.from_numpy(bases).cuda()
            d_o = torch.from_numpy(offs.astype(np.int64)).cuda()
            for k in (21, 47):
                want = oracle.count_kmers(bases, offs, k, True)
                for algo in (kmc.ALGO_WALK, kmc.ALGO_STREAM, kmc.ALGO_SORT):
                    with kmc.KmerCounter(k=k, algo=algo) as kc:
                        kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), 257, n, int(lens.max()))
                        assert kc.export().equals(want), (hex(tail_fill), n_extra, k, algo)


@pytest.mark.parametrize("fasta_bytes,k,seed", [(1e9, 21, 1), (10e9, 31, 2), (10e9, 63, 2)])
def test_baseline_config_sizes(kmc, oracle, fasta_bytes, k, seed):
    """BASELINE.json configs 2, 3 and 5 at their FULL sizes, EXACT: the table must equal, key for key
    and count for count, the exact table of the whole record range (tests/analytic_oracle.py: line and
    adjacent-pair histograms over all 2.2 / 22 M records expanded through the pool; validated against
    the C oracle in the CPU suite).  Plus shard invariance: two record shards cut at an odd place give
    the same table as one batch."""
    torch = pytest.importorskip("torch")
    import analytic_oracle as ao
    s = kmc.Synth(seed=seed)
    n, _ = kmc.synth_records_for_bytes(s, int(fasta_bytes))
    d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda")
    d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr())
    want = ao.exact_table(seed, k, 0, n, True)
    assert want.n_total == n * (400 - k + 1)
    with kmc.KmerCounter(k=k) as kc:
        kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400)
        t = kc.export()
        assert kc.stats().algo_last == kmc.ALGO_WALK
        assert t.equals(want)
        kc.reset()
        h = (n // 2 // 64) * 64 + 17   # an odd cut, not on a tile boundary
        kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), h, h * 400, 400)
        d_o2 = (d_o[h:] - d_o[h]).contiguous()
        kc.add_batch_device(d_b.data_ptr() + h * 400, d_o2.data_ptr(), n - h, (n - h) * 400, 400)
        assert kc.export().equals(want)
    # the general kernels on a 1 GB-sized slice of the same stream (they are 40-400x slower than the
    # walk kernel on this input): exact as well
    m = min(n, 2_000_000)
    want_m = ao.exact_table(seed, k, 0, m, True)
    for algo in (kmc.ALGO_STREAM, kmc.ALGO_SORT):
        with kmc.KmerCounter(k=k, algo=algo) as kc:
            kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), m, m * 400, 400)
            assert kc.export().equals(want_m), algo


def test_config4_50GB_eight_shards_reduced_on_one_gpu(kmc, oracle):
    """BASELINE.json config 4 (50 GB synthetic FASTA, k=31, record batches sharded across 8 GPUs with
    a count-table reduce) at full size on ONE GPU: the 8 rank shards are counted one after the other,
    each packed into its slab exactly as a rank does before the all-gather; the 8 owners then merge
    their share of the gathered slabs.  Size-independent checks: the owners' partitions are
    disjoint, together they hold every k-mer of every shard (analytic total), and their union equals
    the table of all 8 shards counted into a single ctx; the key set is the oracle's."""
    torch = pytest.importorskip("torch")
    kd = importlib.import_module("k-mer-count_amd.distributed")
    import slab_np
    k, world, E = 31, 8, kd.SLAB_ENTRIES
    s = kmc.Synth(seed=3)
    n_all, _ = kmc.synth_records_for_bytes(s, int(50e9))
    spans = [kd.shard_range(n_all, r, world) for r in range(world)]
    n_max = max(c for _, c in spans)
    d_b = torch.empty(n_max * 400 + 64, dtype=torch.uint8, device="cuda")
    d_o = torch.empty(n_max + 1, dtype=torch.int64, device="cuda")
    with kmc.KmerCounter(k=k) as rank_ctx, kmc.KmerCounter(k=k) as all_ctx:
        words = rank_ctx.slab_words(E)
        gathered = torch.zeros(world * words, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for r, (first, cnt) in enumerate(spans):
            kmc.synth_reads_device(s, first, cnt, d_b.data_ptr(), d_o.data_ptr())
            rank_ctx.reset()
            rank_ctx.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), cnt, cnt * 400, 400)
            rank_ctx.pack_slab_device(gathered.data_ptr() + 8 * words * r, E)   # live table, no finalize
            all_ctx.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), cnt, cnt * 400, 400)
            rank_ctx.poll()
            all_ctx.poll()          # both done with the buffers before the next shard overwrites them
            assert rank_ctx.stats().n_kmers == cnt * (400 - k + 1)
        whole = all_ctx.export()
        assert whole.n_total == n_all * (400 - k + 1)
        owned = []
        for p in range(world):
            with kmc.KmerCounter(k=k) as ow:
                ow.merge_slabs_device(gathered.data_ptr(), world, E, p, world)
                t = ow.export()
                assert ow.stats().n_slabs_skipped == 0
                assert np.all(kd.owner_np(t.key_hi, t.key_lo, world) == p)
                owned.append(t)
    assert sum(t.n_distinct for t in owned) == whole.n_distinct
    hi, lo, cnt = slab_np.merge_sorted([t.key_hi for t in owned], [t.key_lo for t in owned], [t.count for t in owned])
    assert np.array_equal(hi, whole.key_hi) and np.array_equal(lo, whole.key_lo) and np.array_equal(cnt, whole.count)
    # exact: the union of the 8 owners' partitions is the exact table of all 111.6 M records
    import analytic_oracle as ao
    want = ao.exact_table(3, k, 0, n_all, True)
    assert whole.equals(want)
    # and every rank shard on its own was exact too (slab r as rank r packed it)
    g = gathered.cpu().numpy().view(np.uint64)
    for r, (first, cnt) in enumerate(spans):
        sl = g[r * words:(r + 1) * words]
        nk = int(sl[0])
        order = np.argsort(sl[8:8 + nk], kind="stable")
        w = ao.exact_table(3, k, first, cnt, True)
        assert nk == w.n_distinct and np.array_equal(sl[8:8 + nk][order], w.key_lo) and np.array_equal(sl[8 + E:8 + E + nk][order], w.count), r


def test_count_file_on_several_contexts_and_cli_gpus(kmc, oracle, tmp_path, monkeypatch):
    """kmc_count_file_multi / `k-mer-count --gpus N`: the file's chunks go round-robin to N contexts
    (one per GPU; on this one-GPU box they share the device), the tables are reduced into the first
    by peer copies + merge.  Same table as one context, in count-table mode and in the reference's
    LR mode (golden digest of the reference's own output)."""
    import hashlib
    import subprocess
    from conftest import ROOT
    monkeypatch.setenv("KMC_INGEST_CHUNK_BYTES", "9000")
    bases, offs = oracle.parse_fasta(SAMPLE)
    for k in (31, 63):
        want = oracle.count_kmers(bases, offs, k, True)
        cs = [kmc.KmerCounter(k=k) for _ in range(3)]
        try:
            nd, nt = kmc.count_file_multi(cs, SAMPLE)
            assert (nd, nt) == (want.n_distinct, want.n_total)
            assert cs[0].export().equals(want)
            assert all(c.stats().n_batches >= 3 for c in cs)   # 10 chunks over 3 contexts
        finally:
            for c in cs:
                c.close()
    exe = os.path.join(ROOT, "bin", "k-mer-count")
    env = dict(os.environ, KMC_CLI_SHARE_DEVICE="0", KMC_INGEST_CHUNK_BYTES="9000")
    out = subprocess.run([exe, SAMPLE, "-k", "21", "--gpus", "4"], capture_output=True, check=True, env=env).stdout
    ref = subprocess.run([oracle.ORACLE_CLI, "count", SAMPLE, "21"], capture_output=True, check=True).stdout
    assert out == ref
    out = subprocess.run([exe, SAMPLE, "--gpus", "2"], capture_output=True, check=True, env=env).stdout
    assert hashlib.sha256(out).hexdigest() == LR["G-full"]["sha256"]
    # more GPUs than the box has: a clean error, exit code 101, nothing on stdout
    r = subprocess.run([exe, SAMPLE, "-k", "21", "--gpus", "64"], capture_output=True, env=dict(os.environ))
    assert r.returncode == 101 and r.stdout == b"" and b"kmc_create" in r.stderr


def test_count_file_fastq(kmc, oracle, tmp_path, monkeypatch):
    """FASTQ in, table out (reader extension, parity unpinned by the reference): the table equals the
    oracle's count over the sequence lines; reads carry N's (windows skipped) and are ragged."""
    import test_abi_host as th
    rng = np.random.default_rng(18)
    text, bases, offs = th._fastq_text(rng, 5000, max_len=400)
    p = tmp_path / "reads.fastq"
    p.write_bytes(text)
    for cb in ("", "100000"):
        if cb:
            monkeypatch.setenv("KMC_INGEST_CHUNK_BYTES", cb)
        for k in (21, 63):
            want = oracle.count_kmers(bases, offs, k, True)
            with kmc.KmerCounter(k=k) as kc:
                nd, nt = kc.count_file(str(p))
                assert (nd, nt) == (want.n_distinct, want.n_total) and kc.export().equals(want), (cb, k)


def test_auto_hands_over_long_high_cardinality_reads(kmc, oracle):
    """KMC_ALGO_AUTO starts long reads on the walk kernel (as pieces), sees after the first sub-batch
    that the memo does not help (random sequence: every k-mer new) and hands the REST of the batch
    to the sort path -- from the end of the last piece walked, in the middle of a read, so the
    windows that span the hand-over point must be counted exactly once."""
    rng = np.random.default_rng(99)
    lens = [100_000] * 40 + [777, 0, 5]
    offs = np.zeros(len(lens) + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
    bases[rng.integers(0, bases.size, 50)] = ord("N")
    for k in (31, 63):
        want = oracle.count_kmers(bases, offs, k, True, method=1)
        t, st = _count(kmc, bases, offs, k, True, kmc.ALGO_AUTO)
        assert st.algo_last == kmc.ALGO_SORT, st.algo_last   # it did switch
        assert st.launches_last >= 2
        assert t.equals(want), k


@pytest.mark.parametrize("k,pool,n_rec", [(31, 0, 100_000), (63, 0, 60_000), (31, 1000, 150_000), (21, 40, 100_000), (5, 0, 50_000), (33, 200, 80_000)])
def test_msd_sort_path_matches_oracle(kmc, oracle, k, pool, n_rec):
    """KMC_ALGO_SORT = extraction + the hand-written MSD radix sort + run-length (kmc_msd.hip.h), on inputs
    from all-distinct (pool 0) to heavily repeated keys (small pools: equal-key segments, merged
    leaves, sub-buckets larger than a wave), one- and two-word keys, both strands: the oracle's table."""
    s = kmc.Synth(seed=11, pool=pool)
    hb, ho = kmc.synth_reads_host(s, 3, n_rec)
    for canonical in (True, False):
        want = oracle.count_kmers(hb, ho, k, canonical, method=1)
        with kmc.KmerCounter(k=k, canonical=canonical, algo=kmc.ALGO_SORT) as kc:
            kc.add_batch(hb, ho)
            got = kc.export()
            assert kc.stats().algo_last == kmc.ALGO_SORT
        assert got.equals(want), (k, pool, canonical, got.n_distinct, want.n_distinct)
    # two batches into one ctx (two runs merged at finalize) == one batch
    h = n_rec // 3
    with kmc.KmerCounter(k=k, algo=kmc.ALGO_SORT) as kc:
        kc.add_batch(hb[:h * 400], ho[:h + 1])
        kc.add_batch(hb[h * 400:], ho[h:] - ho[h])
        assert kc.export().equals(oracle.count_kmers(hb, ho, k, True, method=1))


def test_walk_two_high_cardinality_batches(kmc, oracle):
    """Two batches of all-distinct reads into one ctx under KMC_ALGO_WALK (no hand-over to the sort path): the first
    batch leaves most of its counts in the (k+16)-mer table, which is unfolded at the start of the second batch; the
    launch planner then has to look at the table AFTER that unfold before it saves it in front of a risky launch.
    (It looked before: the first batch's counts were lost on recovery, or KMC_ERR_CAPACITY was raised.)"""
    k = 31
    hb, ho = kmc.synth_reads_host(kmc.Synth(seed=908266423, pool=0), 291, 120859)
    rng = np.random.default_rng(1)
    hbn = hb.copy()
    hbn[rng.integers(0, hb.size, size=hb.size // 5000)] = ord("N")
    cut = 11624
    c0 = int(ho[cut])
    for bases in (hb, hbn):
        want = oracle.count_kmers(bases, ho, k, True, method=1)
        for algo in (kmc.ALGO_WALK, kmc.ALGO_AUTO):
            with kmc.KmerCounter(k=k, algo=algo) as kc:
                kc.add_batch(bases[:c0], ho[:cut + 1])
                kc.add_batch(bases[c0:], ho[cut:] - ho[cut])
                assert kc.export().equals(want), algo


@pytest.mark.parametrize("k", [31, 63])
def test_sort_path_accumulates_batches(kmc, oracle, k, tmp_path):
    """KMC_ALGO_SORT only extracts a batch's keys behind those of the batches before it and sorts when the result is
    needed: batches of growing size (the accumulator is re-allocated with its contents kept), a finalize in the
    middle (one run so far), more batches (a second run, merged at the next finalize), reset (extracted keys are
    dropped with the runs) -- always the oracle's table of everything added since the last reset.  And a file
    read in many chunks (kmc_count_file sizes the accumulator from the file size) gives the whole file's table."""
    import subprocess
    from conftest import ROOT
    s = kmc.Synth(seed=77, pool=0)
    hb, ho = kmc.synth_reads_host(s, 0, 9000)
    cuts = [0, 500, 1500, 4000, 9000]
    def part(a, b):
        return hb[a * 400:b * 400], ho[a:b + 1] - ho[a]
    with kmc.KmerCounter(k=k, algo=kmc.ALGO_SORT) as kc:
        for a, b in zip(cuts[:3], cuts[1:3]):
            kc.add_batch(*part(a, b))
        assert kc.export().equals(oracle.count_kmers(*part(0, cuts[2]), k, True, method=1))          # finalize in the middle
        for a, b in zip(cuts[2:], cuts[3:]):
            kc.add_batch(*part(a, b))
        assert kc.export().equals(oracle.count_kmers(hb, ho, k, True, method=1))                     # run + new keys
        kc.reset()
        kc.add_batch(*part(0, 500))
        kc.reset()                                                                                   # extracted, never sorted: dropped
        kc.add_batch(*part(500, 1500))
        assert kc.export().equals(oracle.count_kmers(*part(500, 1500), k, True, method=1))
    p = tmp_path / "rnd.fasta"
    with open(p, "wb") as f:
        subprocess.run([os.path.join(ROOT, "bin", "kmc-genfasta"), "--bytes", "6000000", "--seed", "9", "--pool", "0"], stdout=f, check=True)
    fb, fo = oracle.parse_fasta(str(p))
    want = oracle.count_kmers(fb, fo, k, True, method=1)
    os.environ["KMC_INGEST_CHUNK_BYTES"] = "400000"
    try:
        with kmc.KmerCounter(k=k) as kc:
            nd, nt = kc.count_file(str(p))
            assert (nd, nt) == (want.n_distinct, want.n_total) and kc.export().equals(want)
            assert kc.stats().n_batches >= 10
    finally:
        del os.environ["KMC_INGEST_CHUNK_BYTES"]


def test_two_word_sort_with_both_leaf_sizes(kmc, oracle):
    """Two-word keys are sorted with leaves of 2048 keys -- until a sort of the ctx has collapsed its keys more
    than fourfold (heavily repeated keys: the smaller leaves are faster there), after which it uses leaves of
    1024 (kmc_msd_leaf_kernel<2, false, 1024>).  Both instantiations must give the oracle's table: the same ctx
    sorts a repetitive batch (sets the flag), the same batch again (small leaves), then a random one."""
    k = 63
    rep_b, rep_o = kmc.synth_reads_host(kmc.Synth(seed=5, pool=40), 0, 6000)    # 2.0 M k-mers, ~60 k distinct
    rnd_b, rnd_o = kmc.synth_reads_host(kmc.Synth(seed=6, pool=0), 0, 6000)     # 2.0 M k-mers, all distinct
    want_rep = oracle.count_kmers(rep_b, rep_o, k, True, method=1)
    want_rnd = oracle.count_kmers(rnd_b, rnd_o, k, True, method=1)
    assert want_rep.n_total >= (1 << 20) and want_rep.n_total >= 4 * want_rep.n_distinct
    with kmc.KmerCounter(k=k, algo=kmc.ALGO_SORT) as kc:
        for hb, ho, want in ((rep_b, rep_o, want_rep), (rep_b, rep_o, want_rep), (rnd_b, rnd_o, want_rnd), (rnd_b, rnd_o, want_rnd)):
            kc.reset()
            kc.add_batch(hb, ho)
            assert kc.export().equals(want)
            assert kc.stats().algo_last == kmc.ALGO_SORT


@pytest.mark.parametrize("algo_name", ["auto", "walk", "stream"])
def test_wrong_prediction_is_recovered_not_fatal(kmc, oracle, algo_name):
    """The launch planner sizes a batch's launches from what earlier batches (and earlier launches of the
    same batch) looked like.  A low-cardinality batch followed by a high-cardinality one on the same ctx,
    and a batch that starts with repeats / N runs and then turns random, used to exhaust table and spill
    area in one oversized launch and fail with KMC_ERR_CAPACITY.  Now the table is saved in front of such
    a launch, put back when the spill area overflows, and the rest of the batch is counted by sorting:
    exact tables, no error."""
    algo = {"auto": kmc.ALGO_AUTO, "walk": kmc.ALGO_WALK, "stream": kmc.ALGO_STREAM}[algo_name]
    k = 31
    lo_b, lo_o = kmc.synth_reads_host(kmc.Synth(seed=21, pool=10), 0, 300_000)      # 3.4 k distinct
    hi_b, hi_o = kmc.synth_reads_host(kmc.Synth(seed=22, pool=0), 0, 12_000)        # 4.4 M distinct, far beyond table + spill
    both_b = np.concatenate([lo_b, hi_b])
    both_o = np.concatenate([lo_o, hi_o[1:] + lo_o[-1]])
    want = oracle.count_kmers(both_b, both_o, k, True, method=1)
    # (1) two batches on one ctx: the second goes out in ONE launch on the history of the first
    with kmc.KmerCounter(k=k, algo=algo) as kc:
        kc.add_batch(lo_b, lo_o)
        kc.finalize()
        assert kc.stats().launches_last >= 1
        kc.add_batch(hi_b, hi_o)
        got = kc.export()
        assert got.equals(want)
        assert kc.stats().n_kmers == want.n_total
        # the ctx keeps working afterwards
        kc.reset()
        kc.add_batch(lo_b, lo_o)
        assert kc.export().equals(oracle.count_kmers(lo_b, lo_o, k, True, method=1))
    # (2) one batch: a long run of N and repeats first (launches ramp up on "no new keys"), then random reads
    n_b = np.full(400 * 50_000, ord("N"), np.uint8)
    n_o = (np.arange(50_001, dtype=np.uint64) * np.uint64(400))
    mix_b = np.concatenate([n_b, lo_b, hi_b])
    mix_o = np.concatenate([n_o, lo_o[1:] + n_o[-1], hi_o[1:] + n_o[-1] + lo_o[-1]])
    with kmc.KmerCounter(k=k, algo=algo) as kc:
        kc.forget_source(memo=True, history=True)
        kc.add_batch(mix_b, mix_o)
        assert kc.export().equals(want)


@pytest.mark.parametrize("k", [21, 31, 40, 47, 48, 63])
def test_walk_second_level_memo_kplus16_table(kmc, oracle, k, monkeypatch):
    """More contexts than the LDS memo holds (pools of 64 and 300 lines: thousands of nodes): full steps
    from k-mer contexts are counted as (k+16)-mers in the second-level table (two key words for k <= 47,
    three above) and unfolded once per launch.  Exact tables; almost nothing is counted k-mer by k-mer;
    and with the table made tiny (KMC_SK_SLOTS) it overflows, the launch is undone and sorted instead."""
    for pool, n_rec in ((64, 120_000), (300, 120_000)):
        s = kmc.Synth(seed=31 + pool, pool=pool)
        hb, ho = kmc.synth_reads_host(s, 0, n_rec)
        want = oracle.count_kmers(hb, ho, k, True, method=1)
        for algo in (kmc.ALGO_WALK, kmc.ALGO_AUTO):
            with kmc.KmerCounter(k=k, algo=algo) as kc:
                for rep in range(2):   # (the second pass runs on the learned memo + history: one launch)
                    kc.reset()
                    kc.add_batch(hb, ho)
                    got = kc.export()
                    st = kc.stats()
                    assert got.equals(want), (k, pool, algo, rep)
                    assert st.algo_last == kmc.ALGO_WALK and st.n_direct * 20 < want.n_total, (k, pool, algo, rep, st.n_direct)
    monkeypatch.setenv("KMC_SK_SLOTS", "4096")
    with kmc.KmerCounter(k=k, algo=kmc.ALGO_WALK) as kc:
        kc.add_batch(hb, ho)
        kc.finalize()
        kc.reset()
        kc.add_batch(hb, ho)          # one launch on history: the tiny table and its spill area overflow
        assert kc.export().equals(want)


@pytest.mark.parametrize("k", [31, 63])
def test_forget_source_keeps_pending_kplus16_counts(kmc, oracle, k, monkeypatch):
    """kmc_forget_source(KMC_FORGET_MEMO) between a WALK batch whose LDS memo overflowed and the next consumer: the
    (k+16)-mer table holds that batch's COUNTS until the deferred unfold, so forgetting the memo has to unfold them
    first (it used to memset them away: the table came out short, silently).  Pool of 64 lines = thousands of
    contexts, far more than the LDS memo holds; with and without a poll in between; counts of two batches add."""
    monkeypatch.setenv("KMC_SK_SLOTS", str(1 << 18))
    s = kmc.Synth(seed=4242, pool=64)
    hb, ho = kmc.synth_reads_host(s, 0, 60_000)
    want = oracle.count_kmers(hb, ho, k, True, method=1)
    for polled in (True, False):
        with kmc.KmerCounter(k=k, algo=kmc.ALGO_WALK) as kc:
            kc.add_batch(hb, ho)
            if polled:
                kc.poll()
            kc.forget_source(memo=True, history=False)
            got = kc.export()
            assert got.equals(want), (k, polled, got.n_total, want.n_total)
            kc.add_batch(hb, ho)                    # memo re-learned; counts of both batches
            kc.forget_source(memo=True, history=True)
            t = kc.export()
            assert np.array_equal(t.key_lo, want.key_lo) and np.array_equal(t.count, want.count * 2), (k, polled)


@pytest.mark.parametrize("k", [31, 63])
def test_count_file_multi_high_cardinality(kmc, oracle, k, tmp_path, monkeypatch):
    """kmc_count_file_multi on all-distinct input: every ctx's table is a sorted run (or table + run), so the
    source ctx's kmc_finalize goes through the MERGE branch whose last kernels used to be still queued when the
    destination's stream started its peer copies of the view.  Two and three ctxs on this one device."""
    import subprocess
    from conftest import ROOT
    p = tmp_path / "rnd.fasta"
    with open(p, "wb") as f:
        subprocess.run([os.path.join(ROOT, "bin", "kmc-genfasta"), "--bytes", "12000000", "--seed", "11", "--pool", "0"], stdout=f, check=True)
    fb, fo = oracle.parse_fasta(str(p))
    want = oracle.count_kmers(fb, fo, k, True, method=1)
    monkeypatch.setenv("KMC_INGEST_CHUNK_BYTES", "700000")
    for n_ctx in (2, 3):
        cs = [kmc.KmerCounter(k=k) for _ in range(n_ctx)]
        try:
            # a table entry next to the runs: the merge branch of kmc_finalize on every source ctx
            for c in cs:
                c.add_batch(fb[:800], fo[:3])
            nd, nt = kmc.count_file_multi(cs, str(p))
            w2 = oracle.count_kmers(np.concatenate([fb] + [fb[:800]] * n_ctx),
                                    np.concatenate([fo] + [fo[1:3] + fo[-1] + 800 * i for i in range(n_ctx)]).astype(np.uint64), k, True, method=1)
            assert (nd, nt) == (w2.n_distinct, w2.n_total), (k, n_ctx)
            assert cs[0].export().equals(w2), (k, n_ctx)
        finally:
            for c in cs:
                c.close()
    assert want.n_total > 0


def test_planner_randomised_batch_sequences(kmc, oracle):
    """A fixed-seed slice of tools/stress_sort_lr.py inside the suite (the launch planner of kmc_add_batch* steers with a
    dozen flags, two tables and a snapshot; its one real bug of round 2 was found by that tool, outside the suite):
    sequences of batches that mix generator pools of 10 / 40 / 1000 lines and all-distinct reads in ONE ctx, with
    finalizes in the middle (table drained into the view, then filled again), resets, polls, forgotten history, ragged
    reads and N bytes, under all four algorithms and k in {21, 31, 47, 48, 63} -- every look at the table must equal the
    oracle's count of everything added since the last reset, and the planner's debug invariant must hold (no risky
    launch armed on counters older than a queued unfold / merge: kmc_stats.n_planner_stale)."""
    rng = np.random.default_rng(int(os.environ.get("KMC_STRESS_SEED", "20261005")))   # (tools/r03_stress.sh runs other seeds)
    algos = [kmc.ALGO_AUTO, kmc.ALGO_WALK, kmc.ALGO_STREAM, kmc.ALGO_SORT]
    n_cases = int(os.environ.get("KMC_STRESS_CASES", "36"))
    for case in range(n_cases):
        k = int(rng.choice([21, 31, 47, 48, 63]))
        algo = algos[case % 4]
        canonical = bool(rng.integers(0, 2))
        acc_b, acc_o = [], [np.zeros(1, np.uint64)]

        def want_now():
            b = np.concatenate(acc_b) if acc_b else np.zeros(0, np.uint8)
            o = np.concatenate(acc_o).astype(np.uint64)
            return oracle.count_kmers(b, o, k, canonical, method=1)

        with kmc.KmerCounter(k=k, canonical=canonical, algo=algo) as kc:
            n_ops = int(rng.integers(2, 6))
            for op in range(n_ops):
                pool = int(rng.choice([0, 10, 40, 1000] if "KMC_STRESS_SEED" not in os.environ else [0, 10, 20, 40, 100, 1000]))
                n_rec = int(rng.integers(200, 70_000 if algo != kmc.ALGO_STREAM else 25_000))
                s = kmc.Synth(seed=int(rng.integers(1, 1 << 30)), pool=pool)
                hb, ho = kmc.synth_reads_host(s, int(rng.integers(0, 1000)), n_rec)
                if rng.integers(0, 4) == 0:      # ragged reads (some shorter than k, some empty)
                    lens = rng.integers(0, 401, size=n_rec)
                    hb = np.concatenate([hb[int(ho[i]):int(ho[i]) + int(lens[i])] for i in range(n_rec)]) if n_rec else hb
                    ho = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
                if rng.integers(0, 4) == 0 and hb.size:   # N bytes: windows skipped (8a-def)
                    hb = hb.copy()
                    hb[rng.integers(0, hb.size, size=max(1, hb.size // 4000))] = ord("N")
                kc.add_batch(hb, ho)
                base = acc_o[-1][-1]
                acc_b.append(hb)
                acc_o.append(ho[1:] + base)
                what = int(rng.integers(0, 6))
                if what == 0:                     # look at the table in the middle: drained into the view, filled again next op
                    assert kc.export().equals(want_now()), ("mid", case, op, k, algo, pool, n_rec)
                elif what == 1:
                    kc.poll()
                elif what == 2:
                    kc.forget_source(memo=bool(rng.integers(0, 2)), history=True)
                elif what == 3 and op + 1 < n_ops:  # start over
                    kc.reset()
                    acc_b, acc_o = [], [np.zeros(1, np.uint64)]
            got = kc.export()
            want = want_now()
            assert got.equals(want), ("end", case, k, algo, canonical, got.n_distinct, want.n_distinct, got.n_total, want.n_total)
            nd, nt = kc.finalize()                # a second finalize with nothing added: the same view
            assert (nd, nt) == (want.n_distinct, want.n_total)
            assert kc.stats().n_planner_stale == 0, (case, k, algo)


def test_finalize_async_small_tables(kmc, oracle):
    """kmc_finalize_async: the finalize of a small table is queued (sorted view on the device, table drained, counters
    published) and the call returns; kmc_reset behind it does not wait either; the next synchronising call takes the
    outcome from there.  Steps of count -> finalize_async -> reset deliver every time (n_async_ok), the last view is the
    table; a table that is NOT small falls back to the synchronous finalize."""
    s = kmc.Synth(seed=5)
    hb, ho = kmc.synth_reads_host(s, 0, 30_000)
    for k in (31, 63):
        want = oracle.count_kmers(hb, ho, k, True, method=1)
        with kmc.KmerCounter(k=k) as kc:
            kc.add_batch(hb, ho)
            assert kc.export().equals(want)           # (learn the source: later batches go out in one launch)
            ok0 = kc.stats().n_async_ok
            for step in range(6):
                kc.reset()
                kc.add_batch(hb, ho)
                kc.finalize_async()
            nd, nt = kc.finalize()                    # looks at the last queued finalize: no second sort
            assert (nd, nt) == (want.n_distinct, want.n_total)
            assert kc.export().equals(want)
            st = kc.stats()
            assert st.n_async_ok - ok0 == 6 and st.n_async_slabs_skipped == 0
            kc.finalize_async()                       # nothing new: a no-op
            kc.add_batch(hb, ho)                      # adding behind an unobserved finalize: the view goes back into the table
            t = kc.export()
            assert np.array_equal(t.key_lo, want.key_lo) and np.array_equal(t.count, want.count * 2)
            kc.finalize_async()
            kc.reset()                                # reset behind an unobserved finalize (drained or not: decided on the device)
            kc.add_batch(hb, ho)
            assert kc.export().equals(want)
    hb0, ho0 = kmc.synth_reads_host(kmc.Synth(seed=6, pool=0), 0, 20_000)   # 7 M distinct 31-mers: not a small table
    want0 = oracle.count_kmers(hb0, ho0, 31, True, method=1)
    with kmc.KmerCounter(k=31) as kc:
        kc.add_batch(hb0, ho0)
        kc.finalize_async()
        assert kc.export().equals(want0)
        kc.reset()
        kc.add_batch(hb, ho)
        kc.add_batch(hb0, ho0)
        kc.finalize_async()
        both = oracle.count_kmers(np.concatenate([hb, hb0]), np.concatenate([ho, ho0[1:] + ho[-1]]).astype(np.uint64), 31, True, method=1)
        assert kc.export().equals(both)


def test_msd_sort_counted_spans(kmc, oracle):
    """Spans of MORE keys than a leaf holds whose keys differ in their last <= 14 bits only end the MSD sort in an LDS
    histogram (kmc_msd.hip.h: kind-2 terminals, kmc_msd_count_kernel) instead of further levels + leaves: short keys
    (k = 8: 6 bits left after one level, fewer bins than threads; k = 12: 14 bits, the widest), a segment that is not
    moved because all its keys share the level's digit (poly-A reads that differ in their last seven bases), and the LR
    mode's rank pairs on the benchmark generator's reads (24-bit keys).  All against the oracle."""
    rng = np.random.default_rng(5)
    for k, n_rec in ((8, 20_000), (12, 12_000)):
        hb, ho = kmc.synth_reads_host(kmc.Synth(seed=4, pool=0), 0, n_rec)
        for canonical in (True, False):
            with kmc.KmerCounter(k=k, canonical=canonical, algo=kmc.ALGO_SORT) as kc:
                kc.add_batch(hb, ho)
                assert kc.export().equals(oracle.count_kmers(hb, ho, k, canonical, method=1)), (k, canonical)
    n_rec, rl = 5000, 100
    reads = np.full((n_rec, rl), ord("A"), dtype=np.uint8)
    reads[:, rl - 7:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_rec, 7))]
    hb, ho = reads.reshape(-1).copy(), np.arange(n_rec + 1, dtype=np.uint64) * rl
    for k in (31, 40):
        with kmc.KmerCounter(k=k, canonical=False, algo=kmc.ALGO_SORT) as kc:
            kc.add_batch(hb, ho)
            assert kc.export().equals(oracle.count_kmers(hb, ho, k, False, method=1)), k
    hb, ho = kmc.synth_reads_host(kmc.Synth(seed=2), 0, 600)
    with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
        kc.add_batch(hb, ho)
        assert kc.export().equals(oracle.count_lr(hb, ho))


def test_reference_mode_empty_and_short_reads(kmc, oracle):
    """LR mode's pair kernel finds the read of a window start from one 64-ary search per workgroup plus 257 staged read
    ends; more reads than that ending within a workgroup's 256 positions (runs of empty reads) take its general search.
    Runs of 0 / 300 / 3000 empty reads between reads of 0..79 bases (no key) and 80..400 bases: the oracle's table."""
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for n_empty in (0, 300, 3000):
        lens = []
        for _ in range(60):
            lens.append(int(rng.integers(80, 401)))
            lens.extend([0] * int(rng.integers(0, n_empty + 1)))
            lens.append(int(rng.integers(0, 80)))
        lens = np.array(lens, dtype=np.uint64)
        ho = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        hb = acgt[rng.integers(0, 4, int(ho[-1]))].copy()
        hb[300:900] = hb[1300:1900]          # repeated stretches: equal keys
        with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
            kc.add_batch(hb, ho)
            assert kc.export().equals(oracle.count_lr(hb, ho)), n_empty
