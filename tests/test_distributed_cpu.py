"""N > 1 path on CPU: world_size-2 `gloo` processes run the same exchange code bench.py uses
(k-mer-count_amd/distributed.py).  The per-rank tables come from the CPU oracle here (no GPU in
this container); on a GPU box the same functions move device tensors over RCCL."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SAMPLE

WORLD = 2


def _merge_np(his, los, cnts):
    """sum counts of equal (hi, lo) keys; sorted ascending (test-side reference merge)."""
    hi = np.concatenate(his); lo = np.concatenate(los); c = np.concatenate(cnts)
    order = np.lexsort((lo, hi))
    hi, lo, c = hi[order], lo[order], c[order]
    if hi.size == 0:
        return hi, lo, c
    new = np.ones(hi.size, bool)
    new[1:] = (hi[1:] != hi[:-1]) | (lo[1:] != lo[:-1])
    idx = np.cumsum(new) - 1
    out = np.zeros(int(idx[-1]) + 1, np.uint64)
    np.add.at(out, idx, c)
    return hi[new], lo[new], out


def _worker(rank, port, k, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        kmc = importlib.import_module("k-mer-count_amd")
        kd = importlib.import_module("k-mer-count_amd.distributed")
        import oracle_py
        bases, offs = oracle_py.parse_fasta(SAMPLE)
        n_reads = len(offs) - 1
        first, cnt = kd.shard_range(n_reads, rank, WORLD)
        sb = bases[int(offs[first]):int(offs[first + cnt])]
        so = offs[first:first + cnt + 1] - offs[first]
        local = oracle_py.count_kmers(sb, so, k, True)          # this rank's shard table
        owner = kd.owner_np(local.key_hi, local.key_lo, WORLD)
        # same function as the library's kmc_owner_of
        for i in range(0, local.n_distinct, max(1, local.n_distinct // 50)):
            assert kmc.owner_of(int(local.key_hi[i]), int(local.key_lo[i]), WORLD) == int(owner[i])
        as_t = lambda a: torch.from_numpy(a.astype(np.int64, copy=True))
        parts = lambda a: [as_t(a[owner == p]) for p in range(WORLD)]
        rhi, rlo, rcnt = kd.exchange_pairs(parts(local.key_hi), parts(local.key_lo), parts(local.count))
        assert all(int(t.numel()) == int(u.numel()) for t, u in zip(rlo, rcnt))
        to_np = lambda ts: [t.numpy().astype(np.uint64) for t in ts]
        mhi, mlo, mcnt = _merge_np(to_np(rhi), to_np(rlo), to_np(rcnt))
        # everything this rank now owns really is its partition
        assert np.all(kd.owner_np(mhi, mlo, WORLD) == rank)
        np.savez(os.path.join(tmpdir, f"owned{rank}.npz"), hi=mhi, lo=mlo, cnt=mcnt)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_reduce(rank, port, k, slab_entries, tmpdir, world=WORLD):
    """distributed.reduce_tables itself (slab all-gather, and the all-to-all behind it when a table
    does not fit its slab), the two ctxs replaced by the numpy stand-in of tests/slab_np.py."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kd = importlib.import_module("k-mer-count_amd.distributed")
        import oracle_py
        import slab_np
        bases, offs = oracle_py.parse_fasta(SAMPLE)
        n_reads = len(offs) - 1
        # uneven shards, so that with a small slab only ONE rank is oversize (mixed case)
        cut = 2  # (2 records: at most 2 x 370 distinct 31-mers)
        if world == 2:
            first, cnt = (0, cut) if rank == 0 else (cut, n_reads - cut)
        else:  # rank 0: 2 records, rank 1: 3 records, the last rank: everything else (the oversize one)
            bounds = [0, cut, cut + 3] + [n_reads] * (world - 2)
            first, cnt = bounds[rank], bounds[rank + 1] - bounds[rank]
        sb = bases[int(offs[first]):int(offs[first + cnt])]
        so = offs[first:first + cnt + 1] - offs[first]
        local = slab_np.CpuCtx(k, oracle_py.count_kmers(sb, so, k, True))
        owner = slab_np.CpuCtx(k)
        sent, got = kd.reduce_tables(local, owner, slab_entries=slab_entries)
        nd, nt = owner.finalize()
        assert got == nd and sent == len(local.lo)
        assert np.all(kd.owner_np(owner.hi, owner.lo, world) == rank)
        np.savez(os.path.join(tmpdir, f"owned{rank}.npz"), hi=owner.hi, lo=owner.lo, cnt=owner.cnt,
                 skipped=owner.stats().n_slabs_skipped, n_local=len(local.lo))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,slab_entries", [(31, 8192), (63, 8192), (31, 2000), (63, 16)])
def test_world2_reduce_tables_slab_and_fallback(oracle, tmp_path, k, slab_entries):
    port = 31000 + (os.getpid() + 7 * k + slab_entries) % 2000
    mp.spawn(_worker_reduce, args=(port, k, slab_entries, str(tmp_path)), nprocs=WORLD, join=True)
    parts = [np.load(tmp_path / f"owned{r}.npz") for r in range(WORLD)]
    import slab_np
    hi, lo, cnt = slab_np.merge_sorted([p["hi"] for p in parts], [p["lo"] for p in parts], [p["cnt"] for p in parts])
    bases, offs = oracle.parse_fasta(SAMPLE)
    want = oracle.count_kmers(bases, offs, k, True)
    assert np.array_equal(hi, want.key_hi) and np.array_equal(lo, want.key_lo) and np.array_equal(cnt, want.count)
    assert sum(len(p["lo"]) for p in parts) == want.n_distinct  # owners are disjoint
    # the case really exercised what its name says
    n_over = sum(1 for p in parts if int(p["n_local"]) > slab_entries)
    assert all(int(p["skipped"]) == n_over for p in parts)
    if slab_entries == 8192:
        assert n_over == 0
    if slab_entries == 2000:
        assert n_over == 1  # one table inline, one by all-to-all
    if slab_entries == 16:
        assert n_over == 2


@pytest.mark.parametrize("k", [21, 63])
def test_world2_reduce_equals_single_table(oracle, tmp_path, k):
    port = 29000 + (os.getpid() + k) % 2000
    mp.spawn(_worker, args=(port, k, str(tmp_path)), nprocs=WORLD, join=True)
    parts = [np.load(tmp_path / f"owned{r}.npz") for r in range(WORLD)]
    hi, lo, cnt = _merge_np([p["hi"] for p in parts], [p["lo"] for p in parts], [p["cnt"] for p in parts])
    bases, offs = oracle.parse_fasta(SAMPLE)
    want = oracle.count_kmers(bases, offs, k, True)
    assert np.array_equal(hi, want.key_hi) and np.array_equal(lo, want.key_lo) and np.array_equal(cnt, want.count)
    # partitions are disjoint: no key is owned twice
    assert sum(len(p["lo"]) for p in parts) == want.n_distinct


def test_shard_range_covers_everything():
    kd = importlib.import_module("k-mer-count_amd.distributed")
    for n in (0, 1, 7, 200, 22371032):
        for w in (1, 2, 3, 8):
            spans = [kd.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


@pytest.mark.parametrize("k,slab_entries,n_over_want", [(63, 3000, 1), (31, 8192, 0), (63, 8, 3)])
def test_world3_reduce_tables_two_word_keys_and_one_oversize_rank(oracle, tmp_path, k, slab_entries, n_over_want):
    """Three ranks (an odd world size: owner = mix(key) * 3 >> 32), two-word keys, uneven shards of which
    exactly one does not fit its slab (k=63, 3000 entries): the two small tables travel in the all-gather,
    the large one by the owner-partitioned all-to-all in which every rank takes part."""
    world = 3
    port = 33000 + (os.getpid() + 11 * k + slab_entries) % 2000
    mp.spawn(_worker_reduce, args=(port, k, slab_entries, str(tmp_path), world), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"owned{r}.npz") for r in range(world)]
    import slab_np
    hi, lo, cnt = slab_np.merge_sorted([p["hi"] for p in parts], [p["lo"] for p in parts], [p["cnt"] for p in parts])
    bases, offs = oracle.parse_fasta(SAMPLE)
    want = oracle.count_kmers(bases, offs, k, True)
    assert np.array_equal(hi, want.key_hi) and np.array_equal(lo, want.key_lo) and np.array_equal(cnt, want.count)
    assert sum(len(p["lo"]) for p in parts) == want.n_distinct  # owners are disjoint
    n_over = sum(1 for p in parts if int(p["n_local"]) > slab_entries)
    assert n_over == n_over_want and all(int(p["skipped"]) == n_over for p in parts)
