"""Multi-GPU path: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).

Reads are independent units (a window never leaves its record, reference
k-mer-count/src/main.rs:73-75) and counting is integer addition, so records shard across ranks
with no data-path collective while counting.  The one real exchange step is the reduce of the
per-GPU count tables.  Open-addressing layouts differ per GPU, so the tables are exchanged as
(key, count) pairs: each rank partitions its sorted table by owner = mix(key) % world
(kmc_partition_device), one all-to-all moves every partition to its owner (every GPU pair has
its own xGMI link, so all links are busy at once), and the owner merges what it receives
(kmc_merge_pairs_device).  The result stays partitioned by owner.

Tables of generator-style input are a few thousand keys, so that exchange is latency-bound: small
tables travel in ONE fixed-size all-gather of "slabs" instead (reduce_tables), with no size
exchange and no host synchronisation in between; the partitioned all-to-all remains for tables
that do not fit a slab.

The exchange itself only moves torch tensors, so the same code runs on gloo/CPU tensors in the
world_size-2 CPU tests (tests/test_distributed_cpu.py drives reduce_tables with a CPU stand-in for
the two ctxs that restates the slab layout in numpy).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [first, first+count) of n_items for `rank`."""
    base, rem = divmod(n_items, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def owner_np(key_hi: np.ndarray, key_lo: np.ndarray, n_parts: int) -> np.ndarray:
    """Vectorised kmc_owner_of (same arithmetic as kmc_table.hip.h:kmc_owner)."""
    with np.errstate(over="ignore"):
        z = key_lo.astype(np.uint64) ^ (key_hi.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return (((z >> np.uint64(32)) * np.uint64(n_parts)) >> np.uint64(32)).astype(np.int64)


class _DevArray:
    """Zero-copy view of library-owned device memory for torch.as_tensor()."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 3, "strides": None}


def device_view(ptr: int, n: int, device) -> torch.Tensor:
    """int64 torch view of n 64-bit words at device address ptr (bits are what matter).
    (device "cpu": host address -- used by the CPU stand-in of the world_size-2 gloo tests.)"""
    if n == 0 or not ptr:
        return torch.empty(0, dtype=torch.int64, device=device)
    if torch.device(device).type == "cpu":
        import ctypes
        return torch.from_numpy(np.frombuffer((ctypes.c_int64 * n).from_address(ptr), dtype=np.int64))
    return torch.as_tensor(_DevArray(ptr, n), device=device)


def _dev_of(ctx) -> torch.device:
    return torch.device("cuda", ctx.device) if ctx.device >= 0 else torch.device("cpu")


def all_to_all_v(send: Sequence[torch.Tensor], group=None) -> List[torch.Tensor]:
    """Variable-size all-to-all of 1-D int64 tensors: send[p] goes to rank p; returns what each
    rank sent here.  RCCL: one all_to_all_single for the sizes and one for the payload.  Gloo has
    no all-to-all, so there the same exchange is written as point-to-point sends."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = send[0].device
    sizes_out = torch.tensor([int(t.numel()) for t in send], dtype=torch.int64, device=dev)
    backend = dist.get_backend(group)
    if backend == "nccl":
        sizes_in = torch.empty_like(sizes_out)
        dist.all_to_all_single(sizes_in, sizes_out, group=group)
        in_splits = [int(x) for x in sizes_in.tolist()]
        out_splits = [int(x) for x in sizes_out.tolist()]
        payload = torch.cat(list(send)) if sum(out_splits) else torch.empty(0, dtype=torch.int64, device=dev)
        recv = torch.empty(sum(in_splits), dtype=torch.int64, device=dev)
        dist.all_to_all_single(recv, payload, output_split_sizes=in_splits, input_split_sizes=out_splits, group=group)
        return list(torch.split(recv, in_splits))
    if dev.type != "cpu":
        # gloo cannot move device tensors point-to-point: stage through the host (rehearsal path only)
        return [t.to(dev) for t in all_to_all_v([t.cpu() for t in send], group)]
    gathered = [torch.empty_like(sizes_out) for _ in range(world)]
    dist.all_gather(gathered, sizes_out, group=group)
    in_splits = [int(g[rank]) for g in gathered]
    recv = [torch.empty(n, dtype=torch.int64, device=dev) for n in in_splits]
    recv[rank].copy_(send[rank])
    ops = []
    for p in range(world):
        if p == rank:
            continue
        if send[p].numel():
            ops.append(dist.P2POp(dist.isend, send[p].contiguous(), p, group))
        if in_splits[p]:
            ops.append(dist.P2POp(dist.irecv, recv[p], p, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv


def exchange_pairs(parts_hi: Optional[Sequence[torch.Tensor]], parts_lo: Sequence[torch.Tensor],
                   parts_cnt: Sequence[torch.Tensor], group=None):
    """Send partition p of (hi, lo, cnt) to rank p.  The two or three arrays of a partition travel
    as ONE message ([lo | cnt | hi]), so the whole exchange is one size all-to-all plus one payload
    all-to-all.  Returns per-source lists (hi is None for one-word keys)."""
    nw = 3 if parts_hi is not None else 2
    world = len(parts_lo)
    send = []
    for p in range(world):
        chunks = [parts_lo[p], parts_cnt[p]] + ([parts_hi[p]] if parts_hi is not None else [])
        send.append(torch.cat(chunks) if parts_lo[p].numel() else parts_lo[p].new_empty(0))
    recv = all_to_all_v(send, group)
    recv_lo, recv_cnt, recv_hi = [], [], ([] if parts_hi is not None else None)
    for t in recv:
        m = int(t.numel()) // nw
        recv_lo.append(t[:m])
        recv_cnt.append(t[m:2 * m])
        if recv_hi is not None:
            recv_hi.append(t[2 * m:3 * m])
    return recv_hi, recv_lo, recv_cnt


SLAB_ENTRIES = 8192  # pairs per slab of the one-all-gather path (the generator's input: 2.6-6.4 k keys)


def _same_stream(ctx, dev) -> bool:
    """True when the ctx queues its work on torch's current stream: then kernels and collectives
    are ordered on the device and no host synchronisation is needed between them."""
    return bool(ctx.stream) and ctx.stream == torch.cuda.current_stream(dev).cuda_stream


def _order(ctx, dev, ctx_first: bool):
    """Order the ctx's stream and torch's current stream (the one the collective runs on) with respect
    to each other.  Nothing to do when the ctx queues its work on torch's current stream.  Otherwise
    ctx_first=True: everything queued on the ctx (the slab pack kernel) has finished before torch's
    stream goes on -- a ctx created without stream= runs on its OWN non-blocking stream, which a
    synchronize of torch's stream does not cover (the all-gather would read a slab that is still being
    written); ctx_first=False: torch's stream (the collective) has finished before the ctx goes on."""
    if dev.type != "cuda" or _same_stream(ctx, dev):
        return
    if ctx_first:
        ctx.sync()
    else:
        torch.cuda.current_stream(dev).synchronize()


_slab_cache = {}


def _slab_buffers(words: int, world: int, dev):
    key = (words, world, str(dev))
    if key not in _slab_cache:
        _slab_cache[key] = (torch.zeros(words, dtype=torch.int64, device=dev),
                            torch.zeros(words * world, dtype=torch.int64, device=dev))
    return _slab_cache[key]


def _all_gather_slabs(gathered: torch.Tensor, slab: torch.Tensor, group=None):
    if dist.get_backend(group) == "nccl" or slab.device.type == "cpu":
        dist.all_gather_into_tensor(gathered, slab, group=group)
    else:  # gloo with device tensors (one-GPU rehearsal): stage through the host
        h = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(h, slab.cpu(), group=group)
        gathered.copy_(h)


def reduce_tables(local, owner, group=None, slab_entries: int = SLAB_ENTRIES, report_sent: bool = True, finalize: bool = True):
    """The RCCL count-table reduce.  `local`: KmerCounter holding this rank's counts;
    `owner`: a second (reset) KmerCounter on the same GPU that receives the keys this rank owns.
    Afterwards owner.export() is this rank's partition of the global table.

    Small tables (at most `slab_entries` keys -- every table of generator-style input) move in ONE
    fixed-size all-gather of slabs (kmc_pack_slab_device / kmc_merge_slabs_device).  The slab is
    packed straight from the live table (a slab need not be sorted), so `local` is not finalized:
    count -> pack -> all-gather -> merge -> owner.finalize() is one stream of device work with a
    single host synchronisation at the end when both ctxs run on torch's current stream.  A rank
    whose table is too large marks its slab "oversize"; every rank sees that in the gathered
    headers (stats().n_slabs_skipped of the owner) and those tables then travel by the
    owner-partitioned all-to-all.  Returns (pairs_sent or None, pairs_owned); report_sent=False
    skips the extra read-back of this rank's own slab header.

    finalize=False (both ctxs on torch's current stream): nothing here waits for the GPU -- count, pack,
    all-gather, merge and the owner's finalize (kmc_finalize_async: sorted view on the device, table drained)
    are queued and the call returns (None, None).  Every such step still DELIVERS its sorted partition on the
    device; what is deferred is the host's look at it.  The caller synchronises when it wants (owner.finalize()
    returns the sizes of the last step's view) and then checks owner.stats(): n_async_ok must have grown by
    one per step and n_async_slabs_skipped must not have grown -- otherwise some step's table did not fit its
    slab (or the small-table path) and that step has to be redone with finalize=True (which routes such tables
    through the all-to-all), after owner.reset(): the owner already holds the slabs that did travel inline,
    and merging them a second time would double their counts."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = _dev_of(local)
    if not finalize and dev.type == "cuda" and not (_same_stream(local, dev) and _same_stream(owner, dev)):
        # checked BEFORE anything is queued: an error here leaves `owner` untouched
        raise ValueError("reduce_tables(finalize=False) needs both ctxs on torch's current stream")
    words = local.slab_words(slab_entries)
    slab, gathered = _slab_buffers(words, world, dev)
    local.pack_slab_device(slab.data_ptr(), slab_entries)
    _order(local, dev, ctx_first=True)      # slab written before the collective reads it
    _all_gather_slabs(gathered, slab, group)
    _order(owner, dev, ctx_first=False)     # gathered slabs landed before the owner's stream reads them
    owner.merge_slabs_device(gathered.data_ptr(), world, slab_entries, rank, world)
    if not finalize:
        # the owner's finalize is QUEUED (kmc_finalize_async): behind it on the stream this step's sorted, owned partition
        # is in place on the device and the owner's table is empty again; nothing here waits for the GPU
        owner.finalize_async()
        return None, None
    got, _ = owner.finalize()
    if local.stats().launches_last != 1:
        # `local` is never finalized on this path, so its launch planner would not learn what the
        # batch looked like; while it still splits a batch into several launches let it look (one
        # small read-back on an idle stream), afterwards nothing
        local.poll()
    skipped = owner.stats().n_slabs_skipped
    sent = None
    if report_sent or skipped:
        hdr = int(gathered[rank * words].item())  # this rank's own header as every rank saw it
        oversize = hdr == -1                      # (KMC_SLAB_OVERSIZE as int64)
        sent = None if oversize else hdr
    if not skipped:
        return sent, got
    # some table did not fit its slab: those ranks send theirs by the partitioned all-to-all
    # (every rank takes part; a rank whose table travelled inline sends nothing)
    n, got = _reduce_tables_a2a(local, owner, group, send=oversize)
    return (n if oversize else sent), got


def _reduce_tables_a2a(local, owner, group=None, send: bool = True):
    """Owner-partitioned all-to-all of (key, count) pairs (tables of any size)."""
    world = dist.get_world_size(group)
    dev = _dev_of(local)
    local.finalize()
    pb, d_hi, d_lo, d_cnt = local.partition_device(world)
    if not send:
        pb = [0] * (world + 1)
    n = pb[world]
    lo = device_view(d_lo, n, dev)
    cnt = device_view(d_cnt, n, dev)
    hi = device_view(d_hi, n, dev) if local.k > 31 else None
    cut = lambda t: [t[pb[p]:pb[p + 1]] for p in range(world)]
    recv_hi, recv_lo, recv_cnt = exchange_pairs(cut(hi) if hi is not None else None, cut(lo), cut(cnt), group)
    got = 0
    keep = []  # keep the received tensors alive until the merge kernels have run
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()  # payload landed before the ctx stream reads it
    for p in range(world):
        m = int(recv_lo[p].numel())
        if not m:
            continue
        rl, rc = recv_lo[p].contiguous(), recv_cnt[p].contiguous()  # (views of the receive buffer: no copy)
        rh = recv_hi[p].contiguous() if recv_hi is not None else None
        keep.append((rl, rc, rh))
        owner.merge_pairs_device(rh.data_ptr() if rh is not None else 0, rl.data_ptr(), rc.data_ptr(), m)
        got += m
    got_total, _ = owner.finalize()
    return n, got_total
