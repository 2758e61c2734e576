"""Multi-GPU plumbing: one process per GPU, independent chunks partitioned across ranks, ONE gather of the finished
streams to rank 0 (torch.distributed: backend "nccl" is RCCL over xGMI on MI355X; "gloo" on CPU for tests).

A single x3 stream does not shard (dictionary, recency order, contexts, models and the coder interval are one adaptive
chain over the whole input, x3.c:372-434), so the unit of distribution is the chunk: contiguous blocks
of chunks per rank, no data-path collective, and a variable-length gather at the end (sizes first, then payloads).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(nchunks: int, world: int, rank: int) -> range:
    """Contiguous block partition (BASELINE config 4: chunk c lives on GPU c // (nchunks/world))."""
    base, rem = divmod(nchunks, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def gather_streams(local: list[bytes], device: torch.device | str = "cpu", group=None, dst: int = 0):
    """Gather every rank's finished chunk streams on `dst`, in rank order (== global chunk order for shard_range).

    Step 1: all_gather of the per-rank (count, total bytes) and per-chunk lengths (tiny).
    Step 2: one gather of the concatenated payload, padded to the largest rank's byte count -- each peer owns a direct
            xGMI link to the root, so the step is bounded by max_rank(bytes)/link bandwidth, no ring needed.
    Returns the list of all streams on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device(device)
    lens = [len(s) for s in local]
    meta = torch.tensor([len(local), sum(lens)], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    totals = [int(m[1]) for m in metas]
    maxc, maxb = max(counts + [1]), max(totals + [1])

    lt = torch.zeros(maxc, dtype=torch.int64, device=dev)
    if lens:
        lt[:len(lens)] = torch.tensor(lens, dtype=torch.int64)
    all_lens = [torch.zeros(maxc, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_lens, lt, group=group)

    payload = torch.zeros(maxb, dtype=torch.uint8, device=dev)
    if totals[rank]:
        payload[:totals[rank]] = torch.from_numpy(np.frombuffer(b"".join(local), dtype=np.uint8).copy()).to(dev)
    bufs = [torch.zeros(maxb, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        blob = bufs[r][:totals[r]].cpu().numpy().tobytes()
        off = 0
        for i in range(counts[r]):
            ln = int(all_lens[r][i])
            out.append(blob[off:off + ln])
            off += ln
    return out


def gather_device_streams(d_out: torch.Tensor, stride: int, lens, group=None, dst: int = 0):
    """Same gather for streams that are still resident in HBM (bench path): d_out holds chunk i at [i*stride, i*stride+lens[i]).
    The payload is compacted on the device and gathered GPU-to-GPU over RCCL; only rank `dst` copies it to the host."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = d_out.device
    lens = [int(x) for x in lens]
    total = sum(lens)
    meta = torch.tensor([len(lens), total], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    totals = [int(m[1]) for m in metas]
    maxc, maxb = max(counts + [1]), max(totals + [1])
    lt = torch.zeros(maxc, dtype=torch.int64, device=dev)
    if lens:
        lt[:len(lens)] = torch.tensor(lens, dtype=torch.int64, device=dev)
    all_lens = [torch.zeros(maxc, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_lens, lt, group=group)
    payload = torch.zeros(maxb, dtype=torch.uint8, device=dev)
    off = 0
    for i, ln in enumerate(lens):
        payload[off:off + ln] = d_out[i * stride:i * stride + ln]
        off += ln
    bufs = [torch.zeros(maxb, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        blob = bufs[r][:totals[r]].cpu().numpy().tobytes()
        o = 0
        for i in range(counts[r]):
            ln = int(all_lens[r][i])
            out.append(blob[o:o + ln])
            o += ln
    return out
