"""Multi-GPU plumbing: one process per GPU, independent chunks partitioned across ranks, ONE gather of the finished
streams to rank 0 (torch.distributed: backend "nccl" is RCCL over xGMI on MI355X; "gloo" on CPU for tests).

A single x3 stream does not shard (dictionary, recency order, contexts, models and the coder interval are one adaptive
chain over the whole input, x3.c:372-434), so the unit of distribution is the chunk: contiguous blocks
of chunks per rank, no data-path collective, and a single variable-length gather at the end.

The gather is ONE collective: every rank sends a fixed-size frame  [count, total, overflow, len_0 .. len_{maxc-1} | payload]
whose size all ranks can compute without talking to each other -- `slot_bytes` bounds a rank's payload (callers pass
raw shard bytes + 25 % + 4 KiB per chunk: an x3 stream of incompressible data grows by a few percent).  Each peer owns a
direct xGMI link to the root, so the step costs max_rank(frame)/link bandwidth; no ring, no size exchange first.  A rank
whose payload does not fit says so in its frame and the root raises OverflowError (the caller can retry with a larger slot).
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


def default_slot_bytes(raw_bytes: int, nchunks: int) -> int:
    """Payload bound of one rank: raw bytes of its shard + 25 % + 4 KiB per chunk, rounded up to 8."""
    return (int(raw_bytes) + int(raw_bytes) // 4 + 4096 * max(int(nchunks), 1) + 7) & ~7


def _frame(lens, payload_u8: torch.Tensor | None, maxc: int, slot_bytes: int, dev) -> torch.Tensor:
    """[count, total, overflow, lens[maxc]] as int64, then slot_bytes of payload, all in one uint8 tensor."""
    total = int(sum(lens))
    head = torch.zeros(3 + maxc, dtype=torch.int64)
    head[0], head[1], head[2] = len(lens), total, 1 if (total > slot_bytes or len(lens) > maxc) else 0
    if lens and len(lens) <= maxc:
        head[3:3 + len(lens)] = torch.tensor(lens, dtype=torch.int64)
    frame = torch.zeros(8 * (3 + maxc) + slot_bytes, dtype=torch.uint8, device=dev)
    frame[:8 * (3 + maxc)] = head.view(torch.uint8).to(dev)
    if payload_u8 is not None and total and not int(head[2]):
        frame[8 * (3 + maxc):8 * (3 + maxc) + total] = payload_u8[:total]
    return frame


def _unframe(bufs, world, maxc):
    hb = 8 * (3 + maxc)
    out = []
    for r in range(world):
        head = bufs[r][:hb].cpu().view(torch.int64)
        if int(head[2]):
            raise OverflowError(f"rank {r}: {int(head[1])} payload bytes / {int(head[0])} chunks do not fit the gather slot")
        cnt, total = int(head[0]), int(head[1])
        blob = bufs[r][hb:hb + total].cpu().numpy().tobytes()
        off = 0
        for i in range(cnt):
            ln = int(head[3 + i])
            out.append(blob[off:off + ln])
            off += ln
    return out


def gather_streams(local: list[bytes], slot_bytes: int, max_chunks: int, device: torch.device | str = "cpu", group=None, dst: int = 0):
    """Gather every rank's finished chunk streams on `dst`, in rank order (== global chunk order for shard_range): one collective.
    `slot_bytes` / `max_chunks`: bounds every rank agrees on (see default_slot_bytes; the largest shard's values).
    Returns the list of all streams on dst, None elsewhere; OverflowError on dst if some rank's streams did not fit."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device)
    lens = [len(s) for s in local]
    payload = torch.from_numpy(np.frombuffer(b"".join(local), dtype=np.uint8).copy()).to(dev) if sum(lens) else None
    frame = _frame(lens, payload, max_chunks, slot_bytes, dev)
    bufs = [torch.empty_like(frame) for _ in range(world)] if rank == dst else None
    dist.gather(frame, bufs, dst=dst, group=group)
    return _unframe(bufs, world, max_chunks) if rank == dst else None


def gather_device_streams(d_out: torch.Tensor, stride: int, lens, slot_bytes: int, max_chunks: int, group=None, dst: int = 0, to_host: bool = True):
    """Same single gather for streams that are still resident in HBM (bench path): d_out holds chunk i at [i*stride, i*stride+lens[i]).
    The payload is compacted on the device (one concat) and gathered GPU-to-GPU over RCCL; only rank `dst` copies it to the host
    (to_host=False: dst gets (lengths of all chunks, [payload of rank 0, payload of rank 1, ...] as device tensors) -- the bitstreams
    stay in HBM and only the frame heads, a few hundred bytes, cross PCIe)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = d_out.device
    lens = [int(x) for x in lens]
    total = sum(lens)
    payload = None
    if total and total <= slot_bytes:
        payload = torch.cat([d_out[i * stride:i * stride + ln] for i, ln in enumerate(lens) if ln])  # one device-side concat
    frame = _frame(lens, payload, max_chunks, slot_bytes, dev)
    bufs = [torch.empty_like(frame) for _ in range(world)] if rank == dst else None
    dist.gather(frame, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if to_host:
        return _unframe(bufs, world, max_chunks)
    # device-resident result: per-chunk lengths (from the small frame heads) + every rank's payload as a view into its frame
    hb = 8 * (3 + max_chunks)
    all_lens, views = [], []
    for r in range(world):
        head = bufs[r][:hb].cpu().view(torch.int64)
        if int(head[2]):
            raise OverflowError(f"rank {r}: {int(head[1])} payload bytes / {int(head[0])} chunks do not fit the gather slot")
        all_lens += [int(head[3 + i]) for i in range(int(head[0]))]
        views.append(bufs[r][hb:hb + int(head[1])])
    return all_lens, views
