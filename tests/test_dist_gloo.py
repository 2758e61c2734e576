"""N>1 path on CPU: world_size-2 gloo, contiguous chunk sharding + the single variable-length gather + container assembly.
The per-chunk compressor is injected (here: the oracle), because the product compressor needs a GPU."""
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
from x3_compressor_amd import _lib, container, dist as xdist, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nchunks, chunk_bytes, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = synth.zipf_bytes(total).tobytes()
        offs = container.split_offsets(total, chunk_bytes)
        assert len(offs) - 1 == nchunks
        orc = oracle_lib.load()
        prm = oracle_lib.params(w_kib=1, t=4)
        mine = xdist.shard_range(nchunks, world, rank)
        local = [orc.compress(data[offs[i]:offs[i + 1]], prm) for i in mine]
        # bounds every rank can compute alone: the largest shard's raw bytes / chunk count
        shards = [xdist.shard_range(nchunks, world, r) for r in range(world)]
        slot = max(xdist.default_slot_bytes(sum(offs[i + 1] - offs[i] for i in sh), len(sh)) for sh in shards)
        streams = xdist.gather_streams(local, slot, max(len(sh) for sh in shards), "cpu")
        if rank == 0:
            raw = [offs[i + 1] - offs[i] for i in range(nchunks)]
            q.put(container.pack(streams, raw, _lib.make_params(w_kib=1, t=4)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nchunks,chunk_bytes,total", [(5, 3000, 13500), (2, 4096, 8192), (3, 2500, 7001)])
def test_two_rank_shard_and_gather(nchunks, chunk_bytes, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nchunks, chunk_bytes, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    blob = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single-process expectation
    data = synth.zipf_bytes(total).tobytes()
    offs = container.split_offsets(total, chunk_bytes)
    orc = oracle_lib.load()
    prm = oracle_lib.params(w_kib=1, t=4)
    want = [orc.compress(data[offs[i]:offs[i + 1]], prm) for i in range(nchunks)]
    params, chunks = container.unpack(blob)
    assert params["window_bytes"] == 1024 and [c[1] for c in chunks] == want
    # every chunk is a standalone x3 stream: decode and stitch
    back = b"".join(orc.decompress(s, raw + 64)[1] for raw, s in chunks)
    assert back == data


def _device_worker(rank, world, port, q):
    """gather_device_streams (what bench.py's N > 1 leg calls) with CPU tensors standing in for HBM: strided slots, ragged lengths"""
    import numpy as np
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stride, lens = 64, ([12, 0, 40] if rank == 0 else [64, 8])
        d_out = torch.zeros(stride * len(lens), dtype=torch.uint8)
        for i, ln in enumerate(lens):
            d_out[i * stride:i * stride + ln] = torch.arange(ln, dtype=torch.uint8) + 10 * rank + i
        streams = xdist.gather_device_streams(d_out, stride, np.array(lens, dtype=np.uint64), xdist.default_slot_bytes(200, 3), 3)
        got = xdist.gather_device_streams(d_out, stride, np.array(lens, dtype=np.uint64), xdist.default_slot_bytes(200, 3), 3, to_host=False)
        if rank == 0:  # the device-resident form carries the same bytes
            all_lens, views = got
            blob, off = torch.cat(views).numpy().tobytes(), 0
            for ln, s in zip(all_lens, streams):
                assert blob[off:off + ln] == s
                off += ln
            assert off == len(blob)
        q.put((rank, streams))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_of_strided_device_streams():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_device_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is None
    want = [bytes((j + 0) % 256 for j in range(12)), b"", bytes((j + 2) % 256 for j in range(40)),
            bytes((j + 10) % 256 for j in range(64)), bytes((j + 11) % 256 for j in range(8))]
    assert got[0] == want


def _overflow_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = [bytes(40)] if rank == 0 else [bytes(500), bytes(8)]   # rank 1 does not fit a 100-byte slot
        try:
            out = xdist.gather_streams(local, 100, 2, "cpu")
            q.put(("ok", rank, None if out is None else [len(s) for s in out]))
        except OverflowError as e:
            q.put(("overflow", rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_gather_slot_overflow_is_reported_on_the_root():
    """the single gather has no size exchange in front of it: a rank whose streams exceed the agreed slot is flagged in its frame"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == "ok" and got[0][1] == 1 and got[0][2] is None          # the sender carries on
    assert got[1][0] == "overflow" and got[1][1] == 0 and "rank 1" in got[1][2]  # the root refuses the frame


def test_shard_range_partitions_exactly():
    for n in (1, 7, 16, 128):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in xdist.shard_range(n, w, r)]
            assert got == list(range(n))
    assert list(xdist.shard_range(128, 8, 3)) == list(range(48, 64))  # config 4: chunk c on GPU c // 16


def test_container_single_chunk_is_raw_stream():
    s = b"\xff\x17\x00\x00"
    assert container.pack([s], [0], _lib.make_params()) == s
    assert container.unpack(s) == (None, [(None, s)])
