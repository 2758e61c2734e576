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
        streams = xdist.gather_streams(local, "cpu")
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
