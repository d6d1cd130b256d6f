"""The N > 1 path on CPU: two gloo ranks shard independent images and gather the finished
bitstreams at rank 0 with the same StreamGather bench.py uses over RCCL.  The encoder itself
needs a GPU, so the per-rank "encode" here is the oracle (a checker standing in for the device);
what is under test is the sharding arithmetic and the padded gather."""
from __future__ import annotations

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank: int, world: int, port: int, n_images: int, ret):
    for p in (str(PKG / "python"), str(ROOT)):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jpegamd
    from jpegamd.sharding import StreamGather, owner_of, shard_range
    from oracle import oracle
    begin, end = shard_range(n_images, world, rank)
    cap = 4096
    gather = StreamGather(cap, torch.device("cpu"), dst=0, depth=2)
    collected = {}
    steps = max(shard_range(n_images, world, r)[1] - shard_range(n_images, world, r)[0] for r in range(world))
    handles = []
    for s in range(steps):
        payload = torch.zeros(cap, dtype=torch.uint8)
        size = torch.zeros(1, dtype=torch.int64)
        if begin + s < end:
            jf = oracle.encode_bmp(jpegamd.synth_bmp(24 + begin + s, 16, 50 + begin + s, 0, 0))
            payload[:len(jf)] = torch.frombuffer(bytearray(jf), dtype=torch.uint8)
            size[0] = len(jf)
        h = gather.start(payload, size, s)
        h.wait()
        if rank == 0:
            for r, stream in enumerate(gather.result(s)):
                b, e = shard_range(n_images, world, r)
                if b + s < e:
                    collected[b + s] = stream
                    assert owner_of(b + s, n_images, world) == r
    if rank == 0:
        ok = sorted(collected) == list(range(n_images))
        for i, stream in collected.items():
            ok &= stream == oracle.encode_bmp(jpegamd.synth_bmp(24 + i, 16, 50 + i, 0, 0))
        ret["ok"] = bool(ok)
        ret["n"] = len(collected)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    sys.path.insert(0, str(PKG / "python"))
    from jpegamd.sharding import shard_range
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_of_bitstreams():
    world, n_images = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_images, ret), nprocs=world, join=True)
    assert ret.get("ok") is True and ret.get("n") == n_images


def _batched_worker(rank: int, world: int, port: int, ret):
    for p in (str(PKG / "python"), str(ROOT)):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jpegamd
    from jpegamd.sharding import BatchedStreamGather
    from oracle import oracle
    slots, steps = 3, 9                                   # three full buffers through a depth-2 ring
    g = BatchedStreamGather(2048, slots, torch.device("cpu"), dst=0, depth=2)
    ok = True
    for s in range(steps):
        g.reserve(s)
        payload, size = g.record(s)
        jf = oracle.encode_bmp(jpegamd.synth_bmp(16 + s, 8 + 8 * rank, 7 * rank + s, 0, 0))
        payload[:len(jf)] = torch.frombuffer(bytearray(jf), dtype=torch.uint8)
        size[0] = len(jf)
        started = g.commit(s)
        assert (started is not None) == (s % slots == slots - 1)
        if started is not None:
            g.wait_all()
            if rank == 0:
                res = g.result(s)
                for r in range(world):
                    for k in range(slots):
                        step = s - slots + 1 + k
                        ok &= res[r][k] == oracle.encode_bmp(jpegamd.synth_bmp(16 + step, 8 + 8 * r, 7 * r + step, 0, 0))
    if rank == 0:
        ret["ok"] = bool(ok)
        ret["collectives"] = g.collectives
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_batched_gather():
    """bench.py's N > 1 exchange: several images per collective, sizes carried inside the records."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_batched_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret.get("ok") is True and ret.get("collectives") == 3


def _oversize_worker(rank, world, port, ret):
    """One stream of rank 1 does not fit its slot: settle() has that rank produce it again at the exact size."""
    for p in (str(PKG / "python"), str(ROOT)):
        sys.path.insert(0, p)
    import jpegamd
    from jpegamd.sharding import BatchedStreamGather
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slots = 4

    def stream_of(r, s):
        # record 2 of rank 1 is a 96x64 noise image (a few KB); the others are tiny
        return oracle.encode_bmp(jpegamd.synth_bmp(96, 64, 5, 1, 0) if (r, s) == (1, 2) else jpegamd.synth_bmp(16 + s, 8, 3 * r + s, 0, 0))

    slot_bytes = (max(len(stream_of(r, s)) for r in range(world) for s in range(slots) if (r, s) != (1, 2)) + 15) // 8 * 8 + 8
    assert len(stream_of(1, 2)) > slot_bytes
    g = BatchedStreamGather(slot_bytes, slots, torch.device("cpu"), dst=0, depth=2)
    for s in range(slots):
        payload, size = g.record(s)
        jf = stream_of(rank, s)
        n = min(len(jf), payload.numel())                 # what an encoder with too small a buffer leaves: a cut stream and the would-be size
        payload[:n] = torch.frombuffer(bytearray(jf[:n]), dtype=torch.uint8)
        size[0] = len(jf)
        g.commit(s)
    g.wait_all()

    def reencode(step, payload, size):
        jf = stream_of(rank, step)
        payload[:len(jf)] = torch.frombuffer(bytearray(jf), dtype=torch.uint8)
        size[0] = len(jf)

    n_over = g.settle(slots - 1, reencode)
    if rank == 0:
        res = g.result(slots - 1)
        ret["ok"] = all(res[r][s] == stream_of(r, s) for r in range(world) for s in range(slots))
        ret["n_over"] = n_over
        ret["collectives"] = g.collectives
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_with_an_oversized_stream():
    """A record that does not fit its slot no longer fails the buffer (a real configs[3] stream with one dense image):
    the offenders travel in a second, exact-size gather."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_oversize_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret.get("ok") is True and ret.get("n_over") == 1 and ret.get("collectives") == 2
