"""The N > 1 path on CPU: two gloo ranks shard independent images and gather the finished
bitstreams at rank 0 with the same gather classes bench.py uses over RCCL.  The encoder itself
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


def _exact_worker(rank: int, world: int, port: int, ret):
    for p in (str(PKG / "python"), str(ROOT)):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jpegamd
    from jpegamd.sharding import ExactStreamGather
    from oracle import oracle
    slots, steps = 3, 12                                  # four full buffers through a depth-3 ring

    def stream_of(r, s):
        return oracle.encode_bmp(jpegamd.synth_bmp(16 + s, 8 + 8 * r, 7 * r + s, 0, 0))

    g = ExactStreamGather(2048, slots, torch.device("cpu"), dst=0, depth=3)
    ok = True
    committed = 0
    for s in range(steps):
        g.reserve(s)
        payload, size = g.record(s)
        jf = stream_of(rank, s)
        payload[:len(jf)] = torch.frombuffer(bytearray(jf), dtype=torch.uint8)
        size[0] = len(jf)
        did = g.commit(s)
        assert did == (s % slots == slots - 1)
        committed += int(did)
        if did and s >= 2 * slots - 1:                    # the buffer committed one commit ago has been posted by now
            prev = s - slots
            g.reserve(prev)                               # (waits for that buffer's transfers; a second wait on a gloo work never returns)
            if rank == 0:
                res = g.result(prev)
                for r in range(world):
                    for k in range(slots):
                        ok &= res[r][k] == stream_of(r, prev - slots + 1 + k)
    g.drain()
    if rank == 0:
        res = g.result(steps - 1)
        for r in range(world):
            for k in range(slots):
                ok &= res[r][k] == stream_of(r, steps - slots + k)
        ret["ok"] = bool(ok)
        ret["exchanges"] = g.exchanges
    else:
        ret["sent"] = g.bytes_sent
        ret["exact"] = sum((len(stream_of(rank, s)) + 7) // 8 * 8 for s in range(steps))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_exact_size_gather():
    """bench.py's N > 1 exchange (gather-v): several images per exchange, every stream at its exact size, the size tables one
    buffer ahead of the transfers."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exact_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret.get("ok") is True and ret.get("exchanges") == 4
    assert ret.get("sent") == ret.get("exact") and ret.get("sent") < 12 * 2048      # what crossed the link is what the streams weigh


def _oversize_worker(rank, world, port, ret):
    """One stream of rank 1 -- and one of the root itself -- does not fit its slot: its owner produces it again at the exact size
    and THAT is what travels (no second exchange)."""
    for p in (str(PKG / "python"), str(ROOT)):
        sys.path.insert(0, p)
    import jpegamd
    from jpegamd.sharding import ExactStreamGather
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slots = 4
    big = {(1, 2), (0, 1)}

    def stream_of(r, s):
        # the big ones are 96x64 noise images (a few KB); the others are tiny
        return oracle.encode_bmp(jpegamd.synth_bmp(96, 64, 5 + r, 1, 0) if (r, s % slots) in big else jpegamd.synth_bmp(16 + s, 8, 3 * r + s, 0, 0))

    slot_bytes = (max(len(stream_of(r, s)) for r in range(world) for s in range(slots) if (r, s) not in big) + 15) // 8 * 8 + 8
    assert all(len(stream_of(r, s)) > slot_bytes for (r, s) in big)
    calls = []

    def reencode(step, payload, size):
        jf = stream_of(rank, step)
        payload[:len(jf)] = torch.frombuffer(bytearray(jf), dtype=torch.uint8)
        size[0] = len(jf)
        calls.append(step)

    g = ExactStreamGather(slot_bytes, slots, torch.device("cpu"), dst=0, depth=3, reencode=reencode)
    for s in range(slots):
        payload, size = g.record(s)
        jf = stream_of(rank, s)
        n = min(len(jf), payload.numel())                 # what an encoder with too small a buffer leaves: a cut stream and the would-be size
        payload[:n] = torch.frombuffer(bytearray(jf[:n]), dtype=torch.uint8)
        size[0] = len(jf)
        g.commit(s)
    g.drain()
    ret[f"calls{rank}"] = list(calls)
    if rank == 0:
        res = g.result(slots - 1)
        ret["ok"] = all(res[r][s] == stream_of(r, s) for r in range(world) for s in range(slots))
        ret["exchanges"] = g.exchanges
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_with_an_oversized_stream():
    """A record that does not fit its slot does not fail the buffer (a real configs[3] stream with one dense image): its owner
    encodes it again into a buffer of the exact size before the transfers are posted."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_oversize_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret.get("ok") is True and ret.get("exchanges") == 1
    assert ret.get("calls0") == [1] and ret.get("calls1") == [2]
