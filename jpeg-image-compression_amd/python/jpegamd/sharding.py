"""Multi-GPU sharding of the encode path: one process per GPU, independent images per rank.

The reference has no distributed layer; the path shards naturally because
images (and, within an image, block-row tiles) are independent.  The only exchange step is
the collection of the finished per-image bitstreams at one rank: a padded gather over
torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in CPU tests).
Nothing here touches pixels; it only moves finished JFIF bytes.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of `n_units` independent units owned by `rank`."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_units, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def owner_of(unit: int, n_units: int, world: int) -> int:
    for r in range(world):
        b, e = shard_range(n_units, world, r)
        if b <= unit < e:
            return r
    raise ValueError("unit out of range")


class StreamGather:
    """Padded gather of one fixed-capacity payload + its byte count per rank to `dst`.

    Buffers are allocated once.  `start()` enqueues the two collectives asynchronously (they
    are ordered after whatever produced `payload`/`size` on the current stream) and returns
    a handle; the producer may overwrite `payload` only after `handle.wait()`.
    """

    def __init__(self, capacity: int, device, group=None, dst: int = 0, depth: int = 2):
        self.group, self.dst, self.capacity = group, dst, int(capacity)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.is_dst = self.rank == dst
        self.depth = depth
        self._recv_payload: List[Optional[List[torch.Tensor]]] = []
        self._recv_sizes: List[Optional[List[torch.Tensor]]] = []
        for _ in range(depth):
            if self.is_dst:
                self._recv_payload.append([torch.empty(self.capacity, dtype=torch.uint8, device=device)
                                           for _ in range(self.world)])
                self._recv_sizes.append([torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.world)])
            else:
                self._recv_payload.append(None)
                self._recv_sizes.append(None)

    class Handle:
        def __init__(self, works, slot):
            self.works, self.slot = works, slot

        def wait(self):
            for w in self.works:
                w.wait()

    def start(self, payload: torch.Tensor, size: torch.Tensor, slot: int) -> "StreamGather.Handle":
        if payload.numel() != self.capacity or payload.dtype != torch.uint8:
            raise ValueError("payload must be a uint8 tensor of the declared capacity")
        slot %= self.depth
        w1 = dist.gather(size, self._recv_sizes[slot], dst=self.dst, group=self.group, async_op=True)
        w2 = dist.gather(payload, self._recv_payload[slot], dst=self.dst, group=self.group, async_op=True)
        return StreamGather.Handle([w1, w2], slot)

    def result(self, slot: int) -> Sequence[bytes]:
        """On dst, after the handle was waited for and the device synchronised: the streams in rank order."""
        if not self.is_dst:
            return []
        slot %= self.depth
        out = []
        for p, s in zip(self._recv_payload[slot], self._recv_sizes[slot]):
            n = int(s.item())
            if n < 0 or n > self.capacity:
                raise RuntimeError(f"gathered size {n} outside capacity {self.capacity}")
            out.append(bytes(p[:n].cpu().numpy()))
        return out
