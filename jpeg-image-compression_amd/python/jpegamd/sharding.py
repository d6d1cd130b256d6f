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


class BatchedStreamGather:
    """Few, large collectives: `slots` consecutive results of a rank travel in ONE gather.

    Each rank owns `depth` staging buffers of `slots` fixed-size records.  A record is `slot_bytes` long: the
    bitstream at offset 0 and its byte count (int64) in the last 8 bytes, so the encoder writes both straight into
    the staging buffer (`record(step)` gives the two tensors to point it at) and one collective moves payloads and
    sizes together.  `commit(step)` starts the asynchronous gather of a buffer when its last record was produced;
    `reserve(step)` makes the calling stream wait for the gather that last read the buffer `step` is about to
    overwrite.  Sized for xGMI: at 8 ranks the root receives 7 x slots x slot_bytes per collective, each peer
    over its own link, instead of one small padded message per image.
    """

    def __init__(self, slot_bytes: int, slots: int, device, group=None, dst: int = 0, depth: int = 2):
        if slot_bytes % 8 or slot_bytes < 16:
            raise ValueError("slot_bytes must be a multiple of 8 (the size field is an aligned int64)")
        self.group, self.dst, self.slot_bytes, self.slots, self.depth = group, dst, int(slot_bytes), int(slots), int(depth)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.is_dst = self.rank == dst
        n = self.slot_bytes * self.slots
        self._stage = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(depth)]
        self._recv = [[torch.empty(n, dtype=torch.uint8, device=device) for _ in range(self.world)] if self.is_dst else None
                      for _ in range(depth)]
        self._work = [None] * depth
        self._extra = [dict() for _ in range(depth)]        # dst: {(rank, record): bytes} of the records collected by settle()
        self.collectives = 0
        self.device = device

    def _where(self, step: int) -> Tuple[int, int]:
        return (step // self.slots) % self.depth, step % self.slots

    def record(self, step: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(payload uint8[slot_bytes - 8], size int64[1]) views of the record `step` writes into."""
        b, k = self._where(step)
        rec = self._stage[b][k * self.slot_bytes:(k + 1) * self.slot_bytes]
        return rec[:self.slot_bytes - 8], rec[self.slot_bytes - 8:].view(torch.int64)

    def reserve(self, step: int) -> None:
        """Order the current stream after the collective that last read the buffer of `step` (no-op if none)."""
        b, _ = self._where(step)
        if self._work[b] is not None:
            self._work[b].wait()

    def commit(self, step: int, force: bool = False):
        """After record `step` was produced on the current stream: start the gather if the buffer is complete."""
        b, k = self._where(step)
        if k != self.slots - 1 and not force:
            return None
        self._work[b] = dist.gather(self._stage[b], self._recv[b], dst=self.dst, group=self.group, async_op=True)
        self.collectives += 1
        return self._work[b]

    def wait_all(self) -> None:
        for w in self._work:
            if w is not None:
                w.wait()

    def settle(self, step: int, reencode=None) -> int:
        """Collective (EVERY rank calls it, once the gather of the buffer holding `step` was waited for): a record whose
        stream did not fit its slot -- the encoder stores the would-be byte count even then, so its size field exceeds
        slot_bytes - 8 -- does not fail the buffer: its rank encodes that image again, `reencode(record_step, payload_tensor,
        size_tensor)`, into a buffer of the exact size, and ONE more gather collects all of them.  `result()` then returns the
        complete streams.  Returns the number of such records over all ranks; 0 (the usual case) costs one small all-reduce."""
        b, _ = self._where(step)
        base = (step // self.slots) * self.slots
        cap = self.slot_bytes - 8
        sizes = self._stage[b].view(self.slots, self.slot_bytes)[:, cap:].contiguous().view(torch.int64).cpu().reshape(-1)
        mine = [k for k in range(self.slots) if int(sizes[k]) > cap]
        agree = torch.tensor([len(mine), max([int(sizes[k]) for k in mine], default=0)], dtype=torch.int64, device=self.device)
        dist.all_reduce(agree, op=dist.ReduceOp.MAX, group=self.group)
        count, biggest = int(agree[0].item()), int(agree[1].item())
        self._extra[b] = dict()
        if count == 0:
            return 0
        if reencode is None:
            raise RuntimeError(f"{count} record(s) did not fit their slot of {cap} bytes and no reencode callback was given")
        rec_bytes = (biggest + 7) // 8 * 8 + 16                       # payload, its byte count, the record's index in the buffer
        ext = torch.zeros(count * rec_bytes, dtype=torch.uint8, device=self.device)
        tail = ext.view(count, rec_bytes)[:, rec_bytes - 16:]
        for j in range(count):
            idx = tail[j, 8:].view(torch.int64)
            idx.fill_(-1)                                             # (unused rows of a rank with fewer offenders)
        for j, k in enumerate(mine):
            row = ext[j * rec_bytes:(j + 1) * rec_bytes]
            reencode(base + k, row[:rec_bytes - 16], row[rec_bytes - 16:rec_bytes - 8].view(torch.int64))
            row[rec_bytes - 8:].view(torch.int64).fill_(k)
        recv = [torch.empty_like(ext) for _ in range(self.world)] if self.is_dst else None
        dist.gather(ext, recv, dst=self.dst, group=self.group)
        self.collectives += 1
        if self.is_dst:
            for r in range(self.world):
                buf = recv[r].cpu().view(count, rec_bytes)
                for j in range(count):
                    n = int(buf[j, rec_bytes - 16:rec_bytes - 8].view(torch.int64).item())
                    k = int(buf[j, rec_bytes - 8:].view(torch.int64).item())
                    if k < 0:
                        continue
                    if not (0 < n <= rec_bytes - 16):
                        raise RuntimeError(f"re-encoded record {k} of rank {r} holds {n} bytes (capacity {rec_bytes - 16})")
                    self._extra[b][(r, k)] = bytes(buf[j, :n].numpy())
        return sum(1 for _ in mine) if not self.is_dst else len(self._extra[b])

    def result(self, step: int) -> List[List[bytes]]:
        """On dst, after wait_all() and a device synchronise: per rank, the streams of the buffer holding `step`."""
        if not self.is_dst:
            return []
        b, _ = self._where(step)
        out = []
        for r in range(self.world):
            buf = self._recv[b][r].cpu()
            streams = []
            for k in range(self.slots):
                rec = buf[k * self.slot_bytes:(k + 1) * self.slot_bytes]
                n = int(rec[self.slot_bytes - 8:].view(torch.int64).item())
                if (r, k) in self._extra[b]:                          # it did not fit its slot: collected by settle()
                    streams.append(self._extra[b][(r, k)])
                    continue
                if n < 0 or n > self.slot_bytes - 8:
                    raise RuntimeError(f"gathered size {n} outside the record ({self.slot_bytes - 8}); call settle() on every rank")
                streams.append(bytes(rec[:n].numpy()))
            out.append(streams)
        return out


# ---------------------------------------------------------------------------------------------------------------
# One image over several GPUs: contiguous block-row shards, one exchange step.
# ---------------------------------------------------------------------------------------------------------------
class ShardedImageEncoder:
    """Encode ONE image with `world` ranks: rank r transforms and entropy-codes block rows shard_range(blocks_h, world, r)
    into unstuffed per-segment bit strings, packs them densely, the root collects every rank's pack (one size
    all-reduce + one padded gather per image) and runs the ordinary finalize over all segments -- bit offsets, 0xFF
    stuffing and the flush depend on the global byte phase, so they happen once, at the root.

    `encoder` is a jpegamd.Encoder sized for the image; buffers are allocated once for `max_words` packed words per
    rank (a rank's share of the worst case by default).  With `group=None` and torch.distributed not initialised the
    object runs as a single rank (useful for tests: `virtual_ranks` splits the work over several contexts on ONE GPU).
    """

    def __init__(self, encoder, width: int, height: int, device, group=None, dst: int = 0, max_words: Optional[int] = None):
        self.enc, self.w, self.h, self.device, self.group, self.dst = encoder, width, height, device, group, dst
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist_on else 1
        self.rank = dist.get_rank(group) if self.dist_on else 0
        self.blocks_w, self.blocks_h = (width + 7) // 8, (height + 7) // 8
        self.segs_per_row = (self.blocks_w + 255) // 256
        rows = -(-self.blocks_h // self.world)
        nseg = rows * self.segs_per_row
        if max_words is None:
            max_words = nseg * ((256 * 1721 + 31) // 32 + 1)          # every block at the worst-case bit count
        self.max_words, self.max_segs = int(max_words), nseg
        self.dense = torch.empty(self.max_words, dtype=torch.int32, device=device)
        self.meta_words = self._meta_words()
        self.meta = torch.zeros(nseg * self.meta_words, dtype=torch.int32, device=device)
        self.total = torch.zeros(1, dtype=torch.int32, device=device)
        if self.rank == dst and self.world > 1:
            self.recv_dense = [torch.empty(self.max_words, dtype=torch.int32, device=device) for _ in range(self.world)]
            self.recv_meta = [torch.empty(nseg * self.meta_words, dtype=torch.int32, device=device) for _ in range(self.world)]

    @staticmethod
    def _meta_words() -> int:
        import jpegamd
        return jpegamd.SEG_META_WORDS

    def rows_of(self, rank: int) -> Tuple[int, int]:
        return shard_range(self.blocks_h, self.world, rank)

    def encode(self, img, out: torch.Tensor, out_size: torch.Tensor, with_container: bool = True, stream: int = 0) -> None:
        """Collective: every rank calls it with the same image description (its `pixels` must cover the rank's rows and
        the block row above).  On `dst` the JFIF bytes land in `out`, their count in `out_size` (device int64).

        Everything -- kernels AND collectives -- is ordered on ONE stream (`stream`, a raw HIP stream handle, or torch's
        current stream): the collectives are issued with that stream current, so RCCL orders them behind the export
        kernels and the import / finalize kernels behind them.  Only the words actually used travel: one MAX all-reduce of
        the ranks' word totals sizes the gather (this is the one host synchronisation of the call) and doubles as the
        capacity check of every rank's pack.  With a gloo group (CPU tests, or several processes sharing one GPU) the
        packs are staged through host memory."""
        by0, by1 = self.rows_of(self.rank)
        ts = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
        with torch.cuda.stream(ts):
            raw = ts.cuda_stream
            self.enc.encode_rows_async(img, by0, by1, raw)
            self.enc.export_segments(img, by0, by1, self.dense.data_ptr(), self.max_words, self.meta.data_ptr(), self.total.data_ptr(), raw)
            if self.world == 1:
                self.enc.finalize_async(img, out.data_ptr(), out.numel(), out_size.data_ptr(), with_container, raw)
                return
            on_host = dist.get_backend(self.group) == "gloo"
            biggest = self.total.to(torch.int64).cpu() if on_host else self.total.to(torch.int64)
            dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=self.group)
            n = int(biggest.item())                                    # host waits here: the words of the fattest pack
            if n > self.max_words:
                raise RuntimeError(f"a rank's segment pack needs {n} words, the exchange buffers hold {self.max_words}")
            n = max(n, 1)
            root = self.rank == self.dst
            if on_host:
                d_h, m_h = self.dense[:n].cpu(), self.meta.cpu()
                rd = [torch.empty_like(d_h) for _ in range(self.world)] if root else None
                rm = [torch.empty_like(m_h) for _ in range(self.world)] if root else None
                dist.gather(d_h, rd, dst=self.dst, group=self.group)
                dist.gather(m_h, rm, dst=self.dst, group=self.group)
                if root:
                    for r in range(self.world):
                        self.recv_dense[r][:n].copy_(rd[r], non_blocking=False)
                        self.recv_meta[r].copy_(rm[r], non_blocking=False)
            else:
                dist.gather(self.dense[:n], [t[:n] for t in self.recv_dense] if root else None, dst=self.dst, group=self.group)
                dist.gather(self.meta, self.recv_meta if root else None, dst=self.dst, group=self.group)
            if root:
                for r in range(self.world):
                    if r == self.rank:
                        continue                                          # the root's own segments are already in place
                    b0, b1 = self.rows_of(r)
                    self.enc.import_segments(img, b0, b1, self.recv_dense[r].data_ptr(), self.recv_meta[r].data_ptr(), raw)
                self.enc.finalize_async(img, out.data_ptr(), out.numel(), out_size.data_ptr(), with_container, raw)


def encode_image_virtual_ranks(jpegamd, img, width: int, height: int, ranks: int, device, root_encoder=None):
    """Single-GPU rehearsal of the sharded path: `ranks` separate encoder contexts each code their block rows, every
    pack is imported into a root context, which finalizes.  Returns the JFIF bytes."""
    blocks_h = (height + 7) // 8
    spr = ((width + 7) // 8 + 255) // 256
    root = root_encoder or jpegamd.Encoder(width, height)
    packs = []
    for r in range(ranks):
        b0, b1 = shard_range(blocks_h, ranks, r)
        enc = jpegamd.Encoder(width, height)
        nseg = max(1, (b1 - b0) * spr)
        words = nseg * ((256 * 1721 + 31) // 32 + 1)
        dense = torch.empty(words, dtype=torch.int32, device=device)
        meta = torch.zeros(nseg * jpegamd.SEG_META_WORDS, dtype=torch.int32, device=device)
        total = torch.zeros(1, dtype=torch.int32, device=device)
        enc.encode_rows_async(img, b0, b1, 0)
        enc.export_segments(img, b0, b1, dense.data_ptr(), words, meta.data_ptr(), total.data_ptr(), 0)
        enc.finish()
        n = int(total.item())
        packs.append((b0, b1, dense[:max(n, 1)].clone(), meta))
    for b0, b1, dense, meta in packs:
        root.import_segments(img, b0, b1, dense.data_ptr(), meta.data_ptr(), 0)
    cap = 4096 + 2 * width * height
    out = torch.empty(cap, dtype=torch.uint8, device=device)
    size = torch.zeros(1, dtype=torch.int64, device=device)
    root.finalize_async(img, out.data_ptr(), cap, size.data_ptr(), True, 0)
    root.finish()
    return bytes(out[:int(size.item())].cpu().numpy())
