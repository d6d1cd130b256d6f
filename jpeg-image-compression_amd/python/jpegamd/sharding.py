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


class ExactStreamGather:
    """gather-v of finished bitstreams: every stream crosses the link at its EXACT size (rounded up to 8 bytes),
    `slots` consecutive results of a rank per exchange, and a rank's own results never move.

    Each rank owns `depth` staging buffers of `slots` records.  A record is `slot_bytes` long -- the bitstream at offset 0, its
    byte count (int64) in the last 8 bytes -- so the encoder writes both straight into the staging buffer (`record(step)`), with
    no copy between the encoder's output and the wire.  The exchange of a buffer is split so that the host never waits:

      commit(step)   the buffer is complete (call it on the stream that produced the records): its size table (`slots` int64) is
                     all-gathered over the ranks on the device and fetched into pinned host memory behind an event; then
                     flush() posts the transfers of the buffers committed EARLIER, whose tables have long arrived.
      flush()        sender: one isend per record, exactly the stream's bytes, all of a buffer in ONE batch (one grouped RCCL
                     launch); root: the matching irecvs into one dense area per rank, at running offsets.  A stream that outgrew
                     its slot (the encoder then leaves the would-be size and a cut stream) is encoded again by its owner,
                     `reencode(step, payload, size)`, into a side buffer of the exact size, and THAT is sent: no second exchange,
                     nothing for the root to know.
      reserve(step)  orders the calling stream behind the transfers that last read the buffer `step` is about to overwrite.
      drain()        flushes and waits for everything (host).
      result(step)   on dst, after drain(): per rank, the streams of the buffer that held `step`.

    A buffer is in flight from its first record until the transfers posted one commit later are through: depth >= 3.
    At 8 ranks the root posts 7 x slots receives per exchange, each peer's bytes over its own xGMI link."""

    def __init__(self, slot_bytes: int, slots: int, device, group=None, dst: int = 0, depth: int = 3, reencode=None):
        if slot_bytes % 8 or slot_bytes < 16:
            raise ValueError("slot_bytes must be a multiple of 8 (the size field is an aligned int64)")
        if depth < 3:
            raise ValueError("depth >= 3: a buffer's transfers are posted one commit after its own")
        self.group, self.dst, self.slot_bytes, self.slots, self.depth = group, dst, int(slot_bytes), int(slots), int(depth)
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist_on else 1
        self.rank = dist.get_rank(group) if self.dist_on else 0
        self.is_dst = self.rank == dst
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.reencode = reencode
        n = self.slot_bytes * self.slots
        self._stage = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(depth)]
        self._sizes = [torch.zeros(self.slots, dtype=torch.int64, device=device) for _ in range(depth)]
        self._all_sizes = [[torch.zeros(self.slots, dtype=torch.int64, device=device) for _ in range(self.world)] for _ in range(depth)]
        self._host = [torch.zeros(self.world, self.slots, dtype=torch.int64) for _ in range(depth)]
        if self.on_gpu:
            self._host = [h.pin_memory() for h in self._host]
        self._event = [None] * depth                         # sizes of the buffer's last commit are in _host behind this event
        self._base = [None] * depth                          # first step of the buffer's last commit
        self._pending: List[int] = []                        # committed buffers whose transfers are not posted yet (oldest first)
        self._works = [[] for _ in range(depth)]             # transfers (and the size all-gather) that still read / write the buffer
        self._side = [dict() for _ in range(depth)]          # record -> exact-size tensor of a stream that outgrew its slot
        self._recv = [[None] * self.world for _ in range(depth)]
        self._table = [None] * depth                         # dst: host copy of the sizes the posted receives were made from
        self._aux = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.exchanges = 0
        self.bytes_sent = 0                                  # by this rank, exact
        self.bytes_own = 0                                   # of this rank's own streams (sent, or -- the root -- kept)
        self.reencoded = 0

    def _where(self, step: int) -> Tuple[int, int]:
        return (step // self.slots) % self.depth, step % self.slots

    def _fetch_table(self, b: int) -> None:
        """(on the current stream) every rank's sizes of buffer b -> pinned host memory"""
        if self.world > 1:
            w = dist.all_gather(self._all_sizes[b], self._sizes[b], group=self.group, async_op=True)
            w.wait()                                         # (RCCL: the current -- side -- STREAM waits, not the host; gloo in the CPU tests: the host)
            table = torch.stack(self._all_sizes[b])
        else:
            table = self._sizes[b].view(1, self.slots)
        self._host[b].copy_(table, non_blocking=True)

    def record(self, step: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(payload uint8[slot_bytes - 8], size int64[1]) views of the record `step` writes into."""
        b, k = self._where(step)
        rec = self._stage[b][k * self.slot_bytes:(k + 1) * self.slot_bytes]
        return rec[:self.slot_bytes - 8], rec[self.slot_bytes - 8:].view(torch.int64)

    def reserve(self, step: int) -> None:
        """Order the current stream behind everything that still reads the buffer of `step` (no-op if nothing does)."""
        b, _ = self._where(step)
        if b in self._pending:                               # (only with too shallow a ring: post them now)
            self.flush()
        for w in self._works[b]:
            w.wait()
        self._works[b] = []

    def commit(self, step: int, force: bool = False) -> bool:
        """After record `step` was produced on the current stream: if the buffer is complete (or `force`), send its size table
        on its way and post the transfers of the buffers committed before.  -> whether the buffer was committed."""
        b, k = self._where(step)
        if k != self.slots - 1 and not force:
            return False
        cap = self.slot_bytes - 8
        # (ONE strided copy kernel: the last int64 of every record; a byte-wise .contiguous() of the same column is a copy per record)
        self._sizes[b].copy_(self._stage[b].view(torch.int64).view(self.slots, self.slot_bytes // 8)[:, -1])
        # The table travels on a side stream: the producing stream never waits for another rank here.
        if self.on_gpu:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self._aux):
                self._aux.wait_event(ready)
                self._fetch_table(b)
                self._event[b] = torch.cuda.Event()
                self._event[b].record()
        else:
            self._fetch_table(b)
        self._base[b] = (step // self.slots) * self.slots
        self._side[b] = dict()
        self.flush()                                         # the buffers committed EARLIER: their tables are on the host by now
        self._pending.append(b)
        return True

    def flush(self) -> None:
        """Post the transfers of every committed buffer whose size table has been requested (waits for the table: it is at least
        one buffer old when commit() calls this)."""
        while self._pending:
            b = self._pending.pop(0)
            if self._event[b] is not None:
                self._event[b].synchronize()
            sizes = self._host[b].clone()
            cap = self.slot_bytes - 8
            ops = []
            if True:                                         # every rank, the root too: its own oversized streams
                mine = sizes[self.rank]
                for k in range(self.slots):
                    n = int(mine[k])
                    self.bytes_own += max(n, 0)
                    if n > cap:                              # the stream outgrew its slot: once more, at the exact size
                        if self.reencode is None:
                            raise RuntimeError(f"record {k} holds {n} bytes, its slot {cap}, and no reencode callback was given")
                        side = torch.zeros((n + 7) // 8 * 8 + 8, dtype=torch.uint8, device=self.device)
                        self.reencode(self._base[b] + k, side[:-8], side[-8:].view(torch.int64))
                        self._side[b][k] = side
                        self.reencoded += 1
            if self.world > 1:
                if not self.is_dst:
                    for k in range(self.slots):
                        n = int(sizes[self.rank][k])
                        if n <= 0:
                            continue
                        n8 = (n + 7) // 8 * 8
                        src = self._side[b][k][:n8] if k in self._side[b] else self._stage[b][k * self.slot_bytes:k * self.slot_bytes + n8]
                        ops.append(dist.P2POp(dist.isend, src, self.dst, self.group))
                        self.bytes_sent += n8
                else:
                    for r in range(self.world):
                        if r == self.rank:
                            continue
                        total = sum((int(x) + 7) // 8 * 8 for x in sizes[r] if int(x) > 0)
                        if self._recv[b][r] is None or self._recv[b][r].numel() < total:
                            self._recv[b][r] = torch.empty(max(total, self.slots * self.slot_bytes), dtype=torch.uint8, device=self.device)
                        off = 0
                        for k in range(self.slots):
                            n = int(sizes[r][k])
                            if n <= 0:
                                continue
                            n8 = (n + 7) // 8 * 8
                            ops.append(dist.P2POp(dist.irecv, self._recv[b][r][off:off + n8], r, self.group))
                            off += n8
                if ops:
                    self._works[b].extend(dist.batch_isend_irecv(ops))
                    self.exchanges += 1
            self._table[b] = sizes

    def drain(self) -> None:
        """Host: post what is left and wait for every transfer."""
        self.flush()
        for b in range(self.depth):
            for w in self._works[b]:
                w.wait()
            self._works[b] = []
        if self.on_gpu:
            torch.cuda.synchronize()

    def result(self, step: int) -> List[List[bytes]]:
        """On dst, after drain(): per rank, the streams of the buffer that held `step`."""
        if not self.is_dst:
            return []
        b, _ = self._where(step)
        sizes = self._table[b]
        if sizes is None:
            raise RuntimeError("result() before the buffer was committed and drained")
        out = []
        for r in range(self.world):
            streams = []
            if r == self.rank:                               # the root's own streams never moved
                stage = self._stage[b].cpu()
                for k in range(self.slots):
                    n = int(sizes[r][k])
                    src = self._side[b][k].cpu() if k in self._side[b] else stage[k * self.slot_bytes:(k + 1) * self.slot_bytes]
                    streams.append(bytes(src[:n].numpy()) if n > 0 else b"")
            else:
                buf = self._recv[b][r].cpu() if self._recv[b][r] is not None else None
                off = 0
                for k in range(self.slots):
                    n = int(sizes[r][k])
                    if n <= 0:
                        streams.append(b"")
                        continue
                    streams.append(bytes(buf[off:off + n].numpy()))
                    off += (n + 7) // 8 * 8
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
