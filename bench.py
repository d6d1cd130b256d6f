#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the BMP -> grayscale baseline-JPEG encode on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU.
Rank 0 prints ONE JSON line.

Workloads (`--workload`, default `auto`):
  image8192  BASELINE.json configs[2], the config the metric and the roofline target are quoted on: 8192x8192 synthetic
             24-bit BMPs.  By default a step codes EIGHT of them (distinct pictures, 16 in rotation per rank: 3.2 GB > the
             256 MiB Infinity Cache) through ONE launch of each kernel (`--images-per-launch`, jpegamd_encode_batch_async):
             the launch, prologue and drain costs of the three kernels are paid once per eight images.  `--images-per-launch
             1` is the single-image step of round 1 (3 rotating inputs); its kernel durations are also measured in every
             default run and reported as roofline.one_image_per_launch.
             This is what `auto` selects at EVERY N: the driver derives the scaling efficiency from the per-N values, so
             the per-rank work has to be the same at N = 1 and N = 8 (weak scaling).
  batch4096  BASELINE.json configs[3]: the batch of 64 independent 4096x4096 images, 64 / N per rank per step (N = 8: 8 each,
             one launch; N = 1: all 64, two launches of 32 = JPEGAMD_MAX_BATCH, as much work per launch as eight 8192^2),
             distinct seeds (tests/golden/batch4096.json holds the compiled reference's answers for all 64); every finished
             bitstream collected at rank 0 by RCCL (N = 1: `--force-gather`, a one-rank group).
Reference quantisation table (Q=50) unless --quality says otherwise; pixel rows resident in HBM; output = complete
JFIF file bytes in HBM.  A "step" is one pass of the hot path (k_tile_encode -> k_segment_merge -> k_finalize; `--pipeline stitch`:
k_tile_encode -> k_stitch, the single-pass kernel the library itself takes for 16384^2-class pictures) over the step's images, on one HIP stream by default (`--streams`: k_tile_encode is a persistent kernel that fills the GPU; launches on
several streams queue behind each other's workgroups and were measured slower); every step is a complete encode.  With N > 1 every rank
encodes its own images (weak scaling, no data-path collective inside the encode) and the finished bitstreams are
collected at rank 0 by a gather-v over RCCL -- per `--gather-every` images one exchange of exact-size transfers
(jpegamd.sharding.ExactStreamGather: size tables one buffer ahead, one grouped send per rank), overlapped with the following steps.

Extra objects on the JSON line:
  "roofline"      HBM roofline of the encode as SURVEY.md 8d defines it: algorithmic bytes of ONE launch (BMP rows read + JFIF
                  bytes written, per image x images_per_launch) / SUM of the kernels' durations / 8 TB/s.  Durations are
                  HIP-event timed inside this run through the C-ABI's event ring -- every kernel launched with its own begin / end
                  events on the stream it runs on (a kernel's own duration, as a kernel trace shows it) -- in SEPARATE single-stream
                  passes behind the timed region, whose own launches carry no events: once right behind the region (`sustained`) and
                  once 250 ms later (the headline figures; the state a kernel trace of `bench.py --streams 1` sees).
                  `dominant_frac` is the same bytes over k_tile_encode alone, `hbm_read_frac` the read bytes alone over the sum,
                  `per_image_us` the sum divided by images_per_launch, `one_image_per_launch` the literal configs[2] launch shape.
  "configs3"      N > 1 (or --force-gather): a short leg of BASELINE configs[3] behind the main one -- 64 / N independent 4096^2 images
                  per rank and step (launches of up to 32), every bitstream gathered at rank 0.
  "cpu_baseline"  the compiled reference natural_c (oracle/_ref), single thread, on a bounded sample: with its own
                  flags (no -O) and with -O2; rank 0, N == 1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
ROTATE = int(os.environ.get("JPEGAMD_BENCH_ROTATE", "3"))   # image8192: rotating inputs per rank (1 = experiment: the input stays in the 256 MB Infinity Cache)
BATCH_TOTAL = 64               # batch4096: BASELINE configs[3]'s batch, split over the ranks


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", choices=["auto", "image8192", "batch4096"], default="auto")
    ap.add_argument("--width", type=int, default=0, help="override the workload's image width")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--kind", type=int, default=0, help="synthetic content: 0 photo-like, 1 noise, 2 flat, 3 gradient")
    ap.add_argument("--quality", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-image-pass", action="store_true",
                    help="skip the one-image-per-launch comparison pass (kernel traces of the batched default: every launch then codes the same number of images)")
    ap.add_argument("--images-per-launch", type=int, default=0, choices=[0, 1, 2, 4, 8, 16, 32],
                    help="images coded by ONE launch of each kernel (jpegamd_encode_batch_async); 0 = 8 for image8192 (8 images per step), "
                         "min(32, images per rank) for batch4096")
    ap.add_argument("--pipeline", choices=["auto", "pair", "stitch"], default="auto",
                    help="kernels behind k_tile_encode (jpegamd_encoder_set_pipeline): k_segment_merge + k_finalize, or the single-pass k_stitch")
    ap.add_argument("--roofline-idle-ms", type=float, default=250.0, help="idle time in front of the second single-stream roofline pass")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (one encoder context each) the launches alternate over.  One by default: k_tile_encode is a "
                         "persistent kernel that fills the GPU, so launches on several streams only queue behind each other's "
                         "workgroups (measured: 4 streams 0.544, 2 streams 0.567, 1 stream 0.534 ms per step of eight images)")
    ap.add_argument("--burn-in-ms", type=float, default=40.0,
                    help="untimed steps in front of the W warm-up steps, for this many milliseconds: the first ~12 ms of sustained load "
                         "after an idle period run 10-30 %% slow on MI355X (power-management transient, profiles/r03_load_onset.txt), and "
                         "W = 5 steps are 2 ms.  0 = none")
    ap.add_argument("--force-gather", action="store_true", help="run the N > 1 gather path with a one-rank group (rehearsal on one GPU)")
    ap.add_argument("--gather-every", type=int, default=32,
                    help="N > 1: images per rank carried by one gather to rank 0 (few, large collectives)")
    ap.add_argument("--cpu-sample-rows", type=int, default=8192,
                    help="rows of the first image the CPU baseline encodes (bounded sample)")
    ap.add_argument("--configs3-steps", type=int, default=20, help="steps of the configs[3] leg behind the main workload (N > 1; 0 = none)")
    return ap.parse_args()


def make_inputs(args, rank, torch, jpegamd, w, h, seeds):
    """-> (list of device pixel tensors, stride, host bytes of the first image's BMP file on rank 0)."""
    size = jpegamd.synth_bmp_into(0, 0, w, h)                     # query
    host = torch.empty(size, dtype=torch.uint8).pin_memory()
    stride = (3 * w + 3) & ~3
    dev, first = [], None
    for i, seed in enumerate(seeds):
        got = jpegamd.synth_bmp_into(host.data_ptr(), size, w, h, seed, args.kind, 0)
        assert got == size
        if i == 0 and rank == 0:
            first = bytes(host.numpy())                            # keep a copy for the CPU baseline
        dev.append(host[54:54 + stride * h].cuda(non_blocking=False).contiguous())
    return dev, stride, first


def cpu_info():
    model, cores = "unknown", os.cpu_count() or 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, cores


def cpu_baseline(first_bmp: bytes, args, w, h):
    """Time the reference's own code (oracle/_ref) on a bounded sample: its Makefile flags (no -O), and -O2."""
    import numpy as np
    rows = min(h, max(8, args.cpu_sample_rows // 8 * 8))
    stride = (3 * w + 3) & ~3
    px = np.frombuffer(first_bmp, np.uint8, offset=54).reshape(h, stride)
    top = px[h - rows:, :]                                        # bottom-up file: last `rows` stored rows = top of image
    from oracle import oracle
    model, cores = cpu_info()
    sample = f"top {rows} rows of the first {w}x{h} image ({w * rows / 1e6:.1f} Mpx), single thread"

    class BMPImage(ctypes.Structure):
        _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("data", ctypes.c_void_p)]

    def time_ref(lib_path):
        rgb = np.ascontiguousarray(top[::-1, :3 * w].reshape(rows, w, 3)[:, :, ::-1])   # RGB top-down, tight
        lib = ctypes.CDLL(str(lib_path))
        lib.saveJPEGGrayscale.restype = ctypes.c_bool
        lib.saveJPEGGrayscale.argtypes = [ctypes.c_char_p, ctypes.POINTER(BMPImage)]
        img = BMPImage(w, rows, rgb.ctypes.data)
        out = f"/dev/shm/jpegamd_cpu_baseline_{os.getpid()}.jpg"
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)                                       # the reference prints progress lines
        try:
            t0 = time.perf_counter()
            ok = lib.saveJPEGGrayscale(out.encode(), ctypes.byref(img))
            dt = time.perf_counter() - t0
        finally:
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        try:
            os.unlink(out)
        except OSError:
            pass
        return dt if ok else None

    base = {"unit": "Mpixels/s", "cores": 1, "cpu_model": model, "host_cores": cores}
    if oracle.REF_LIB.exists():
        dt = time_ref(oracle.REF_LIB)
        if dt:
            res = dict(base, value=round(w * rows / dt / 1e6, 3), kind="reference", seconds=round(dt, 2),
                       sample=sample + "; natural_c built with its own flags (natural_c/Makefile:4: -g, no -O), in-memory "
                                       "BMPImage in, JPEG written to /dev/shm")
            if oracle.REF_LIB_O2.exists():
                dt2 = time_ref(oracle.REF_LIB_O2)
                if dt2:
                    res["O2"] = {"value": round(w * rows / dt2 / 1e6, 3), "unit": "Mpixels/s", "seconds": round(dt2, 2),
                                 "flags": "the same sources with -O2 added (identical bytes)"}
            return res
    # port: this repo's restatement (-O2)
    hdr = bytearray(first_bmp[:54])
    hdr[22:26] = int(rows).to_bytes(4, "little", signed=True)
    bmp = bytes(hdr) + top.tobytes()
    t0 = time.perf_counter()
    oracle.encode_bmp(bmp, args.quality)
    dt = time.perf_counter() - t0
    return dict(base, value=round(w * rows / dt / 1e6, 3), kind="port", seconds=round(dt, 2),
                sample=sample + "; oracle/natural_oracle.c at -O2")


def main():
    args = parse_args()
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner on
    # stdout when its first communicator is created), so the process's fd 1 is pointed at stderr for the whole run
    # and the JSON line goes out through a private duplicate of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import jpegamd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the encode path has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                   # rehearsal of the N > 1 exchange path with a one-rank RCCL group
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def measure(workload, K, W, images_per_launch, primary):
        """One workload through warm-up, timed region and the per-kernel passes -> (JSON line fields on rank 0, parity_ok, first BMP)."""
        B = images_per_launch
        if workload == "image8192":
            w, h, ips = args.width or 8192, args.height or 8192, B
            nrot = ROTATE if B == 1 else max(ROTATE, 2 * B)          # a launch reads B distinct pictures, two launches never the same ones
            seeds = [1000 + rank * nrot + i for i in range(nrot)]
        else:
            w, h, ips = args.width or 4096, args.height or 4096, max(B, BATCH_TOTAL // world)
            seeds = [2000 + (rank * ips + i) % BATCH_TOTAL for i in range(ips)]
        inputs, stride, first_bmp = make_inputs(args, rank, torch, jpegamd, w, h, seeds)
        nimg = len(inputs)
        nstreams = max(1, args.streams)
        encs = [jpegamd.Encoder(w, B * ((h + 7) // 8 * 8)) for _ in range(nstreams)]
        if args.pipeline != "auto":
            for e in encs:
                e.set_pipeline({"pair": jpegamd.PIPELINE_PAIR, "stitch": jpegamd.PIPELINE_STITCH}[args.pipeline])
        cap = 4096 + w * h // 2 if args.kind != 1 and args.quality <= 75 else 4096 + 2 * w * h     # >10x the typical photo-like output
        nbuf = max(2 * nstreams * B, nimg, B + 1)
        outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        sizes = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(nbuf)]
        imgs = [jpegamd.Encoder.image(t.data_ptr(), w, h, stride, True, jpegamd.ORDER_BGR, args.quality) for t in inputs]
        tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]

        # N > 1: the finished bitstreams are collected at rank 0, `gather_every` images per exchange, every stream at its exact size
        # (each peer's bytes cross its own xGMI link to the root).  The encoder writes into staging records whose size is fixed
        # before the timed region from the sizes the inputs actually produce (+5 %), agreed over all ranks; a stream that outgrows
        # its record is encoded again by its owner at the exact size (ExactStreamGather).
        gather = None
        G = max(1, args.gather_every)
        if dist is not None:
            from jpegamd.sharding import ExactStreamGather
            biggest = 0
            for im in imgs:
                encs[0].encode_async(im, outs[0].data_ptr(), cap, sizes[0].data_ptr(), True, tstreams[0].cuda_stream)
                biggest = max(biggest, int(encs[0].finish().jfif_bytes))
            t = torch.tensor([biggest], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            slot_bytes = ((int(t.item()) * 21 // 20 + 4096 + 255) // 256) * 256 + 8
            gather = ExactStreamGather(slot_bytes, G, torch.device("cuda", local_rank), dst=0, depth=3,
                                       reencode=lambda n, payload, size: reencode(n, payload, size))
            rec_ptrs = {}
            for st_i in range(3 * G):
                pl, sz = gather.record(st_i)
                rec_ptrs[st_i] = (pl.data_ptr(), pl.numel(), sz.data_ptr(), sz)
        last_image = [-1]

        def encode_launch(m):
            """m-th launch of the run: images n = m * B .. m * B + B - 1 (image n of the run is picture n % nimg)."""
            si = m % nstreams                                         # launches alternate over the streams / contexts
            with torch.cuda.stream(tstreams[si]):
                ns = list(range(m * B, m * B + B))
                if gather is None:
                    optrs = [(outs[n % nbuf].data_ptr(), cap, sizes[n % nbuf].data_ptr()) for n in ns]
                else:
                    for n in ns:
                        if n % G < nstreams * B:                      # a stream's first write into this buffer: behind the
                            gather.reserve(n)                         # collective that last read it
                    optrs = [rec_ptrs[n % (3 * G)][:3] for n in ns]
                if B == 1:
                    encs[si].encode_async(imgs[ns[0] % nimg], optrs[0][0], optrs[0][1], optrs[0][2], True, tstreams[si].cuda_stream)
                else:
                    encs[si].encode_batch_async([imgs[n % nimg] for n in ns], [o[0] for o in optrs], min(o[1] for o in optrs),
                                                [o[2] for o in optrs], True, tstreams[si].cuda_stream)
                if gather is not None and ns[-1] % G == G - 1:
                    commit(ns[-1], False)
            last_image[0] = m * B + B - 1

        def step(i):
            for j in range(ips // B):
                encode_launch(i * (ips // B) + j)

        def commit(n, force):
            cur = torch.cuda.current_stream()
            for sj in tstreams:                                       # the buffer's records were produced on all streams
                if sj != cur:
                    cur.wait_event(sj.record_event())
            gather.commit(n, force=force)

        def drain():
            if gather is not None and last_image[0] >= 0:
                if last_image[0] % G != G - 1:
                    with torch.cuda.stream(tstreams[0]):
                        commit(last_image[0], True)
                gather.drain()
            torch.cuda.synchronize()

        def reencode(n, payload, size):
            encs[0].encode_async(imgs[n % nimg], payload.data_ptr(), payload.numel(), size.data_ptr(), True, tstreams[0].cuda_stream)
            try:
                encs[0].finish()
            except jpegamd.JpegAmdError as err:                       # (the capacity status is sticky: the first encode of this image may have set it)
                if err.code != -8:
                    raise
            if not 0 < int(size.item()) <= payload.numel():
                raise RuntimeError(f"bench.py: image {n} encoded again into {payload.numel()} bytes reports {int(size.item())}")

        def check_capacity(where):
            """The capacity status is sticky on the device: finish() reports an overflow of ANY encode since the last finish."""
            for e in encs:
                try:
                    e.finish()
                except jpegamd.JpegAmdError as err:
                    if err.code == -1:                                # nothing pending on this context
                        continue
                    if gather is not None and err.code == -8:         # a stream outgrew its record: its owner encodes it again at the exact size (counted below)
                        continue
                    raise RuntimeError(f"bench.py: an encode in the {where} did not fit its output buffer ({err})")

        # burn-in (untimed, part of the set-up like the input generation): steps until the device has been under sustained load
        # for --burn-in-ms; the W warm-up steps follow without a pause
        burn_steps = 0
        if args.burn_in_ms > 0:
            t_b = time.perf_counter()
            while (time.perf_counter() - t_b) * 1e3 < args.burn_in_ms:
                for _ in range(8):
                    step(burn_steps)
                    burn_steps += 1
                tstreams[0].synchronize()
        W0 = burn_steps                                              # steps are numbered on from the burn-in (buffers rotate with the step number)
        for i in range(W):
            step(W0 + i)
        drain()
        if W or burn_steps:
            check_capacity("warm-up")

        n_timed = K * ips
        n_launches = n_timed // B
        own0 = gather.bytes_own if gather is not None else 0
        for e in encs:
            e.set_profiling(0)                                        # the timed region's launches carry no events (plain hipLaunchKernelGGL)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            step(W0 + W + i)
        t_issued = time.perf_counter()                               # host-side cost of enqueueing the K steps
        drain()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        elapsed = t1 - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        last = (W0 + W + K) * ips - 1
        last_ctx = (last // B) % nstreams
        def finish_timed(e):
            try:
                return e.finish()                                     # raises if ANY timed encode overflowed on that context
            except jpegamd.JpegAmdError as err:
                if gather is not None and err.code in (-1, -8):       # (oversized streams went out re-encoded: gather.reencoded)
                    return None
                raise
        st = finish_timed(encs[last_ctx])
        for si, e in enumerate(encs):
            if si != last_ctx and n_launches > si:
                finish_timed(e)
        if gather is not None:                                        # every record of the last buffers carries a plausible size
            for k in range(min(3 * G, n_timed)):
                n_bytes = int(rec_ptrs[k][3].item())
                if n_bytes <= 0:
                    raise RuntimeError(f"bench.py: gather record {k} holds {n_bytes} bytes (capacity {rec_ptrs[k][1]})")

        # Kernel durations for the roofline.  The library launches every kernel with its own begin / end events
        # (hipExtLaunchKernelGGL), so a duration is the kernel's own, as in a kernel trace -- but with several streams the
        # kernels of different images share the GPU in the timed region and stretch.  A short single-stream pass over the
        # same inputs gives each kernel alone (this is what rocprofv3 --stats sees for `bench.py --streams 1`).
        # Two such passes: one right behind the timed region (the GPU still in the power / clock state of sustained load: the
        # k_tile_encode runs a few percent shorter there, the other two kernels a little shorter) and one after a short
        # idle.  `roofline` is built from the second, the state a kernel trace of `bench.py --streams 1` sees (its launch gaps
        # keep the GPU a quarter idle); the first is reported beside it as `sustained`.
        sustained, single_ns = None, None
        P, SKIP = (70, 10) if primary else (30, 6)                                          # the first SKIP launches of a pass (clock ramp after the idle) are not averaged

        def single_stream_pass(nb):
            """P launches of nb images each on stream 0 -> mean own durations of the three kernels (ns per launch)."""
            encs[0].set_profiling(P)
            torch.cuda.synchronize()
            for i in range(P):
                if nb == 1:
                    encs[0].encode_async(imgs[i % nimg], outs[0].data_ptr(), cap, sizes[0].data_ptr(), True, tstreams[0].cuda_stream)
                else:
                    ns = [(i * nb + j) % nimg for j in range(nb)]
                    encs[0].encode_batch_async([imgs[n] for n in ns], [outs[j].data_ptr() for j in range(nb)], cap,
                                               [sizes[j].data_ptr() for j in range(nb)], True, tstreams[0].cuda_stream)
            torch.cuda.synchronize()
            encs[0].finish()
            prof = [encs[0].profile(s) for s in range(SKIP, P)]
            return tuple(sum(getattr(p, f) for p in prof) / len(prof) for f in ("ns_transform", "ns_entropy", "ns_pack", "ns_total"))

        if primary:
            h_tr, h_en, h_pk, _ = single_stream_pass(B)
            sustained = {"transform_us": round(h_tr / 1e3, 2), "merge_us": round(h_en / 1e3, 2), "finalize_us": round(h_pk / 1e3, 2),
                         "sum_kernels_us": round((h_tr + h_en + h_pk) / 1e3, 2), "measured": f"single-stream pass of {P - SKIP} launches right behind the timed region"}
            time.sleep(args.roofline_idle_ms / 1e3)
        ns_tr, ns_en, ns_pk, ns_tot = single_stream_pass(B)
        roof_note = (f"single-stream pass of {P - SKIP} launches with per-kernel events, {args.roofline_idle_ms:.0f} ms after the timed region "
                     "(the timed region itself launches without events)")
        if primary and B > 1 and not args.no_one_image_pass:       # the same three kernels over ONE image per launch: the literal configs[2] launch shape
            single_ns = single_stream_pass(1)
        ns_sum = ns_tr + ns_en + ns_pk                                # sum of the kernels' own durations (no launch gaps)

        # parity spot check against the committed natural_c golden: the last image of the run (B = 1), or one more batched launch
        # whose first picture has a golden (goldens exist for the first three seeds of rank 0) and whose other outputs must equal
        # single-image encodes of the same pictures
        batch_self_ok = True
        if B == 1:
            pick = last % nimg
            encs[0].encode_async(imgs[pick], outs[1].data_ptr(), cap, sizes[1].data_ptr(), True, tstreams[0].cuda_stream)
            encs[0].finish()
            out_bytes = bytes(outs[1][: int(sizes[1].item())].cpu().numpy())
        else:
            pick = 0
            def batched(order):
                encs[0].encode_batch_async([imgs[j % nimg] for j in order], [outs[j].data_ptr() for j in range(B)], cap,
                                           [sizes[j].data_ptr() for j in range(B)], True, tstreams[0].cuda_stream)
                encs[0].finish()
                return [bytes(outs[j][: int(sizes[j].item())].cpu().numpy()) for j in range(B)]
            batch_out = batched(list(range(B)))
            rotated = batched([(j + 1) % B for j in range(B)])           # the same pictures at other positions of the launch
            out_bytes = batch_out[0]
            batch_self_ok = all(rotated[j] == batch_out[(j + 1) % B] for j in range(B))
        parity, parity_ok = "unchecked", True
        gold_file = ROOT / "tests" / "golden" / ("large.json" if workload == "image8192" else "batch4096.json")
        if gold_file.exists() and rank == 0:
            key = f"{w}x{h}_seed{seeds[pick]}_kind{args.kind}_q{args.quality}"
            ent = json.loads(gold_file.read_text()).get(key)
            if ent:
                parity_ok = ent["sha256"] == hashlib.sha256(out_bytes).hexdigest() and ent["size"] == len(out_bytes)
                parity = "sha256 == natural_c golden" if parity_ok else "MISMATCH vs natural_c golden"
                if B > 1:
                    gold = json.loads(gold_file.read_text())
                    others = [gold.get(f"{w}x{h}_seed{seeds[j % nimg]}_kind{args.kind}_q{args.quality}") for j in range(1, B)]
                    n_gold = 1 + sum(1 for j, g in enumerate(others, 1) if g)
                    gold_ok = all(g is None or (g["sha256"] == hashlib.sha256(batch_out[j]).hexdigest()) for j, g in enumerate(others, 1))
                    parity_ok = parity_ok and gold_ok and batch_self_ok
                    parity = (f"sha256 == natural_c golden ({n_gold} of the {B} pictures of a batched launch have one; all {B} come out the same "
                              "at other positions of the launch)") if parity_ok else "MISMATCH (batched launch vs natural_c golden / vs the same pictures at other positions)"

        if gather is not None and rank == 0:
            per_rank = gather.result(last)
            used = last % G + 1
            if len(per_rank) != world or any(s[:2] != b"\xff\xd8" or s[-2:] != b"\xff\xd9" for r in per_rank for s in r[:used]):
                raise RuntimeError("gathered streams are not complete JFIF files")
            if workload == "batch4096" and gold_file.exists() and args.kind == 0 and args.quality == 50:
                gold = json.loads(gold_file.read_text())               # every gathered stream of the last buffer against the reference
                for r in range(world):
                    for k in range(used):
                        n = last - (used - 1) + k
                        ent = gold.get(f"{w}x{h}_seed{2000 + (r * ips + n % nimg) % BATCH_TOTAL}_kind0_q50")
                        if ent and hashlib.sha256(per_rank[r][k]).hexdigest() != ent["sha256"]:
                            parity, parity_ok = f"MISMATCH vs natural_c golden (rank {r}, image {n})", False

        if rank != 0:
            return None, parity_ok, None

        mpx = w * h / 1e6
        read_bytes = stride * h
        algo_bytes = read_bytes + len(out_bytes)
        launch_bytes = algo_bytes * B                                 # algorithmic bytes of ONE launch of each kernel (B images; the last image's size stands for all)
        traffic, traffic_note = None, None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():
            try:
                ent = json.loads(tf.read_text()).get(f"{w}x{h}_kind{args.kind}", {})
                if ent.get("pipeline_bytes_per_image"):
                    traffic = ent["pipeline_bytes_per_image"] * B
                    traffic_note = ("not measured in this run: " + ent.get("how", "profiles/hbm_traffic.json")
                                    + f"; per image x {B} images per launch")
            except Exception:
                traffic = None
        one = None
        if single_ns:
            one = {"transform_us": round(single_ns[0] / 1e3, 2), "merge_us": round(single_ns[1] / 1e3, 2), "finalize_us": round(single_ns[2] / 1e3, 2),
                   "sum_kernels_us": round(sum(single_ns[:3]) / 1e3, 2), "achieved": round(algo_bytes / sum(single_ns[:3]), 1),
                   "frac": round(algo_bytes / sum(single_ns[:3]) / HBM_PEAK_GBS, 4),
                   "measured": f"single-stream pass of {P - SKIP} single-image launches (the literal BASELINE configs[2] launch shape) behind the batched one"}
        if workload == "batch4096":
            what = f" (BASELINE configs[3]: the batch of 64, {ips} per rank)"
        elif B == 1:
            what = " (BASELINE configs[2])"
        else:
            what = (f" ({B} images per launch of each kernel: an extension of BASELINE configs[2], whose literal shape -- one image per launch -- "
                    "is roofline.one_image_per_launch)")
        line = {
            "metric": "Mpixels/s encode (BMP -> grayscale baseline JPEG, bit-exact vs natural_c)",
            "value": round(world * ips * mpx * K / elapsed, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "burn_in": {"ms": args.burn_in_ms, "steps": burn_steps},
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 in, f32 DCT, int16 coefficients",
            "data": "synthetic",
            "config": {"workload": f"{w}x{h} synthetic RGB BMP (kind {args.kind}), Q={args.quality}, {ips} image(s)/step/rank, "
                                   f"{nimg} distinct inputs/rank" + what,
                       "images_per_step": world * ips,
                       "images_per_launch": B,
                       "streams_per_rank": nstreams,
                       "parallelism": f"dp{world} (independent images per rank"
                                      + (", async RCCL gather of bitstreams to rank 0)" if dist is not None else ")")},
            "roofline": {"bound": "hbm", "kernel": ("k_tile_encode + k_segment_merge + k_finalize" if ns_en else "k_tile_encode + k_stitch (its duration under finalize_us)") + " (sum of durations, SURVEY.md 8d)",
                         "achieved": round(launch_bytes / ns_sum, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(launch_bytes / ns_sum / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes": launch_bytes, "images_per_launch": B,
                         "hbm_read_frac": round(read_bytes * B / ns_sum / HBM_PEAK_GBS, 4),
                         "dominant_kernel": "k_tile_encode", "dominant_frac": round(launch_bytes / ns_tr / HBM_PEAK_GBS, 4),
                         "per_image_us": round(ns_sum / B / 1e3, 2),
                         "kernel_us": round(ns_tr / 1e3, 2), "merge_us": round(ns_en / 1e3, 2), "finalize_us": round(ns_pk / 1e3, 2),
                         "sum_kernels_us": round(ns_sum / 1e3, 2), "first_to_last_event_us": round(ns_tot / 1e3, 2),
                         "throughput_frac": round(algo_bytes * ips / (elapsed / K * 1e9) / HBM_PEAK_GBS, 4),
                         "measured": roof_note,
                         "one_image_per_launch": one,
                         "sustained": (dict(sustained, frac=round(launch_bytes / (sustained["sum_kernels_us"] * 1e3) / HBM_PEAK_GBS, 4)) if sustained else None)},
            "host_issue_us_per_step": round((t_issued - t0) / K * 1e6, 2),
            "jfif_bytes": len(out_bytes),
            "exact_fallbacks_per_image": (int(st.exact_fallbacks) // B if st is not None else None),       # (the counter sums over the launch's B images)
            "parity": parity,
        }
        if gather is not None:
            line["gather"] = {"images_per_exchange": G, "staging_record_bytes": slot_bytes, "exchanges": gather.exchanges,
                              "streams_encoded_again_at_exact_size": gather.reencoded,
                              "gather_GBps_per_rank": round((gather.bytes_own - own0) / elapsed / 1e9, 2),
                              "note": "exact stream bytes a rank hands to the root per second of the timed region (rank 0's own stay where they are); "
                                      "at N ranks the root takes in (N - 1) x this, each peer over its own xGMI link (~64 GB/s per direction)"}
        return line, parity_ok, first_bmp

    workload = args.workload if args.workload != "auto" else "image8192"
    world_ = int(os.environ.get("WORLD_SIZE", "1"))
    ipl = args.images_per_launch or (8 if workload == "image8192" else min(32, max(1, BATCH_TOTAL // world_)))
    line, parity_ok, first_bmp = measure(workload, args.steps, args.warmup, ipl, True)
    # N > 1: BASELINE configs[3] (the batch of 64 4096^2 images at 8 ranks) as a short second leg, so that a multi-GPU run of the
    # default command reports the gathered-batch configuration as well
    if dist is not None and workload == "image8192" and args.configs3_steps > 0 and not (args.width or args.height):
        c3, c3_ok, _ = measure("batch4096", args.configs3_steps, 5, min(32, max(1, BATCH_TOTAL // world_)), False)
        parity_ok = parity_ok and c3_ok
        if rank == 0:
            line["configs3"] = {k: c3[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "jfif_bytes", "parity")}
            line["configs3"]["roofline"] = {k: c3["roofline"][k] for k in ("frac", "achieved", "sum_kernels_us", "per_image_us", "images_per_launch")}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0 if parity_ok else 3
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(first_bmp, args, args.width or 8192 if workload == "image8192" else args.width or 4096,
                                                args.height or 8192 if workload == "image8192" else args.height or 4096)
        except Exception as e:                                    # a missing checker must not hide the GPU number
            line["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(line), file=json_out, flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0 if parity_ok else 3


if __name__ == "__main__":
    sys.exit(main())
