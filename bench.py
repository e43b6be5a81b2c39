"""Headline benchmark: faces/sec end-to-end (detect + align + crop) on synthetic batches.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch that is already
resident in HBM: uint8 NHWC images -> RetinaFace (fp32 MFMA convs) -> decode /
NMS / strategy -> 5-point similarity -> warpAffine crops (uint8, on device).
Workload at N=1 is BASELINE.json configs[1]: batch 64, 640x640, strategy
"largest", det_threshold 0.6, output 256x256.  Multi-GPU: every rank owns an
independent batch (weak scaling, no data-path collective); weights are
broadcast once from rank 0 over RCCL.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (conv
engine, fp32 MFMA peak) and `cpu_baseline` (the oracle timed on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (not the 2:1-sparse figure)
RETINA_GFLOP_1024 = 226.64      # algorithmic FLOP / image @1024^2 (scales with H*W)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="detect", choices=["detect", "full"],
                    help="detect = BASELINE configs[1] (detect+align+crop, the headline metric); "
                         "full = configs[2] (detect + RRDB enhance + align + BiSeNet parse, batch 32 @1024)")
    ap.add_argument("--enhance", default="all", choices=["all", "none", "rule"],
                    help="workload=full: which images go through RRDB (all / none / the reference's face-area rule)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (64 detect / 32 full)")
    ap.add_argument("--size", type=int, default=None, help="image side (640 detect / 1024 full)")
    ap.add_argument("--out-size", type=int, default=256)
    ap.add_argument("--strategy", default="largest")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32"],
                    help="conv arithmetic: split-fp16 MFMA (fp32-equivalent accuracy) or exact fp32 MFMA")
    ap.add_argument("--graph", action="store_true",
                    help="replay the detection step from a captured HIP graph (small, launch-bound batches)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank: exercises the multi-GPU code path")
    ap.add_argument("--free-running", action="store_true",
                    help="with --streams S: do not re-join the streams after every step (S independent workers)")
    ap.add_argument("--no-live-roofline", action="store_true",
                    help="time the conv launches in one extra step after the timed region instead of inside it")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the batch is split over (2: the two half-batches overlap on the device like "
                         "process_dir's GPU workers, +4..6 %% over 1; the default stays 1 so that per-launch durations "
                         "- HIP events here, rocprofv3 in profiles/ - are those of kernels that own the device)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    return ap.parse_args()


def main():
    args = parse()
    full = args.workload == "full"
    if args.batch is None:
        args.batch = 32 if full else 64
    if args.size is None:
        args.size = 1024 if full else 640
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from face_crop_plus_amd import weights, align, engine as E
    from face_crop_plus_amd.retinaface import RetinaFace

    sd = weights.generate_state_dict("retinaface")
    if dist is not None:
        from face_crop_plus_amd.dist import broadcast_state_dict
        sd = broadcast_state_dict(sd, dev)        # rank 0's weights -> all ranks, one flat RCCL broadcast
    det = RetinaFace(args.strategy, 0.6).load(dev, sd, args.precision)
    from face_crop_plus_amd.cropper import landmarks_target
    tgt = torch.from_numpy(landmarks_target((args.out_size, args.out_size), 0.65)).to(dev)

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    images = torch.randint(0, 256, (args.batch, args.size, args.size, 3), generator=g, dtype=torch.uint8).to(dev)
    face_total = torch.zeros((), dtype=torch.int64, device=dev)

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else [None]
    chunks = list(torch.chunk(images, len(streams)))
    face_total_s = {st: torch.zeros((), dtype=torch.int64, device=dev) for st in streams if st is not None}

    enh = par = None
    if full:
        from face_crop_plus_amd.rrdb import RRDBNet
        from face_crop_plus_amd.bise import BiSeNet
        if args.enhance != "none":
            enh = RRDBNet(0.001).load(dev, weights.generate_state_dict("rrdb"), args.precision)
        par = BiSeNet({"glasses": [6]}, {"eyes": [4, 5]}, 32).load(dev, weights.generate_state_dict("bisenet"),
                                                                   args.precision)

    graphed = det.graphed(args.batch, args.size, args.size) if args.graph else None

    def step_chunk(imgs, count):
        nonlocal graphed
        if graphed is not None:
            graphed[0].copy_(imgs)
            graphed[2].replay()
            res = graphed[1]
        else:
            res = det.detect(imgs, max_faces=imgs.shape[0] if args.strategy != "all" else None)
        if enh is not None:
            imgs = imgs.clone()                      # enhancement rewrites the batch in place
            which = list(range(imgs.shape[0]))
            if args.enhance == "rule":               # rrdb.py:124-140 needs the landmarks on the host
                nf_h = int(res["face_offset"][-1].item())
                which = enh.gate(imgs.shape[0], imgs.shape[1], imgs.shape[2], res["landmarks"][:nf_h].cpu().numpy(),
                                 res["img_idx"][:nf_h].cpu().tolist())
            enh.enhance_u8(imgs, which)
        crops, ok, _ = align.crop_align(imgs, res["img_idx"], res["landmarks"], tgt,
                                        (args.out_size, args.out_size), 0)
        if par is not None:
            par.parse(crops)                         # label maps + class histograms stay on the device
        if count:
            nf = torch.clamp(res["face_offset"][-1].to(torch.int64), max=res["max_faces"])
            valid = (torch.arange(res["max_faces"], device=dev) < nf) & (ok != 0)
            return crops, valid.sum()
        return crops, None

    def step(count=True):
        """One pass of the hot path over the batch.  With --streams S the batch is split into S
        independent sub-batches on S HIP streams, so one sub-batch's tail wave of workgroups
        overlaps the next one's head (the path has no cross-image dependency)."""
        if streams[0] is None:
            crops, nv = step_chunk(images, count)
            if count:
                face_total.add_(nv)
            return crops
        cur = torch.cuda.current_stream()
        outs = []
        for st, imgs in zip(streams, chunks):
            if not args.free_running:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(step_chunk(imgs, count))
                if count and args.free_running:
                    face_total_s[st].add_(outs[-1][1])
        if args.free_running:
            return [c for c, _ in outs]          # the streams stay de-phased, like process_dir's GPU workers
        for st in streams:
            cur.wait_stream(st)
        if count:
            for _, nv in outs:
                face_total.add_(nv)
        return [c for c, _ in outs]

    # one untimed initialisation pass (lazy module loads, per-shape tile tuning), then the W warm-up steps, which are
    # identical to the timed steps
    E.Autotune.enabled = not args.no_autotune
    step(True)
    torch.cuda.synchronize()
    E.Autotune.enabled = False
    for _ in range(args.warmup):
        step(True)
    torch.cuda.synchronize()
    face_total.zero_()
    # roofline evidence: HIP events around every conv launch of the timed steps themselves, on the launch stream
    # (one event pair costs ~2 us of host time; --no-live-roofline measures one extra step after the timed region
    # instead, as do --graph / --streams, whose launches cannot carry per-launch events)
    live = rank == 0 and not args.no_live_roofline and graphed is None and streams[0] is None
    if live:
        E.ConvStats.timing = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for v in face_total_s.values():
        v.zero_()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for v in face_total_s.values():
        face_total.add_(v)
    faces = face_total.clone()
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(faces, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_faces = int(faces.item())

    # ---- roofline of the dominant kernel (conv engine): per-launch HIP events on the launch stream
    roofline = None
    if rank == 0:
        nsteps = args.steps
        if not live:
            E.ConvStats.timing = []
            graphed_saved, graphed = graphed, None      # per-launch events need the eager launches
            for imgs in chunks:                         # sub-batches one after the other on ONE stream: kernels that
                step_chunk(imgs, False)                 # share the device would inflate each other's durations
            graphed = graphed_saved
            torch.cuda.synchronize()
            nsteps = 1
        conv_ms = sum(a.elapsed_time(b) for a, b, _ in E.ConvStats.timing) / nsteps
        conv_flops = sum(f for _, _, f in E.ConvStats.timing) / nsteps
        launches = len(E.ConvStats.timing) // nsteps
        E.ConvStats.timing = None
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_conv.json")) \
            if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        prof = [f for f in prof if ("f16x3" in f) == (args.precision == "f16x3")]
        if prof and args.batch == 64 and args.size == 640:
            # HBM bytes per conv launch from the committed rocprofv3 --pmc passes of this same command
            with open(os.path.join(ROOT, "profiles", prof[-1])) as f:
                traffic = round(json.load(f)["hbm_bytes_per_launch"])
            traffic_src = "profiles/" + prof[-1]
        split = args.precision == "f16x3"
        peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        roofline = {"bound": "mfma",
                    "kernel": ("conv_igemm_f16x3 (3x v_mfma_f32_32x32x16_f16 per product: x = hi + lo split)" if split
                               else "conv_igemm_f32 (v_mfma_f32_32x32x2_f32)"),
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4),
                    # the split path executes 3 matrix FLOP per algorithmic FLOP: utilisation of the f16 pipe
                    "executed_frac": round(achieved * (3 if split else 1) / peak, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per conv launch (PMC, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                    "traffic_source": traffic_src,
                    "measured": ("HIP events around every conv launch of the timed steps" if live
                                 else "HIP events around every conv launch of one extra single-stream pass over the batch after the timed region"),
                    "launches_per_step": launches, "algorithmic_gflop_per_step": round(conv_flops / 1e9, 2),
                    "avg_launch_ms": round(conv_ms / launches, 4), "conv_ms_per_step": round(conv_ms, 3)}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(sd, images[:32].cpu(), args, tgt.cpu().numpy())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": "faces/sec end-to-end (detect+align+crop)",
            "value": round(total_faces / elapsed, 2), "unit": "faces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3" if args.precision == "f16x3" else "f32",
            "data": "synthetic (uniform uint8 images resident in HBM; seeded random-init weights; file I/O excluded)",
            "config": {"workload": (f"full pipeline (detect + RRDB enhance[{args.enhance}] + align + BiSeNet parse)" if full
                                    else "RetinaFace detect + 5-pt align/crop") +
                                   f", batch={args.batch}/GPU synthetic "
                                   f"{args.size}x{args.size} RGB, strategy={args.strategy}, det_threshold=0.6, "
                                   f"output {args.out_size}x{args.out_size}" +
                                   (f", {len(streams)} HIP streams x {args.batch // len(streams)} images" if len(streams) > 1 else ""),
                       "global_batch": args.batch * world, "image_size": args.size, "parallelism": f"dp{world}",
                       "faces_per_step": total_faces / max(args.steps, 1),
                       "images_per_s": round(args.batch * world * args.steps / elapsed, 2)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def pick_cpu_threads():
    """Thread count for the CPU baseline: the fastest of a few candidates on a short
    conv probe (oversubscribed boxes get *slower* with every core)."""
    import torch.nn.functional as F
    n = host_cores()
    cands = sorted({c for c in (8, 16, 32, 64, n) if c <= n} | {min(n, 8)})
    x = torch.randn(2, 64, 160, 160)
    w = torch.randn(64, 64, 3, 3)
    best, best_t = cands[0], float("inf")
    for c in cands:
        torch.set_num_threads(c)
        F.conv2d(x, w, padding=1)
        t = time.perf_counter()
        for _ in range(3):
            F.conv2d(x, w, padding=1)
        t = time.perf_counter() - t
        if t < best_t:
            best, best_t = c, t
    return best


def run_cpu_baseline(sd, images_u8, args, tgt):
    """The oracle (CPU restatement of the reference path, kind="port") on a bounded
    sample of the same workload, all host cores."""
    from oracle import retinaface_ref as R, align_ref as A
    cores = pick_cpu_threads()
    torch.set_num_threads(cores)
    x = images_u8.permute(0, 3, 1, 2).float()

    def run(k):
        t = time.perf_counter()
        lm, idx = R.predict(x[:k], sd, args.strategy, 0.6)
        crops = A.crop_align(images_u8[:k].numpy(), None, idx, lm, tgt, (args.out_size, args.out_size), "constant")
        return time.perf_counter() - t, len(crops)

    t1, _ = run(1)                       # warm-up + per-image cost estimate
    k = int(max(2, min(images_u8.shape[0], args.cpu_seconds / max(t1, 1e-3))))
    t, nf = run(k)
    return {"value": round(nf / t, 3), "unit": "faces/s", "cores": cores, "kind": "port",
            "sample": f"{k} images of the same synthetic {args.size}x{args.size} batch, torch-CPU fp32 + numpy oracle, "
                      f"{t:.1f} s wall"}


if __name__ == "__main__":
    main()
