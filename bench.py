"""Headline benchmark: faces/sec end-to-end (detect + align + crop) on synthetic batches.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch that is already
resident in HBM: uint8 NHWC images -> RetinaFace (MFMA convs) -> decode / NMS /
strategy -> 5-point similarity -> warpAffine crops (uint8, on device).
Workload at N=1 (and per rank at N>1) is the configuration north_star quotes the
metric on: detect + align + crop on batch 32 of synthetic 1024x1024 RGB images
(configs[3]'s per-GPU share: 256 / 8), strategy "largest", det_threshold 0.6,
output 256x256, the detector on two HIP streams.  BASELINE configs[1] (batch 64
@640x640) is measured in the same run and reported in `extra` and, as scalars,
in `config` / `roofline` (`configs1_*`); `--batch 64 --size 640` makes it the
headline.  Multi-GPU: every rank owns an independent batch (weak scaling, no
data-path collective); weights are broadcast once from rank 0 over RCCL.
`roofline.mean_sclk_mhz` / `mean_power_w` are sampled while the timed steps run
(a slow box — lower clock at the socket power cap — reads differently from a
regression).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (conv
engine), `cpu_baseline` (the oracle timed on the host cores) and — at N=1 —
`extra`: the other BASELINE configurations / modes measured in the same run
(configs[2] with and without RRDB enhancement, configs[4]'s 4K frames, the exact-fp32 mode), each with
its own ms_per_step, steps and roofline sub-record.

The detector is run the way the product runs it (`RetinaFace.streams` = 2: the
two halves of the batch on two HIP streams).  Per-launch conv durations for the
roofline are taken with HIP events in separate single-stream passes right after
the timed region: launches that share the device would inflate each other's
durations (`--streams 1` times them inside the timed region instead).
"""
from __future__ import annotations

import argparse
import gc
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
                    help="detect = detect+align+crop (the headline metric; batch 32 @1024 = north_star's geometry, "
                         "--batch 64 --size 640 = BASELINE configs[1]); "
                         "full = configs[2] (detect + RRDB enhance + align + BiSeNet parse, batch 32 @1024)")
    ap.add_argument("--enhance", default="all", choices=["all", "none", "rule"],
                    help="workload=full: which images go through RRDB (all / none / the reference's face-area rule)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 32)")
    ap.add_argument("--size", type=int, default=None, help="image side (default 1024)")
    ap.add_argument("--out-size", type=int, default=256)
    ap.add_argument("--strategy", default="largest")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` workloads (configs[2], fp32 mode)")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32"],
                    help="conv arithmetic: split-fp16 MFMA (fp32-equivalent accuracy) or exact fp32 MFMA")
    ap.add_argument("--graph", action="store_true",
                    help="replay the detection step from a captured HIP graph (small, launch-bound batches)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank: exercises the multi-GPU code path")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the detector splits a batch over (product default 2; 1 = single stream, conv "
                         "launches then timed inside the timed region)")
    ap.add_argument("--roofline-steps", type=int, default=3, help="single-stream passes the conv launches are timed in")
    ap.add_argument("--launch-table", default=None, metavar="CSV",
                    help="write the per-launch table of the roofline passes (label, us, TFLOP/s, algorithmic GB/s) to this file")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--no-telemetry", action="store_true", help="do not sample clock / power beside the timed regions")
    return ap.parse_args()


class Telemetry:
    """Engine clock + socket power of one GPU sampled on a thread while a region of the benchmark runs, so that a reader of
    the line can tell a slow box (low clock at the power cap, a lower cap) from a regression without opening profiles/.
    Sources, first that works: the amdsmi Python binding of the ROCm image; the amdgpu hwmon files in sysfs
    (power1_input | power1_average in uW, freq1_input in Hz).  Nothing here touches the data path; every failure degrades to
    ``{"source": None}``."""

    disabled = False                     # --no-telemetry

    def __init__(self, device_index=0, period_s=0.01):
        import threading
        self.period, self.samples, self._stop, self._thread = period_s, [], threading.Event(), None
        self.source, self._read, self.cap_w = None, None, None
        if Telemetry.disabled:
            return
        try:
            self._init_amdsmi(device_index)
        except Exception:
            try:
                self._init_sysfs(device_index)
            except Exception:
                self.source = None

    @staticmethod
    def _bdf(device_index):
        pr = torch.cuda.get_device_properties(device_index)
        return f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"

    def _init_amdsmi(self, device_index):
        import amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        h = handles[0]
        try:                                              # several GPUs: the one torch calls `device_index`, by PCI address
            want = self._bdf(device_index)
            for cand in handles:
                if str(amdsmi.amdsmi_get_gpu_device_bdf(cand)).lower().startswith(want):
                    h = cand
                    break
            else:
                h = handles[device_index]
        except Exception:
            h = handles[min(device_index, len(handles) - 1)]
        clk = amdsmi.AmdSmiClkType.GFX

        def num(v):
            return float(v) if isinstance(v, (int, float)) or (isinstance(v, str) and v.replace(".", "", 1).isdigit()) else None

        def read():
            pw = amdsmi.amdsmi_get_power_info(h)
            w = num(pw.get("current_socket_power"))
            if not w:
                w = num(pw.get("average_socket_power"))
            ck = amdsmi.amdsmi_get_clock_info(h, clk)
            return num(ck.get("clk", ck.get("cur_clk"))), w
        mhz, w = read()
        if mhz is None and w is None:
            raise RuntimeError("amdsmi returns no clock / power on this box")
        try:
            cap = amdsmi.amdsmi_get_power_cap_info(h)
            c = num(cap.get("power_cap"))
            self.cap_w = c / 1e6 if c and c > 1e5 else c     # uW in older bindings, W in newer ones
        except Exception:
            pass
        self._read, self.source = read, "amdsmi (amdsmi_get_clock_info GFX, amdsmi_get_power_info socket power)"

    def _init_sysfs(self, device_index):
        import glob
        want = self._bdf(device_index)
        every = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in every if want in os.path.realpath(c).lower()]          # the device torch calls `device_index`
        if not cards:
            with_hwmon = [c for c in every if os.path.isdir(os.path.join(c, "hwmon"))]
            cards = [with_hwmon[min(device_index, len(with_hwmon) - 1)]]
        hw = sorted(glob.glob(os.path.join(cards[0], "hwmon", "hwmon*")))[0]
        pfile = next(f for f in (os.path.join(hw, n) for n in ("power1_input", "power1_average")) if os.path.isfile(f))
        ffile = os.path.join(hw, "freq1_input")

        def read():
            with open(pfile) as f:
                w = float(f.read()) / 1e6
            mhz = None
            if os.path.isfile(ffile):
                with open(ffile) as f:
                    mhz = float(f.read()) / 1e6
            return mhz, w
        read()
        capf = os.path.join(hw, "power1_cap")
        if os.path.isfile(capf):
            with open(capf) as f:
                self.cap_w = float(f.read()) / 1e6
        self._read, self.source = read, f"sysfs {hw} ({os.path.basename(pfile)}, freq1_input)"

    def _loop(self):
        while not self._stop.is_set():
            try:
                self.samples.append((time.perf_counter(),) + tuple(self._read()))
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self._read is not None:
            import threading
            self.samples, self._stop = [], threading.Event()
            self._thread = threading.Thread(target=self._loop, name="fcp-telemetry", daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2)
            self._thread = None

    def summary(self, t0=None, t1=None):
        """Mean / min / max over the samples taken inside [t0, t1] (perf_counter; default: all)."""
        rows = [r for r in self.samples if (t0 is None or r[0] >= t0) and (t1 is None or r[0] <= t1)]
        mhz = [r[1] for r in rows if r[1]]
        w = [r[2] for r in rows if r[2]]
        st = lambda v: (round(sum(v) / len(v), 1), round(min(v), 1), round(max(v), 1)) if v else (None, None, None)
        (cm, cl, ch), (pm, pl, ph) = st(mhz), st(w)
        return {"source": self.source, "samples": len(rows), "period_ms": round(self.period * 1e3, 1),
                "mean_sclk_mhz": cm, "min_sclk_mhz": cl, "max_sclk_mhz": ch,
                "mean_power_w": pm, "min_power_w": pl, "max_power_w": ph, "power_cap_w": self.cap_w}


class Pipeline:
    """One configuration of the hot path on one GPU: models + a resident synthetic batch + step()."""

    def __init__(self, dev, sd_det, *, full, batch, size, out_size, strategy, precision, enhance, streams, seed,
                 graph=False, sd_enh=None, sd_par=None):
        from face_crop_plus_amd import align
        from face_crop_plus_amd.cropper import landmarks_target
        from face_crop_plus_amd.retinaface import RetinaFace
        self.dev, self.full, self.batch, self.size, self.out_size = dev, full, batch, size, out_size
        self.strategy, self.precision, self.enhance = strategy, precision, enhance
        self.align = align
        self.det = RetinaFace(strategy, 0.6).load(dev, sd_det, precision)
        self.det.streams = streams
        self.tgt = torch.from_numpy(landmarks_target((out_size, out_size), 0.65)).to(dev)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.images = torch.randint(0, 256, (batch, size, size, 3), generator=g, dtype=torch.uint8).to(dev)
        self.face_total = torch.zeros((), dtype=torch.int64, device=dev)
        self.enhanced_total = 0
        self.enh = self.par = None
        if full:
            from face_crop_plus_amd.rrdb import RRDBNet
            from face_crop_plus_amd.bise import BiSeNet
            if enhance != "none":
                self.enh = RRDBNet(0.001).load(dev, sd_enh, precision)
            self.par = BiSeNet({"glasses": [6]}, {"eyes": [4, 5]}, 32).load(dev, sd_par, precision)
        self.graphed = self.det.graphed(batch, size, size) if graph else None

    def step(self, count=True):
        from face_crop_plus_amd import trace
        imgs = self.images
        with trace.range("fcp:detect"):
            if self.graphed is not None:
                self.graphed[0].copy_(imgs)
                self.graphed[2].replay()
                res = self.graphed[1]
            else:
                res = self.det.detect(imgs, max_faces=imgs.shape[0] if self.strategy != "all" else None)
        if self.enh is not None:
            with trace.range("fcp:enhance"):
                imgs = imgs.clone()                      # enhancement rewrites the batch in place
                which = list(range(imgs.shape[0]))
                if self.enhance == "rule":               # rrdb.py:124-140 needs the landmarks on the host
                    nf_h = int(res["face_offset"][-1].item())
                    which = self.enh.gate(imgs.shape[0], imgs.shape[1], imgs.shape[2],
                                          res["landmarks"][:nf_h].cpu().numpy(), res["img_idx"][:nf_h].cpu().tolist())
                self.enhanced_total += len(which)
                self.enh.enhance_u8(imgs, which)
        with trace.range("fcp:align"):
            # the estimate kernel masks the rows beyond face_offset[n] and adds the number of faces it keeps (a live row
            # with a non-degenerate transform, cropper.py:529-531) to the device-side total: no counting launches
            crops, ok, _ = self.align.crop_align(imgs, res["img_idx"], res["landmarks"], self.tgt,
                                                 (self.out_size, self.out_size), 0,
                                                 face_count=res["face_offset"][-1:],
                                                 valid_total=self.face_total if count else None)
        if self.par is not None:
            with trace.range("fcp:parse"):
                self.par.parse(crops)                    # label maps + class histograms stay on the device
        self.last = (res, crops, ok)
        return crops

    def describe(self):
        head = (f"full pipeline (detect + RRDB enhance[{self.enhance}] + align + BiSeNet parse)" if self.full
                else "RetinaFace detect + 5-pt align/crop")
        s = (f"{head}, batch={self.batch}/GPU synthetic {self.size}x{self.size} RGB, strategy={self.strategy}, "
             f"det_threshold=0.6, output {self.out_size}x{self.out_size}")
        if self.det.streams > 1:
            s += f", detector on {self.det.streams} HIP streams x {self.batch // self.det.streams} images"
        return s


class Pipeline4K(Pipeline):
    """BASELINE configs[4] on one GPU: 3840x2160 frames resident in HBM -> batch builder (INTER_AREA to 1024x576 +
    224-px top / bottom pads, utils.py:316-335) -> detect with strategy "all" -> face-area gate -> RRDB on the gated
    frames -> un-pad + align -> BiSeNet parse.  With random-init weights the face count is an artefact of the weights
    (SURVEY.md section 8d): the detection threshold is raised until K faces per frame is in a realistic range, and K is
    reported."""

    def __init__(self, dev, sd_det, *, batch, precision, streams, seed, sd_enh, sd_par, out_size, k_max=16):
        super().__init__(dev, sd_det, full=True, batch=batch, size=1024, out_size=out_size, strategy="all",
                         precision=precision, enhance="rule", streams=streams, seed=seed, sd_enh=sd_enh, sd_par=sd_par)
        from face_crop_plus_amd import _native as N, batch as B
        from face_crop_plus_amd.align import border_code
        self.N, self.border = N, border_code("constant")
        g = torch.Generator(device="cpu").manual_seed(seed)
        fh, fw = 2160, 3840
        # structure at several scales + noise, so that INTER_AREA has something to average
        coarse = torch.randint(0, 256, (batch, fh // 8, fw // 8, 3), generator=g, dtype=torch.uint8)
        frames = coarse.repeat_interleave(8, 1).repeat_interleave(8, 2).to(dev)
        frames = (frames.to(torch.int16) + torch.randint(-24, 25, frames.shape, generator=g, dtype=torch.int16).to(dev)
                  ).clamp_(0, 255).to(torch.uint8)
        self.blob = frames.reshape(-1)
        items = np.zeros(batch, B.ITEM_DTYPE)
        ww, hh, pad, _, interp = B.batch_geometry(fh, fw, (1024, 1024))
        for i in range(batch):
            items[i] = (i * fh * fw * 3, fh, fw, hh, ww, pad[0], pad[2], interp, 0)
        self.items = items
        self.items_dev = torch.from_numpy(items.view(np.uint8).copy()).to(dev)
        self.pads = torch.tensor([pad] * batch, dtype=torch.int32, device=dev)
        self.images = torch.empty((batch, 1024, 1024, 3), dtype=torch.uint8, device=dev)
        self.frame_hw, self.resized_hw, self.pad = (fh, fw), (hh, ww), pad
        # K calibration (untimed): raise the threshold until <= k_max faces per frame survive NMS
        self._build()
        self.k_per_frame = None
        for vis in (0.6, 0.8, 0.9, 0.95, 0.98, 0.99, 0.995, 0.999, 0.9995, 0.9999):
            self.det.vis_threshold = vis
            res = self.det.detect(self.images, paddings=self.pads)
            k = int(res["face_offset"][-1].item()) / batch
            if k <= k_max:
                break
        self.k_per_frame = k

    def _build(self):
        N = self.N
        N.check(N.lib().fcp_build_batch_u8(N.ptr(self.blob), self.blob.numel(), self.items.ctypes.data, N.ptr(self.items_dev),
                                           self.batch, 1024, 1024, self.border, N.ptr(self.images), N.stream_ptr()),
                "fcp_build_batch_u8")

    def step(self, count=True):
        from face_crop_plus_amd import trace
        with trace.range("fcp:build_batch"):
            self._build()
        imgs = self.images
        with trace.range("fcp:detect"):
            res = self.det.detect(imgs, paddings=self.pads)
        nf = int(res["face_offset"][-1].item())
        with trace.range("fcp:enhance"):               # rrdb.py:124-140 (the gate only uses coordinate differences)
            which = self.enh.gate(self.batch, 1024, 1024, res["landmarks"][:nf].cpu().numpy(),
                                  res["img_idx"][:nf].cpu().tolist())
            self.enhanced_total += len(which)
            self.enh.enhance_u8(imgs, which)
        with trace.range("fcp:align"):
            crops, ok, _ = self.align.crop_align(imgs, res["img_idx"], res["landmarks"], self.tgt,
                                                 (self.out_size, self.out_size), 0, paddings=self.pads,
                                                 face_count=res["face_offset"][-1:],
                                                 valid_total=self.face_total if count else None)
        with trace.range("fcp:parse"):
            if crops.shape[0]:
                self.par.parse(crops)
        self.last = (res, crops, ok)
        return crops

    def describe(self):
        return (f"full pipeline on 4K frames (batch builder + detect + RRDB enhance[rule] + align + BiSeNet parse), batch="
                f"{self.batch}/GPU synthetic 3840x2160 RGB -> 1024x576 + 224-px pads, strategy=all, det_threshold="
                f"{self.det.vis_threshold} (raised until K = {self.k_per_frame:.1f} faces per frame), output "
                f"{self.out_size}x{self.out_size}")


def time_pipeline(p: Pipeline, steps, warmup, autotune=True, dist=None, live_events=False, telemetry=None):
    """Initialisation pass (lazy loads, tile tuning), `warmup` untimed steps, then exactly `steps` timed steps
    bracketed by barrier + synchronize.  Returns (elapsed seconds, faces counted in the timed steps).  ``telemetry``: a
    ``Telemetry`` that samples clock / power while the timed steps run; its summary of exactly that window is left in
    ``p.telemetry``."""
    from face_crop_plus_amd import engine as E
    E.Autotune.enabled = autotune
    p.step(True)
    torch.cuda.synchronize()
    E.Autotune.enabled = False
    for _ in range(warmup):
        p.step(True)
    torch.cuda.synchronize()
    p.face_total.zero_()
    p.enhanced_total = 0
    if live_events:
        E.ConvStats.timing = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if telemetry is not None:
        telemetry.__enter__()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.step(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if telemetry is not None:
        telemetry.__exit__()
        p.telemetry = telemetry.summary(t0, t1)
    return t1 - t0, p.face_total.clone()


def _pmc_traffic(key, launches=None):
    """Committed PMC measurement (HBM bytes per conv launch) of workload `key`, newest round first: counters cannot be
    read from inside the process, so `traffic` is the rocprofv3 --pmc measurement of the same command.  A file recorded at
    another ABI version than the library's, or for a step with another number of conv launches than this run's, is a
    measurement of a different build: it is named, with the reason, and NOT quoted (traffic = None).
    Returns (bytes per launch | None, "profiles/<file>" | None, stale reason | None)."""
    pdir = os.path.join(ROOT, "profiles")
    prof = sorted(f for f in os.listdir(pdir) if f.endswith(f"_{key}.json")) if os.path.isdir(pdir) else []
    if not prof:
        return None, None, None
    with open(os.path.join(pdir, prof[-1])) as f:
        rec = json.load(f)
    from face_crop_plus_amd._native import ABI_VERSION
    stale = None
    if rec.get("abi_version") != ABI_VERSION:
        stale = f"recorded at ABI {rec.get('abi_version', 'unknown')}, the library is ABI {ABI_VERSION}"
    elif launches is not None and rec.get("launches_per_step") != launches:
        stale = f"recorded for {rec.get('launches_per_step')} conv launches per step, this step has {launches}"
    return (None if stale else round(rec["hbm_bytes_per_launch"])), "profiles/" + prof[-1], stale


# the workloads whose `roofline.traffic` bench.py quotes from profiles/ (tools/profile_round.sh regenerates all of them in one call)
TRAFFIC_KEYS = ("c3det_pmc", "f16x3_pmc_conv", "c3_pmc", "rrdb_pmc", "f32_pmc_conv")

AUTOTUNE_ROOFLINE = True            # --no-autotune / FCP_AUTOTUNE=0 hold for the roofline passes too (recorded in the entry)
LAUNCH_TABLE = None                 # --launch-table: CSV path for the per-launch table of the headline workload's roofline passes


def conv_roofline(p: Pipeline, nsteps, live, timed_ms=None, traffic_key=None, table=False):
    """Roofline record of the conv engine from per-launch HIP events on the launch stream: the events of the timed
    steps (`live`, single-stream runs) or of `nsteps` extra single-stream passes right after the timed region.
    `timed_ms`: ms per step of the timed region (the product configuration, possibly two detector streams): the record
    then also carries `frac_timed` = algorithmic conv FLOP / whole-step time / peak — a lower bound of what the conv
    engine sustained inside the timed region itself (the step also holds the non-conv kernels)."""
    from face_crop_plus_amd import engine as E
    timed_streams = p.det.streams
    pass_sclk = None                    # mean engine clock of the per-launch passes (sampled when the launch table is written)
    if not live:
        saved, p.det.streams = p.det.streams, 1
        graphed, p.graphed = p.graphed, None          # per-launch events need the eager launches
        # the full-batch shapes of the single-stream pass are new to the tile tuner (the timed region tuned the
        # sub-batch shapes of its streams): tune them in an untimed step, like the timed region's initialisation pass
        tuned, E.Autotune.enabled = E.Autotune.enabled, AUTOTUNE_ROOFLINE
        p.step(False)
        torch.cuda.synchronize()
        E.Autotune.enabled = tuned
        E.ConvStats.timing = []
        tele = Telemetry(p.dev.index or 0, period_s=0.005) if table else None
        if tele is not None:
            tele.__enter__()
        for _ in range(nsteps):
            p.step(False)
        torch.cuda.synchronize()
        if tele is not None:
            tele.__exit__()
            pass_sclk = tele.summary().get("mean_sclk_mhz")
        p.det.streams, p.graphed = saved, graphed
    timing, E.ConvStats.timing = E.ConvStats.timing, None
    if LAUNCH_TABLE and timed_ms is not None and table:
        # Per launch: measured us beside two floors — algorithmic HBM bytes at the rate the streaming copy kernel reaches
        # (6.3 TB/s) and EXECUTED matrix FLOP (3 MFMAs per product on the fp16x3 path) at the dense f16 peak scaled to the
        # mean engine clock of these passes.  floor_max = perfect overlap of the two, floor_sum = none (at the socket power cap
        # joules add: DESIGN.md section 6); us / floor_sum <= 1.15 is physics on this socket, above it a kernel problem.
        # tools/launch_ledger.py adds W / MHz / joules per launch from steady loops of each launch alone.
        import csv
        per = len(timing) // nsteps
        mult = 3 if p.precision == "f16x3" else 1
        peak_at_clock = (F16_MFMA_PEAK_TFLOPS if mult == 3 else FP32_MFMA_PEAK_TFLOPS) * 1e12 * ((pass_sclk or 2400.0) / 2400.0)
        with open(LAUNCH_TABLE, "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["launch", "us", "algorithmic_tflops", "algorithmic_gb_per_s", "executed_gflop", "algorithmic_mb", "pass_mean_sclk_mhz",
                         "floor_hbm_us", "floor_mfma_us", "floor_max_us", "floor_sum_us", "us_over_floor_max", "us_over_floor_sum"])
            for i in range(per):
                us = sum(timing[i + s_ * per][0].elapsed_time(timing[i + s_ * per][1]) for s_ in range(nsteps)) / nsteps * 1e3
                f_hbm, f_mfma = timing[i][4] / 6.3e12 * 1e6, timing[i][2] * mult / peak_at_clock * 1e6
                wr.writerow([timing[i][3], round(us, 1), round(timing[i][2] / us / 1e6, 1), round(timing[i][4] / us / 1e3, 1),
                             round(timing[i][2] * mult / 1e9, 2), round(timing[i][4] / 1e6, 2), pass_sclk,
                             round(f_hbm, 1), round(f_mfma, 1), round(max(f_hbm, f_mfma), 1), round(f_hbm + f_mfma, 1),
                             round(us / max(f_hbm, f_mfma), 2), round(us / (f_hbm + f_mfma), 2)])
    conv_ms = sum(t[0].elapsed_time(t[1]) for t in timing) / nsteps
    conv_flops = sum(t[2] for t in timing) / nsteps
    launches = len(timing) // nsteps
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12
    split = p.precision == "f16x3"
    peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
    traffic, traffic_src, traffic_stale = _pmc_traffic(traffic_key, launches) if traffic_key else (None, None, None)
    rec = {"bound": "mfma",
           "kernel": ("conv_igemm_f16x3 family (3x v_mfma_f32_32x32x16_f16 per product: x = hi + lo split)" if split
                      else "conv_igemm_f32 (v_mfma_f32_32x32x2_f32)"),
           "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
           # the split path executes 3 matrix FLOP per algorithmic FLOP: utilisation of the f16 pipe
           "executed_frac": round(achieved * (3 if split else 1) / peak, 4),
           "streams": 1, "autotuned": bool(AUTOTUNE_ROOFLINE),
           "traffic": traffic,
           "traffic_unit": "HBM bytes per conv launch (PMC, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
           "traffic_source": traffic_src,
           # PMC counters cannot be read from inside the process: `traffic` is the committed rocprofv3 --pmc
           # measurement of this same command, not a value measured in this run
           "traffic_static": traffic is not None,
           "traffic_stale": traffic_stale,        # why the newest committed measurement is not quoted (None: it is, or there is none)
           "measured": ("HIP events around every conv launch of the timed steps" if live else
                        f"HIP events around every conv launch of {nsteps} single-stream passes over the batch right "
                        f"after the timed region"),
           "launches_per_step": launches, "algorithmic_gflop_per_step": round(conv_flops / 1e9, 2),
           "avg_launch_ms": round(conv_ms / launches, 4), "conv_ms_per_step": round(conv_ms, 3)}
    if timed_ms is not None:
        timed = conv_flops / (timed_ms * 1e-3) / 1e12
        rec.update(achieved_timed=round(timed, 2), frac_timed=round(timed / peak, 4), timed_streams=timed_streams,
                   timed_note=f"algorithmic conv FLOP of a step / ms_per_step of the timed region ({timed_streams} detector "
                              f"stream(s); the step also contains the non-conv kernels, so this is a lower bound); "
                              f"`achieved` / `frac` come from per-launch events of single-stream passes")
    return rec


def hbm_kernel_records(p: Pipeline, reps=20):
    """HBM-bound (non-conv) kernels of the step, each timed alone with HIP events on the launch stream over `reps`
    back-to-back launches on the tensors of the last step: SURVEY.md 8(d) "HBM GB/s for the non-conv kernels".
    bytes = algorithmic bytes, peak = 8 TB/s (MI355X_MICROARCH.md)."""
    from face_crop_plus_amd import _native as N, retinaface as RF, align
    res, crops, ok = p.last
    det, dev = p.det, p.dev
    n, h, w = p.images.shape[:3]
    heads = res["heads"]
    P = res["cand_score"].shape[1]
    lib, st = N.lib(), N.stream_ptr()
    nf = min(int(res["face_offset"][-1].item()), res["max_faces"])
    ncand = int(res["cand_count"].sum().item())
    f = res["landmarks"].shape[0]
    oh = ow = p.out_size
    ws = torch.empty((int(lib.fcp_retina_nms_workspace_bytes(n, P)),), dtype=torch.uint8, device=dev)
    mat = torch.empty((f, 6), dtype=torch.float64, device=dev)
    okb = torch.empty((f,), dtype=torch.int32, device=dev)
    pads = getattr(p, "pads", None)

    def decode():
        N.check(lib.fcp_retina_decode(heads[0].ptr(), heads[1].ptr(), heads[2].ptr(), n, h, w, float(det.vis_threshold),
                                      float(det.variance[0]), float(det.variance[1]), N.ptr(res["cand_score"]),
                                      N.ptr(res["cand_box"]), N.ptr(res["cand_ldm"]), N.ptr(res["cand_prior"]),
                                      N.ptr(res["cand_count"]), None, None, None, st))

    def nms():
        N.check(lib.fcp_retina_nms_select(N.ptr(res["cand_score"]), N.ptr(res["cand_box"]), N.ptr(res["cand_count"]), n, P,
                                          float(det.nms_threshold), RF.STRATEGIES[det.strategy], N.ptr(ws),
                                          N.ptr(res["keep_pos"]), N.ptr(res["keep_count"]), N.ptr(res["sel_pos"]),
                                          N.ptr(res["sel_count"]), st))

    def gather():
        N.check(lib.fcp_retina_gather_faces(N.ptr(res["cand_ldm"]), N.ptr(res["sel_pos"]), N.ptr(res["sel_count"]), n, P,
                                            N.ptr(pads), f, N.ptr(res["face_offset"]), N.ptr(res["landmarks"]),
                                            N.ptr(res["img_idx"]), st))

    def estimate():
        N.check(lib.fcp_estimate_transform_counted(N.ptr(res["landmarks"]), N.ptr(p.tgt), f, 5, 0, N.ptr(res["face_offset"][-1:]),
                                                   N.ptr(mat), N.ptr(okb), None, st))

    def warp():
        N.check(lib.fcp_warp_affine_u8(N.ptr(p.images), n, h, w, N.ptr(res["img_idx"]), N.ptr(mat), N.ptr(okb), N.ptr(pads),
                                       f, oh, ow, 0, N.ptr(crops), st))

    estimate()
    torch.cuda.synchronize()
    m = mat[:max(nf, 1)].cpu().numpy().reshape(-1, 6)
    det2 = np.abs(m[:, 0] * m[:, 4] - m[:, 1] * m[:, 3])
    src_px = np.minimum(np.where(det2 > 0, oh * ow / np.maximum(det2, 1e-30), 0.0), h * w)   # source footprint per face
    work = {   # kernel -> (fn, algorithmic bytes per launch, what they are)
        "retina_decode_kernel": (decode, n * P * 16 * 4 + ncand * (4 + 16 + 40 + 4),
                                 "16 fp32 head values per prior read + compacted candidates written"),
        "retina_nms_kernel": (nms, ncand * (4 + 16) * 2 + ncand * 8,
                              "candidate scores + boxes read, sorted keys / boxes written once (latency / LDS bound)"),
        "retina_gather_kernel": (gather, n * 4 * 2 + nf * (40 + 4) * 2, "selected landmarks gathered (launch-latency bound)"),
        "estimate_transform_kernel": (estimate, f * (40 + 48 + 4), "5 points in, 2x3 f64 + flag out (launch-latency bound)"),
        "warp_affine_kernel<4>": (warp, nf * oh * ow * 3 + int(src_px.sum()) * 3,
                                  "crop bytes written + source footprint (crop area / |det M|) read once"),
    }
    out = {}
    for name, (fn, nbytes, what) in work.items():
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"bound": "hbm", "avg_launch_us": round(ms * 1e3, 2), "bytes": int(nbytes), "bytes_are": what,
                     "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(nbytes / (ms * 1e-3) / 1e9 / 8000.0, 4)}
    out["_note"] = (f"{reps} back-to-back launches each, HIP events on the launch stream, tensors of the last timed step "
                    f"(n={n}, P={P}, {ncand} candidates, {nf} faces); back-to-back launches include the ~1.5 us "
                    f"dependent-launch boundary")
    return out


def run_extra(dev, sds, args):
    """The other configurations, one GPU, measured like the headline (same Pipeline / timing / roofline code)."""
    out = {}

    def one(key, note, steps, warmup, cls=Pipeline, traffic_key=None, hbm=False, **kw):
        try:
            if cls is Pipeline:
                kw["strategy"] = args.strategy
            p = cls(dev, sds["retinaface"], out_size=args.out_size, streams=args.streams,
                    seed=4321, sd_enh=sds.get("rrdb"), sd_par=sds.get("bisenet"), **kw)
            elapsed, faces = time_pipeline(p, steps, warmup, autotune=not args.no_autotune, telemetry=Telemetry(dev.index or 0))
            rec = {"workload": p.describe(), "note": note, "value": round(int(faces.item()) / elapsed, 2),
                   "unit": "faces/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
                   "dtype": p.precision}
            if p.enh is not None:
                rec["images_enhanced_per_step"] = p.enhanced_total / steps
            if cls is Pipeline4K:
                rec["faces_per_frame"] = round(int(faces.item()) / steps / p.batch, 2)
                rec["frames_per_s"] = round(p.batch * steps / elapsed, 2)
            rec["roofline"] = conv_roofline(p, 1 if p.enh is not None else 2, live=False,
                                            timed_ms=elapsed / steps * 1e3, traffic_key=traffic_key)
            attach_telemetry(rec["roofline"], getattr(p, "telemetry", None))
            if hbm:
                rec["hbm_kernels"] = hbm_kernel_records(p)
            out[key] = rec
            del p
        except Exception as e:                           # an extra must never take the headline line down
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        gc.collect()
        torch.cuda.empty_cache()

    from face_crop_plus_amd import weights
    sds = dict(sds, rrdb=weights.generate_state_dict("rrdb"), bisenet=weights.generate_state_dict("bisenet"))
    if (args.batch, args.size) != (64, 640):
        one("c2_detect_align_crop_640", "BASELINE configs[1]: detect + align + crop on batch 64 @640x640 (the headline of rounds "
            "1-5)", 20, 5, full=False, batch=64, size=640, precision="f16x3", enhance="none", traffic_key="f16x3_pmc_conv", hbm=True)
    if (args.batch, args.size) != (32, 1024):
        one("c3_detect_align_crop_1024", "the north-star metric at the north-star geometry: detect + align + crop (no parse, no "
            "RRDB) on batch 32 @1024x1024 — the per-GPU rate that decides >= 10 k faces/s on 8 GPUs (needs >= 1250)", 20, 5,
            full=False, batch=32, size=1024, precision="f16x3", enhance="none", traffic_key="c3det_pmc", hbm=True)
    one("c3_full_no_enhance", "BASELINE configs[2] without enhancement: detect + align + BiSeNet parse", 5, 2,
        full=True, batch=32, size=1024, precision="f16x3", enhance="none", traffic_key="c3_pmc")
    one("c3_full_enhance_all", "configs[2] with RRDB on EVERY image (worst case), batch reduced to 2: the enhancer "
        "costs ~37.6 TFLOP per 1024x1024 image", 2, 1, full=True, batch=2, size=1024, precision="f16x3", enhance="all",
        traffic_key="rrdb_pmc")
    one("c3_full_enhance_rule", "configs[2] with the reference's face-area gate (rrdb.py:124-140), batch 8", 2, 1,
        full=True, batch=8, size=1024, precision="f16x3", enhance="rule")
    one("c5_4k_all", "BASELINE configs[4] on one GPU: 4K frames, strategy=all, RRDB by the reference's gate; the frame "
        "decode / H2D is outside the timed region, the resize + pad batch builder inside", 2, 1, cls=Pipeline4K,
        batch=4, precision="f16x3")
    one("c2_detect_f32", "headline workload in the exact-fp32 mode (v_mfma_f32_32x32x2_f32, peak 157.3 TFLOP/s)", 5, 2,
        full=False, batch=64, size=640, precision="f32", enhance="none", traffic_key="f32_pmc_conv")
    return out


def main():
    args = parse()
    global LAUNCH_TABLE, AUTOTUNE_ROOFLINE
    LAUNCH_TABLE = args.launch_table
    AUTOTUNE_ROOFLINE = not args.no_autotune and os.environ.get("FCP_AUTOTUNE", "1") != "0"
    Telemetry.disabled = args.no_telemetry
    if args.no_autotune:
        from face_crop_plus_amd import engine as _E
        _E.Autotune.use_tables = False              # heuristic tiles: neither tuning launches nor the shipped / user tables
                                                    # (what FCP_TUNE_TABLES=0 does; FCP_AUTOTUNE=0 alone keeps the tables)
    full = args.workload == "full"
    if args.batch is None:
        args.batch = 32
    if args.size is None:
        args.size = 1024
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                        # never returns: one rank per GPU under torch.distributed.run
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible: refusing to run "
                         f"{args.gpus} ranks on fewer devices")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"RCCL group has {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")

    from face_crop_plus_amd import weights

    # Every network the workload runs comes from rank 0 through one flat RCCL broadcast each (north_star: "RCCL broadcast of
    # weights + per-rank independent batches"; 228.9 MB for the three) — what the product's loader
    # (weights.load_state_dict) does inside a process group.  The other ranks' generated copies are only the key / shape
    # templates the flat buffer is cut by; broadcast_state_dict zeroes them before the collective.
    names = ["retinaface"] + (["rrdb", "bisenet"] if full else [])
    sds = {k: weights.generate_state_dict(k) for k in names}
    broadcast = None
    if dist is not None:
        from face_crop_plus_amd.dist import broadcast_state_dict
        torch.cuda.synchronize()
        tb = time.perf_counter()
        nbytes = 0
        for k in names:
            sds[k] = broadcast_state_dict(sds[k], dev)
            nbytes += sum(4 * v.numel() for kk, v in sds[k].items() if not kk.endswith("num_batches_tracked"))
        torch.cuda.synchronize()
        broadcast = {"networks": names, "bytes": int(nbytes), "ms": round((time.perf_counter() - tb) * 1e3, 2),
                     "note": "one flat fp32 RCCL broadcast per network from rank 0, incl. the host-side flatten / H2D / D2H / unflatten"}
    sd = sds["retinaface"]
    p = Pipeline(dev, sd, full=full, batch=args.batch, size=args.size, out_size=args.out_size, strategy=args.strategy,
                 precision=args.precision, enhance=args.enhance if full else "none", streams=args.streams,
                 seed=1234 + rank, graph=args.graph, sd_enh=sds.get("rrdb"), sd_par=sds.get("bisenet"))
    live = rank == 0 and args.streams <= 1 and not args.graph
    tele = Telemetry(local_rank) if rank == 0 else None
    elapsed, faces = time_pipeline(p, args.steps, args.warmup, autotune=not args.no_autotune, dist=dist, live_events=live,
                                   telemetry=tele)
    last_timed = p.last                            # outputs of the LAST TIMED step (the roofline passes below overwrite p.last)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(faces, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_faces = int(faces.item())

    roofline = cpu_baseline = extra = parity_check = None
    hbm_kernels = None
    if rank == 0:
        # committed PMC measurement of the same workload (profiles/): north-star geometry or configs[1]
        key = None
        if not full and (args.batch, args.size) == (32, 1024) and args.precision == "f16x3":
            key = "c3det_pmc"
        elif not full and (args.batch, args.size) == (64, 640):
            key = "f16x3_pmc_conv" if args.precision == "f16x3" else "f32_pmc_conv"
        roofline = conv_roofline(p, args.steps if live else args.roofline_steps, live,
                                 timed_ms=elapsed / args.steps * 1e3, traffic_key=key, table=True)
        attach_telemetry(roofline, getattr(p, "telemetry", None))
        if not args.graph:
            try:
                hbm_kernels = hbm_kernel_records(p)
            except Exception as e:                       # a side record must never take the headline line down
                hbm_kernels = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, parity_check = run_cpu_baseline(sd, p.images[:32].cpu(), args, p.tgt.cpu().numpy(), last_timed)
    describe = p.describe()
    if rank == 0 and world == 1 and not args.no_extra and not full:
        del p
        gc.collect()
        torch.cuda.empty_cache()
        extra = run_extra(dev, sds, args)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": "faces/sec end-to-end (detect+align+crop)",
            "value": round(total_faces / elapsed, 2), "unit": "faces/s",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist is not None else 0), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3" if args.precision == "f16x3" else "f32",
            "data": "synthetic (uniform uint8 images resident in HBM; seeded random-init weights; file I/O excluded)",
            "config": {"workload": describe,
                       "global_batch": args.batch * world, "image_size": args.size, "parallelism": f"dp{world}",
                       "faces_per_step": total_faces / max(args.steps, 1),
                       "images_per_s": round(args.batch * world * args.steps / elapsed, 2)},
            "roofline": roofline, "hbm_kernels": hbm_kernels, "cpu_baseline": cpu_baseline,
            "parity_check": parity_check,
        }
        if broadcast is not None:
            line["weight_broadcast"] = broadcast
            line["config"]["broadcast_bytes"], line["config"]["broadcast_ms"] = broadcast["bytes"], broadcast["ms"]
        if isinstance(hbm_kernels, dict) and roofline is not None:
            # SURVEY 8(d) "HBM GB/s for the non-conv kernels", compactly inside `roofline` as well (the full records stay in
            # `hbm_kernels`): kernel -> [us per launch, algorithmic GB/s, fraction of the 8 TB/s peak]
            roofline["hbm_bound_kernels"] = {k: [v["avg_launch_us"], v["achieved"], v["frac"]] for k, v in hbm_kernels.items()
                                             if isinstance(v, dict) and "achieved" in v}
        line = finalize_line(line, extra)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def attach_telemetry(roofline, tele):
    """Clock / power of the timed region into a roofline record: the two means as scalars (what a reader compares between
    boxes / rounds), the whole summary beside them, and the MFMA peak the mean clock allows (peak x sclk / 2400 MHz) with the
    fraction of THAT the timed region reached."""
    if roofline is None or not tele:
        return roofline
    roofline["mean_sclk_mhz"], roofline["mean_power_w"] = tele.get("mean_sclk_mhz"), tele.get("mean_power_w")
    roofline["telemetry"] = dict(tele, window="the timed steps (between the two synchronisations)")
    if tele.get("mean_sclk_mhz") and roofline.get("achieved_timed"):
        at_clock = roofline["peak"] * tele["mean_sclk_mhz"] / 2400.0
        roofline["peak_at_mean_sclk"] = round(at_clock, 1)
        roofline["frac_timed_at_mean_sclk"] = round(roofline["achieved_timed"] / at_clock, 4)
    return roofline


def finalize_line(line, extra):
    """Attach the `extra` records, lift the north-star record (batch 32 @1024^2, detect + align + crop: the per-GPU rate that
    decides >= 10 k faces/s on 8 GPUs) into `config` / `roofline` — nested, and as scalars for consumers that flatten nested
    records — and fix the key order: the contract's scalar fields first (the line starts with {"metric": ...), the long side
    records in the middle, `config` / `roofline` / `cpu_baseline` last, so that a consumer that keeps only the tail of the line
    still sees the workload, the roofline and the north-star record."""
    if extra is not None:
        line["extra"] = extra
        c1 = extra.get("c2_detect_align_crop_640", {})
        if "value" in c1:                                   # BASELINE configs[1] beside a north-star-geometry headline
            for where in (line["roofline"], line["config"]):
                where["configs1_value"] = c1["value"]
                where["configs1_unit"] = "faces/s, batch 64 @640x640, detect+align+crop, 1 GPU (BASELINE configs[1])"
                where["configs1_ms_per_step"], where["configs1_steps"] = c1["ms_per_step"], c1["steps"]
                where["configs1_frac"] = c1["roofline"]["frac"]
                where["configs1_frac_timed"] = c1["roofline"].get("frac_timed")
                where["configs1_mean_sclk_mhz"] = c1["roofline"].get("mean_sclk_mhz")
                where["configs1_mean_power_w"] = c1["roofline"].get("mean_power_w")
        ns = extra.get("c3_detect_align_crop_1024", {})
        if "value" in ns:
            line["config"]["north_star_geometry"] = {
                "workload": ns["workload"], "value": ns["value"], "unit": ns["unit"], "ms_per_step": ns["ms_per_step"],
                "steps": ns["steps"], "warmup": ns["warmup"], "roofline_frac": ns["roofline"]["frac"],
                "roofline_frac_timed": ns["roofline"].get("frac_timed")}
            for where in (line["roofline"], line["config"]):
                where["north_star_geometry_value"] = ns["value"]
                where["north_star_geometry_unit"] = "faces/s, batch 32 @1024x1024, detect+align+crop, 1 GPU"
                where["north_star_geometry_ms_per_step"] = ns["ms_per_step"]
                where["north_star_geometry_steps"] = ns["steps"]
                where["north_star_geometry_frac"] = ns["roofline"]["frac"]
                where["north_star_geometry_frac_timed"] = ns["roofline"].get("frac_timed")
    last = [k for k in ("config", "roofline", "cpu_baseline") if k in line]
    line = {**{k: v for k, v in line.items() if k not in last}, **{k: line[k] for k in last}}
    assert next(iter(line)) == "metric"
    return line


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command line as N ranks (one per GPU) under
    torch.distributed.run on 127.0.0.1, so that a plain invocation can never fall through to a silent 1-rank run."""
    import socket
    import subprocess
    if not torch.cuda.is_available() or torch.cuda.device_count() < n:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        raise SystemExit(f"--gpus {n} but only {have} GPU(s) are visible: refusing to run {n} ranks on fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


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


def cpu_model():
    """Model name of the host CPU (BASELINE.md section 4 asks for the CPU model and core count beside the CPU baseline)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


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


def run_cpu_baseline(sd, images_u8, args, tgt, last=None):
    """The oracle (CPU restatement of the reference path, kind="port") on a bounded
    sample of the same workload, all host cores.  The oracle's outputs on those images are then used as the checker
    of the timed batch itself (`parity_check`): selected-face indices must be equal, landmarks within 1e-3 px
    (north_star's tolerance), and the GPU crops byte-equal to the oracle's estimate + warp of the GPU's landmarks
    (cropper.py:514-547); the run fails otherwise."""
    from oracle import retinaface_ref as R, align_ref as A
    cores = pick_cpu_threads()
    torch.set_num_threads(cores)
    x = images_u8.permute(0, 3, 1, 2).float()

    def run(k):
        t = time.perf_counter()
        lm, idx = R.predict(x[:k], sd, args.strategy, 0.6)
        crops = A.crop_align(images_u8[:k].numpy(), None, idx, lm, tgt, (args.out_size, args.out_size), "constant")
        return time.perf_counter() - t, len(crops), lm, idx

    t1, _, _, _ = run(1)                 # warm-up + per-image cost estimate
    k = int(max(2, min(images_u8.shape[0], args.cpu_seconds / max(t1, 1e-3))))
    t, nf, lm_ref, idx_ref = run(k)
    base = {"value": round(nf / t, 3), "unit": "faces/s", "cores": cores, "host_cores": host_cores(),
            "host_logical_cpus": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port",
            "cores_note": "cores = torch threads the oracle ran on (the fastest of a short conv probe: oversubscribed boxes get "
                          "slower with every core); host_cores = cores this process may use (affinity mask / cgroup quota)",
            "sample": f"{k} images of the same synthetic {args.size}x{args.size} batch, torch-CPU fp32 + numpy oracle, "
                      f"{t:.1f} s wall"}
    return base, (check_against_oracle(last, images_u8, k, lm_ref, idx_ref, tgt, args.out_size, A) if last is not None else None)


def check_against_oracle(last, images_u8, k, lm_ref, idx_ref, tgt, out_size, A, pads=None):
    """Outputs of the last timed step on images 0..k-1 against the oracle's on the same images."""
    res, crops, ok = last
    nf = int(res["face_offset"][-1].item())
    idx = res["img_idx"][:nf].cpu().numpy()
    sel = idx < k                                          # faces are ordered by image (retinaface.py:363-408)
    lm = res["landmarks"][:nf].cpu().numpy()[sel]
    idx_l = idx[sel].tolist()
    indices_equal = idx_l == [int(i) for i in idx_ref]
    err = float(np.abs(lm - np.asarray(lm_ref)).max()) if indices_equal and len(idx_l) else (0.0 if indices_equal else float("nan"))
    okh = ok[:nf].cpu().numpy()[sel] != 0
    crops_ref = A.crop_align(images_u8[:k].numpy(), pads, idx_l, lm, tgt, (out_size, out_size), "constant")
    got = crops[:nf].cpu().numpy()[sel][okh]
    differing = int((got != crops_ref).sum()) if got.shape == crops_ref.shape else -1
    rec = {"images": k, "faces": len(idx_l), "indices_equal": bool(indices_equal), "max_landmark_err_px": err,
           "tolerance_px": 1e-3, "crop_bytes_differing": differing, "crop_bytes_compared": int(got.size),
           "checker": "oracle/retinaface_ref.predict + oracle/align_ref.crop_align on the first images of the timed batch "
                      "(outputs of the last timed step)"}
    rec["vacuous"] = len(idx_l) == 0
    # zero faces on both sides would "agree" with nothing compared: a regression that suppresses every detection must not pass
    if len(idx_l) == 0 or int(got.size) == 0 or not indices_equal or not (err < 1e-3) or differing != 0:
        print(json.dumps({"parity_check": rec}), file=sys.stderr, flush=True)
        raise SystemExit("bench.py: the timed batch does not match the oracle (see parity_check on stderr)")
    return rec


if __name__ == "__main__":
    main()
