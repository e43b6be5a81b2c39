"""Per-launch ledger of one detector step: every conv-engine launch of a single-stream step is

  (a) timed IN SITU with HIP events (what bench.py's launch table holds), and
  (b) re-launched ALONE in a steady loop for ~SECS seconds while engine clock and socket power are sampled (bench.Telemetry),
      which gives W, MHz and joules per launch for each of them,

and priced against two floors:  floor_hbm_us = algorithmic HBM bytes / 6.3 TB/s (the rate the copy kernel reaches),
floor_mfma_us = EXECUTED matrix FLOP (3 MFMAs per product on the fp16x3 path) / (2.5 PFLOP/s x mean sclk of its loop / 2400 MHz);
floor_max = max of the two (perfect overlap), floor_sum = their sum (none: joules add at the socket power cap).

    python tools/launch_ledger.py OUT.csv [batch size secs]          (GPU box; FCP_BOUNDARY=ctypes is forced)

A launch whose in-situ time is <= 1.15 x floor_sum is physics on this socket; above that it is a kernel problem."""
import csv
import os
import sys
import time

os.environ["FCP_BOUNDARY"] = "ctypes"            # the replay closures re-enqueue C-ABI descriptors
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from face_crop_plus_amd import engine as E, weights
from face_crop_plus_amd.retinaface import RetinaFace

HBM_RATE = 6.3e12                                # B/s: what the streaming copy kernel sustains (profiles/r04_probes.md)
PEAK = 2.5e15                                    # dense f16 MFMA FLOP/s at 2400 MHz


def main():
    out_csv = sys.argv[1]
    batch, size, secs = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (32, 1024, 0.4)
    dev = torch.device("cuda:0")
    det = RetinaFace("largest", 0.6).load(dev, weights.generate_state_dict("retinaface"))
    det.streams = 1
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randint(0, 256, (batch, size, size, 3), generator=g, dtype=torch.uint8).to(dev)
    det.detect(imgs, max_faces=batch)            # tile tuning / lazy init
    torch.cuda.synchronize()
    E.Autotune.enabled = False
    # (a) in situ: three steps with per-launch events
    E.ConvStats.timing = []
    for _ in range(3):
        det.detect(imgs, max_faces=batch)
    torch.cuda.synchronize()
    timing, E.ConvStats.timing = E.ConvStats.timing, None
    per = len(timing) // 3
    insitu = [sum(timing[i + k * per][0].elapsed_time(timing[i + k * per][1]) for k in range(3)) / 3 * 1e3 for i in range(per)]
    # capture the relaunch closures of one step (they keep their tensors alive)
    E.ConvStats.replay = []
    det.detect(imgs, max_faces=batch)
    torch.cuda.synchronize()
    replay, E.ConvStats.replay = E.ConvStats.replay, None
    assert len(replay) == per and all(r[0] == timing[i][3] for i, r in enumerate(replay)), "capture and timing disagree"
    tele = bench.Telemetry(0, period_s=0.005)
    rows = []
    for i, (label, flops, byts, relaunch) in enumerate(replay):
        relaunch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(20, int(secs * 1e6 / max(insitu[i], 5.0)))
        with tele:
            t_a = time.perf_counter()
            e0.record()
            for _ in range(reps):
                relaunch()
            e1.record()
            torch.cuda.synchronize()
            t_b = time.perf_counter()
        lead = min(0.1, 0.25 * (t_b - t_a))                                   # the clock settles within the first tenth of a second
        s = tele.summary(t_a + lead, t_b)
        loop_us = e0.elapsed_time(e1) / reps * 1e3
        mhz, watts = s["mean_sclk_mhz"], s["mean_power_w"]
        executed = flops * (3 if det.precision == 1 else 1)
        f_hbm = byts / HBM_RATE * 1e6
        f_mfma = executed / (PEAK * (mhz or 2400.0) / 2400.0) * 1e6
        rows.append({"launch": label, "us_in_situ": round(insitu[i], 1), "us_loop": round(loop_us, 1),
                     "algorithmic_gflop": round(flops / 1e9, 2), "executed_gflop": round(executed / 1e9, 2),
                     "algorithmic_mb": round(byts / 1e6, 2), "loop_sclk_mhz": mhz, "loop_power_w": watts,
                     "joules_per_launch": round(watts * loop_us * 1e-6, 4) if watts else None,
                     "floor_hbm_us": round(f_hbm, 1), "floor_mfma_us": round(f_mfma, 1),
                     "floor_max_us": round(max(f_hbm, f_mfma), 1), "floor_sum_us": round(f_hbm + f_mfma, 1),
                     "us_over_floor_max": round(insitu[i] / max(f_hbm, f_mfma), 2),
                     "us_over_floor_sum": round(insitu[i] / (f_hbm + f_mfma), 2)})
        print(rows[-1], flush=True)
    with open(out_csv, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0]))
        wr.writeheader()
        wr.writerows(rows)
    tot = lambda k: sum(r[k] for r in rows if r[k])
    print(f"step: {tot('us_in_situ') / 1e3:.2f} ms in situ, floor_max {tot('floor_max_us') / 1e3:.2f} ms, floor_sum {tot('floor_sum_us') / 1e3:.2f} ms, "
          f"{tot('joules_per_launch'):.1f} J per step (loop power x loop time); telemetry source: {tele.source}")
    over = [r for r in rows if r["us_over_floor_sum"] > 1.15]
    print(f"{len(over)} of {len(rows)} launches above 1.15 x floor_sum, {sum(r['us_in_situ'] for r in over) / 1e3:.2f} ms of the step:")
    for r in sorted(over, key=lambda r: -r["us_in_situ"]):
        print(f"  {r['launch']}: {r['us_in_situ']} us = {r['us_over_floor_sum']} x floor_sum ({r['floor_hbm_us']} + {r['floor_mfma_us']})")


if __name__ == "__main__":
    main()
