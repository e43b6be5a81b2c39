"""bench.py's own plumbing that needs no GPU: the launcher guard of --gpus N and the oracle checker of the timed batch."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_devices_fails_loudly_instead_of_running_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "refusing to run 2 ranks" in r.stderr + r.stdout
    assert '"metric"' not in r.stdout                       # no JSON line from a silent 1-rank run
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")   # a launcher that started the wrong number of ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr + r.stdout


def test_parity_checker_accepts_the_oracle_and_rejects_a_shifted_landmark(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    from oracle import align_ref as A
    rng = np.random.default_rng(4)
    k, out = 3, 32
    images = torch.from_numpy(rng.integers(0, 256, (k + 1, 96, 96, 3), dtype=np.uint8))
    tgt = A.landmarks_target((out, out), 0.65)
    # one face per image: a similarity image of the target somewhere inside the picture
    lm = np.stack([tgt * 1.4 + np.array([10.0 + 5 * i, 14.0]) for i in range(k + 1)]).astype(np.float32)
    idx = list(range(k + 1))
    crops = A.crop_align(images.numpy(), None, idx, lm, tgt, (out, out), "constant")
    last = ({"face_offset": torch.tensor([0, 1, 2, 3, 4]), "img_idx": torch.tensor(idx, dtype=torch.int32), "landmarks": torch.from_numpy(lm)},
            torch.from_numpy(crops), torch.ones(k + 1, dtype=torch.int32))
    rec = bench.check_against_oracle(last, images, k, lm[:k], idx[:k], tgt, out, A)
    assert rec["indices_equal"] and rec["faces"] == k and rec["max_landmark_err_px"] == 0.0 and rec["crop_bytes_differing"] == 0
    assert rec["crop_bytes_compared"] == k * out * out * 3
    for breakage in ("landmark", "index", "crop"):
        lm_ref, idx_ref, bad = lm[:k].copy(), idx[:k], last
        if breakage == "landmark":
            lm_ref[1, 2, 0] += 0.01                        # 1e-2 px from the oracle: above the 1e-3 tolerance
        elif breakage == "index":
            idx_ref = [0, 2, 2]
        else:
            c2 = crops.copy(); c2[0, 5, 5, 1] ^= 1
            bad = (last[0], torch.from_numpy(c2), last[2])
        with pytest.raises(SystemExit, match="does not match the oracle"):
            bench.check_against_oracle(bad, images, k, lm_ref, idx_ref, tgt, out, A)


def test_parity_checker_rejects_a_vacuous_pass():
    """Zero faces on both sides (a regression that suppresses every detection in the oracle AND on the GPU) must not count as
    parity: nothing was compared."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import align_ref as A
    out = 32
    images = torch.zeros((2, 64, 64, 3), dtype=torch.uint8)
    tgt = A.landmarks_target((out, out), 0.65)
    last = ({"face_offset": torch.tensor([0, 0, 0]), "img_idx": torch.zeros(4, dtype=torch.int32), "landmarks": torch.zeros(4, 5, 2)},
            torch.zeros((4, out, out, 3), dtype=torch.uint8), torch.zeros(4, dtype=torch.int32))
    with pytest.raises(SystemExit, match="does not match the oracle"):
        bench.check_against_oracle(last, images, 2, np.zeros((0, 5, 2), np.float32), [], tgt, out, A)


def test_eight_gpu_preflight_self_launch_and_guards(monkeypatch):
    """Pre-flight for the 8-GPU node (no such node in any round so far): `python bench.py --gpus 8` re-executes itself as
    eight ranks under torch.distributed.run on 127.0.0.1 with dmabuf IPC kept in the environment, and every mismatch between
    --gpus, WORLD_SIZE and the visible devices stops the run before any rank could silently measure the wrong thing."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return 0

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.self_launch(8)
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]        # the ranks get the caller's own flags
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                       # RCCL needs dmabuf IPC on this driver
    # fewer devices than ranks: refused by the launcher path ...
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(SystemExit, match="only 4 GPU"):
        bench.self_launch(8)
    # ... and by a rank that a foreign launcher started (WORLD_SIZE = 8 on a 4-GPU box), before it touches a device
    for k, v in (("WORLD_SIZE", "8"), ("RANK", "5"), ("LOCAL_RANK", "5")):
        monkeypatch.setenv(k, v)
    with pytest.raises(SystemExit, match="only 4 GPU"):
        bench.main()
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.main()


def test_cpu_baseline_host_description():
    sys.path.insert(0, ROOT)
    import bench
    assert isinstance(bench.cpu_model(), str) and bench.cpu_model()
    assert 1 <= bench.host_cores() <= (os.cpu_count() or 1)


def test_bench_line_shape_for_consumers_that_flatten_or_keep_the_tail():
    """The JSON line must start with the contract's {"metric": ...}, end with config / roofline / cpu_baseline, and carry the
    north-star record (batch 32 @1024^2) as scalars inside `roofline` and `config` (a parser that drops nested records keeps them)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    line = {"metric": "faces/sec end-to-end (detect+align+crop)", "value": 3500.0, "unit": "faces/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 18.0, "config": {"workload": "w"}, "roofline": {"frac": 0.12}, "hbm_kernels": {"k": {"x" * 50: 1}},
            "cpu_baseline": {"value": 2.3}, "parity_check": {"faces": 32}}
    extra = {"c3_detect_align_crop_1024": {"workload": "n", "value": 1380.0, "unit": "faces/s", "ms_per_step": 23.2, "steps": 10, "warmup": 3,
                                            "roofline": {"frac": 0.1326, "frac_timed": 0.125}},
             "c3_full_no_enhance": {"error": "x"}}
    out = bench.finalize_line(line, extra)
    keys = list(out)
    assert keys[0] == "metric" and keys[-3:] == ["config", "roofline", "cpu_baseline"] and "extra" in keys[:-3]
    text = json.dumps(out)
    assert text.startswith('{"metric"')
    for where in ("roofline", "config"):
        assert out[where]["north_star_geometry_value"] == 1380.0 and out[where]["north_star_geometry_frac"] == 0.1326
        assert out[where]["north_star_geometry_ms_per_step"] == 23.2 and out[where]["north_star_geometry_steps"] == 10
    assert text.rindex('"north_star_geometry_value"') > text.rindex('"hbm_kernels"')      # visible in the tail
    # round 6: the north-star geometry IS the headline; BASELINE configs[1] comes from `extra` and is lifted as configs1_* scalars
    line6 = {"metric": "faces/sec end-to-end (detect+align+crop)", "value": 1400.0, "config": {"workload": "w"}, "roofline": {"frac": 0.13},
             "cpu_baseline": {"value": 1.0}}
    extra6 = {"c2_detect_align_crop_640": {"workload": "c", "value": 3500.0, "unit": "faces/s", "ms_per_step": 18.2, "steps": 20, "warmup": 5,
                                           "roofline": {"frac": 0.112, "frac_timed": 0.122, "mean_sclk_mhz": 2100.0, "mean_power_w": 1390.0}}}
    out6 = bench.finalize_line(line6, extra6)
    for where in ("roofline", "config"):
        assert out6[where]["configs1_value"] == 3500.0 and out6[where]["configs1_frac_timed"] == 0.122
        assert out6[where]["configs1_mean_sclk_mhz"] == 2100.0 and "north_star_geometry_value" not in out6[where]
    plain = bench.finalize_line({"metric": "m", "value": 1.0, "config": {}, "roofline": {}}, None)   # --no-extra: nothing to lift
    assert list(plain) == ["metric", "value", "config", "roofline"]


def test_telemetry_samples_a_window_and_lands_in_the_roofline_record():
    """Clock / power sampling beside the timed region: a reader thread, mean / min / max over exactly the timed window, the two
    means as scalars of the roofline record and the MFMA peak re-priced at the mean clock; no source -> no fields, no failure."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    t = bench.Telemetry.__new__(bench.Telemetry)              # no GPU here: feed the reader by hand
    import threading
    t.period, t.samples, t._stop, t._thread, t.cap_w = 0.002, [], threading.Event(), None, 1400.0
    seq = iter([(2400.0, 300.0)] * 3 + [(2000.0 + 10 * i, 1390.0 + (i % 3)) for i in range(10000)])
    t._read, t.source = (lambda: next(seq)), "fake"
    with t:
        time.sleep(0.03)
        t0 = time.perf_counter()
        time.sleep(0.08)
        t1 = time.perf_counter()
        time.sleep(0.02)
    s = t.summary(t0, t1)
    assert 10 <= s["samples"] < len(t.samples) and s["source"] == "fake" and s["power_cap_w"] == 1400.0
    assert s["min_sclk_mhz"] > 2000.0 and s["max_sclk_mhz"] < 2400.0 + 10 * len(t.samples)      # the idle head is outside the window
    assert s["min_sclk_mhz"] <= s["mean_sclk_mhz"] <= s["max_sclk_mhz"] and 1390.0 <= s["mean_power_w"] <= 1392.0
    roof = bench.attach_telemetry({"peak": 2500.0, "achieved_timed": 300.0, "frac_timed": 0.12}, dict(s, mean_sclk_mhz=2000.0))
    assert roof["mean_sclk_mhz"] == 2000.0 and roof["mean_power_w"] == s["mean_power_w"]
    assert roof["peak_at_mean_sclk"] == round(2500.0 * 2000 / 2400, 1) and roof["frac_timed_at_mean_sclk"] == round(300.0 / (2500.0 * 2000 / 2400), 4)
    assert bench.attach_telemetry({"peak": 1.0}, None) == {"peak": 1.0}
    dead = bench.Telemetry.__new__(bench.Telemetry)
    dead.period, dead.samples, dead._stop, dead._thread, dead._read, dead.source, dead.cap_w = 0.01, [], threading.Event(), None, None, None, None
    with dead:
        pass
    assert dead.summary()["mean_sclk_mhz"] is None and dead.summary()["samples"] == 0
    real = bench.Telemetry(0)                                  # this container has no GPU: must degrade, not raise
    assert real.source is None or isinstance(real.source, str)
