"""Reduce rocprofv3 --pmc CSVs (separate FETCH_SIZE / WRITE_SIZE / SQ passes of the same bench
command) to a per-launch summary of the conv kernel.

    python tools/summarize_pmc.py gpurun_out/pmc_r1 profiles/r01_pmc_conv.json

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are KiB;
on gfx950 FETCH_SIZE counts 128-byte requests as 64 B for wide coalesced reads, so reads are doubled
(checked here on the stem launch, whose input size is known exactly)."""
import collections, csv, json, sys

src, dst = sys.argv[1], sys.argv[2]
LAUNCHES = int(sys.argv[3]) if len(sys.argv) > 3 else 66     # conv launches of one step
# optional: images per step and side of the detector input (calibration launch = layer1.0.conv1 reading the pooled stem
# map n x (side/4)^2 x 64 x 4 B once) and a description of the command
CAL_N = int(sys.argv[4]) if len(sys.argv) > 4 else 64
CAL_SIDE = int(sys.argv[5]) if len(sys.argv) > 5 else 640
WHAT = sys.argv[6] if len(sys.argv) > 6 else "bench.py --steps 1 --warmup 1 --no-cpu-baseline (batch 64, 640x640)"


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [v for v in d.values() if any(k in v["name"] for k in ("conv_igemm", "conv3x3_halo", "stem_pool_kernel", "bneck_chain"))][-LAUNCHES:]


import glob
f, w, s = (load(glob.glob(f"{src}/{k}/**/*counter_collection.csv", recursive=True)[0]) for k in ("fetch", "write", "sq"))
fetch_kib = sum(v["FETCH_SIZE"] for v in f)
write_kib = sum(v["WRITE_SIZE"] for v in w)
# Calibration launch with an exactly known read size.  fp32 path: the stem conv reads the NHWC4 fp32 batch once.
# fp16x3 path (fused uint8 stem, whose byte loads and halo re-reads are not "wide coalesced reads"): the next
# launch, layer1.0.conv1, reads the pooled 64-channel stem map (64 x 160 x 160 x 64 x 4 B) once.
fused = "stem_pool_kernel" in f[0]["name"]
cal = 1 if fused else 0
stem_expected_kib = (CAL_N * (CAL_SIDE // 4) ** 2 * 64 * 4 if fused else CAL_N * CAL_SIDE * CAL_SIDE * 16) / 1024
read_corr = 2.0
mfma_busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in s)
gui = sum(v["GRBM_GUI_ACTIVE"] for v in s)               # summed over the 8 XCDs
import os, re
_hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fcp_hip.h")).read()
out = {
    "command": f"{WHAT}, last step's {LAUNCHES} conv launches",
    # what bench.py checks before it quotes this file as `roofline.traffic` (a measurement of another build is not evidence)
    "abi_version": int(re.search(r"#define FCP_ABI_VERSION (\d+)", _hdr).group(1)), "launches_per_step": LAUNCHES,
    "fetch_size_kib_raw": fetch_kib, "write_size_kib": write_kib,
    "fetch_calibration": {"launch": f[cal]["name"][:60], "reported_kib": f[cal]["FETCH_SIZE"],
                          "expected_kib": stem_expected_kib, "ratio": f[cal]["FETCH_SIZE"] / stem_expected_kib,
                          "correction_applied": read_corr},
    "hbm_bytes_per_step": (fetch_kib * read_corr + write_kib) * 1024,
    "hbm_bytes_per_launch": (fetch_kib * read_corr + write_kib) * 1024 / LAUNCHES,
    "mfma_busy_cycles": mfma_busy, "grbm_gui_active_sum_8xcd": gui,
    "mfma_util": mfma_busy / (gui / 8 * 1024),
    "note": "SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the 1024 SIMDs (64 per "
            "v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_f16; includes zero-padded cout / stem-K work); "
            "GRBM_GUI_ACTIVE is summed over the 8 XCDs, so mfma_util = busy / (gui/8 * 1024)",
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
