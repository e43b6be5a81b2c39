"""Two detector streams with CU masks (hipExtStreamCreateWithCUMask): does giving each half-batch its own CUs / XCDs beat letting
the two streams' workgroups interleave over the whole device?   python tools/probe_cu_mask.py [batch size]
Masks tried (256 CUs = 8 x 32-bit words): none (product), low / high halves, even / odd words, even / odd bits, 3/4 + 3/4 overlapping."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from face_crop_plus_amd import weights, engine as E

batch, size = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 1024)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = ctypes.CDLL(torch.utils.cpp_extension.ROCM_HOME + "/lib/libamdhip64.so") if False else ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


F, Z = 0xFFFFFFFF, 0
masks = {
    "low | high halves": ([F, F, F, F, Z, Z, Z, Z], [Z, Z, Z, Z, F, F, F, F]),
    "even | odd words": ([F, Z, F, Z, F, Z, F, Z], [Z, F, Z, F, Z, F, Z, F]),
    "even | odd bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "even | odd nibbles": ([0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8),
    "all | all (masked API, no restriction)": ([F] * 8, [F] * 8),
}
sd = weights.generate_state_dict("retinaface")
bench.Telemetry.disabled = True
p = bench.Pipeline(dev, sd, full=False, batch=batch, size=size, out_size=256, strategy="largest", precision="f16x3", enhance="none", streams=2, seed=1)
for rep in range(2):
    el, faces = bench.time_pipeline(p, 20, 5)
    print(f"product streams: {el / 20 * 1e3:.3f} ms/step, {int(faces.item()) / el:.1f} faces/s", flush=True)
    for name, (ma, mb) in masks.items():
        st = E.thread_streams(dev)
        keep = st["side"].get((2, False))
        st["side"][(2, False)] = [masked_stream(ma), masked_stream(mb)]
        try:
            el, faces = bench.time_pipeline(p, 20, 5)
            print(f"{name}: {el / 20 * 1e3:.3f} ms/step, {int(faces.item()) / el:.1f} faces/s", flush=True)
        finally:
            st["side"][(2, False)] = keep
