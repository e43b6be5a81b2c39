"""End-to-end process_dir throughput including file decode / encode (not the bench metric) — helper.
   python tools/bench_process_dir.py [n_files] [size] [num_processes] [io_threads]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from face_crop_plus_amd import Cropper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
nproc = int(sys.argv[3]) if len(sys.argv) > 3 else 2
io = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with tempfile.TemporaryDirectory() as d:
    src, dst = os.path.join(d, "in"), os.path.join(d, "out")
    os.makedirs(src)
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (size // 8, size // 8, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((size, size), Image.BICUBIC))
    for i in range(n):
        Image.fromarray(np.roll(img, i, 1)).save(os.path.join(src, f"{i:05d}.jpg"), quality=90)
    c = Cropper(resize_size=size, batch_size=int(os.environ.get("FCP_BENCH_BATCH", "64")), num_processes=nproc, device="cuda:0", weights={"retinaface": "generated"})
    c.gpu_workers = nproc                                     # exactly nproc GPU worker threads (the product default: max(2, num_processes))
    if io:
        c.io_threads = io
    if os.environ.get("FCP_IO_PROCS"):                      # "readers,writers"
        c.io_processes = tuple(int(v) for v in os.environ["FCP_IO_PROCS"].split(","))
    if os.environ.get("FCP_SWITCH_US"):
        sys.setswitchinterval(float(os.environ["FCP_SWITCH_US"]) * 1e-6)
    c.process_dir(src, dst + "_warm", desc=None)
    t0 = time.time()
    c.process_dir(src, dst, desc=None)
    dt = time.time() - t0
    procs = c._io_procs
    print(f"{n} jpg {size}x{size}, num_processes={nproc}, io_threads={c.io_threads}, io_processes="
          f"{(procs.readers, procs.writers) if procs is not None else 'off (threads)'}, host cores {os.cpu_count()}: "
          f"{n / dt:.1f} images/s ({len(os.listdir(dst))} crops written)")
    if os.environ.get("FCP_STREAM_MATRIX"):                 # which of the workers' streams share a hardware queue (engine._streams_overlap)
        import torch
        from face_crop_plus_amd import engine as E
        names, streams = ["D"], [torch.cuda.default_stream(c.device)]
        for i, slot in enumerate(E._stream_sets.get(c.device.index or 0, [])):
            if slot["main"] is not None:
                names.append(f"w{i}.main"); streams.append(slot["main"])
            for k, v in slot["side"].items():
                for j, s_ in enumerate(v):
                    names.append(f"w{i}.s{j}"); streams.append(s_)
        groups = []
        for nm, s_ in zip(names, streams):
            for g in groups:
                if not E._streams_overlap(g[0][1], s_):
                    g.append((nm, s_)); break
            else:
                groups.append([(nm, s_)])
        print("   hardware queues: " + " | ".join(",".join(nm for nm, _ in g) for g in groups))
