"""Device-only throughput with TWO host threads, each running whole steps on its own resident batch (what process_dir's two GPU
workers do, without the file I/O), against one thread:  python tools/probe_two_workers.py [batch size steps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from face_crop_plus_amd import weights, engine as E

batch, size, steps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 1024, 20)
dev = torch.device("cuda:0")
sd = weights.generate_state_dict("retinaface")
bench.Telemetry.disabled = True
mk = lambda seed: bench.Pipeline(dev, sd, full=False, batch=batch, size=size, out_size=256, strategy="largest", precision="f16x3",
                                 enhance="none", streams=2, seed=seed)
p0 = mk(1)
el, faces = bench.time_pipeline(p0, steps, 5)
print(f"one thread: {el / steps * 1e3:.3f} ms/step, {int(faces.item()) / el:.1f} faces/s")
for nthreads, det_streams in ((2, 2), (2, 1)):
    pipes = [p0] + [mk(2 + i) for i in range(nthreads - 1)]
    for p in pipes:
        p.det.streams = det_streams
    ready, go = threading.Barrier(nthreads + 1), threading.Barrier(nthreads + 1)

    def run(p):
        with torch.cuda.device(dev), torch.cuda.stream(E.thread_main_stream(dev)):
            for _ in range(3):
                p.step(True)
            torch.cuda.current_stream().synchronize()
            p.face_total.zero_()
            ready.wait(); go.wait()
            for _ in range(steps // nthreads):
                p.step(True)
            torch.cuda.current_stream().synchronize()
    ts = [threading.Thread(target=run, args=(p,)) for p in pipes]
    [t.start() for t in ts]
    ready.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go.wait()
    [t.join() for t in ts]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    faces = sum(int(p.face_total.item()) for p in pipes)
    print(f"{nthreads} threads x {det_streams} detector stream(s): {el / (steps // nthreads * nthreads) * 1e3:.3f} ms/step, {faces / el:.1f} faces/s")
