"""Two independent half-batch pipelines on two streams (no fork / join per step) against the product's
fork-join split of one batch — probe for the step-boundary cost of RetinaFace.streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from face_crop_plus_amd import weights, engine as E
dev = torch.device("cuda:0")
sd = weights.generate_state_dict("retinaface")
kw = dict(full=False, size=640, out_size=256, strategy="largest", precision="f16x3", enhance="none", seed=1234)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(pipes, streams, stagger=False):
    for p in pipes:
        p.step(True)
    torch.cuda.synchronize()
    E.Autotune.enabled = False
    for _ in range(3):
        for p, s in zip(pipes, streams):
            with torch.cuda.stream(s):
                p.step(True)
    torch.cuda.synchronize()
    if stagger:                      # start the second pipeline half a step late
        with torch.cuda.stream(streams[0]):
            pipes[0].step(True)
        time.sleep(0.011)
    t0 = time.perf_counter()
    for _ in range(steps):
        for p, s in zip(pipes, streams):
            with torch.cuda.stream(s):
                p.step(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(p.batch for p in pipes) * steps / dt


one = bench.Pipeline(dev, sd, batch=64, streams=2, **kw)
print(f"one pipeline, batch 64, fork-join over 2 streams: {run([one], [torch.cuda.current_stream()]):8.1f} faces/s", flush=True)
del one
two = [bench.Pipeline(dev, sd, batch=32, streams=1, **kw) for _ in range(2)]
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
print(f"two pipelines, batch 32 each, own streams:        {run(two, ss):8.1f} faces/s", flush=True)
print(f"same, second one started half a step late:        {run(two, ss, True):8.1f} faces/s", flush=True)
three = two + [bench.Pipeline(dev, sd, batch=32, streams=1, **kw)]
ss.append(torch.cuda.Stream())
print(f"three pipelines, batch 32 each:                   {run(three, ss):8.1f} faces/s", flush=True)
del two, three
full = [bench.Pipeline(dev, sd, batch=64, streams=1, **kw) for _ in range(2)]
print(f"two pipelines, batch 64 each (whole steps alternate between two streams): {run(full, ss[:2]):8.1f} faces/s", flush=True)
full3 = full + [bench.Pipeline(dev, sd, batch=64, streams=1, **kw)]
print(f"three pipelines, batch 64 each:                   {run(full3, ss[:3]):8.1f} faces/s", flush=True)
