"""Multi-process plumbing on a real MI355X (the driver's box has ONE GPU):

* two ranks sharing cuda:0 shard a directory with ``dist.shard`` (gloo process group: RCCL refuses two ranks
  on one device) — the union of their outputs must equal the single-process output set byte for byte;
* a world-size-1 RCCL ("nccl") group on the GPU: the flat weight broadcast and the sum / max reductions of
  ``dist.py`` run through RCCL itself, which the CPU gloo tests cannot show.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KW = dict(output_size=64, resize_size=160, strategy="all", det_threshold=0.55, batch_size=2, device="cuda:0",
          weights={"retinaface": "generated"})


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, src, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from face_crop_plus_amd import Cropper, weights
    from face_crop_plus_amd import dist as D
    # ranks start from different weights; rank 0's are broadcast (bench.py does the same over RCCL)
    sd = weights.generate_state_dict("retinaface", seed=0 if rank == 0 else 5)
    sd = D.broadcast_state_dict(sd)
    kw = dict(KW, weights={"retinaface": sd})
    c = Cropper(**kw)
    c.process_dir(src, os.path.join(out, f"rank{rank}"), desc=None)
    n = len(os.listdir(os.path.join(out, f"rank{rank}"))) if os.path.isdir(os.path.join(out, f"rank{rank}")) else 0
    total = D.all_reduce_scalar(n, "sum")
    with open(os.path.join(out, f"total{rank}.txt"), "w") as f:
        f.write(str(int(total)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_union_equals_single_process(tmp_path, device):
    from PIL import Image
    from face_crop_plus_amd import Cropper
    src = tmp_path / "src"
    src.mkdir()
    rng = np.random.default_rng(9)
    for i in range(11):
        h, w = int(rng.integers(100, 180)), int(rng.integers(100, 180))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(src / f"f{i:02d}.png")
    single = tmp_path / "single"
    Cropper(**KW).process_dir(str(src), str(single), desc=None)
    ref = {f: (single / f).read_bytes() for f in sorted(os.listdir(single))}
    assert len(ref) > 8
    out = tmp_path / "ranks"
    out.mkdir()
    mp.spawn(_shard_worker, args=(2, _free_port(), str(src), str(out)), nprocs=2, join=True)
    got = {}
    per_rank = []
    for r in (0, 1):
        d = out / f"rank{r}"
        files = sorted(os.listdir(d)) if d.is_dir() else []
        per_rank.append(files)
        for f in files:
            assert f not in got, f"{f} written by both ranks"
            got[f] = (d / f).read_bytes()
    assert per_rank[0] and per_rank[1], "one rank did no work"
    assert got == ref, "union of the ranks' outputs differs from the single-process output set"
    assert (out / "total0.txt").read_text() == (out / "total1.txt").read_text() == str(len(ref))


def _rccl_worker(rank, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from face_crop_plus_amd import weights
    from face_crop_plus_amd import dist as D
    # the three networks of configs[2] / [3] (what `bench.py --workload full --gpus N` broadcasts; 228.9 MB)
    same, nbytes = True, 0
    for name in ("retinaface", "rrdb", "bisenet"):
        sd = weights.generate_state_dict(name)
        bc = D.broadcast_state_dict(sd, device=torch.device("cuda:0"))
        keys = [k for k in sd if not k.endswith("num_batches_tracked")]
        same = same and all(torch.equal(bc[k], sd[k]) for k in keys)
        nbytes += sum(4 * sd[k].numel() for k in keys)
    s = D.all_reduce_scalar(41.0, "sum", device=torch.device("cuda:0"))
    m = D.all_reduce_scalar(7.5, "max", device=torch.device("cuda:0"))
    with open(out, "w") as f:
        f.write(f"{int(same)} {s} {m} {dist.get_backend()} {nbytes}")
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world1_broadcast_and_reduce(tmp_path, device):
    out = tmp_path / "rccl.txt"
    mp.spawn(_rccl_worker, args=(_free_port(), str(out)), nprocs=1, join=True)
    same, s, m, backend, nbytes = out.read_text().split()
    assert same == "1" and float(s) == 41.0 and float(m) == 7.5 and backend == "nccl"
    assert 220e6 < int(nbytes) < 240e6                          # detector + enhancer + parser, fp32


def _cli_worker(rank, world, port, src, out, ckpt):
    """`python -m face_crop_plus_amd` under a launcher: rank 0 alone reads the checkpoint, everybody crops."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", FCP_DIST_BACKEND="gloo", FCP_OFFLINE="1",
                      FCP_WEIGHTS_DIR=ckpt if rank == 0 else os.path.join(out, "nowhere"), TORCH_HOME=os.path.join(out, "th"))
    os.environ.pop("FCP_WEIGHTS", None)
    from face_crop_plus_amd import weights as W
    touched = []
    if rank != 0:
        W.find_checkpoint = lambda m: touched.append(m)
        torch.load = lambda *a, **k: touched.append(a)
    from face_crop_plus_amd.__main__ import main
    main(["-i", src, "-o", os.path.join(out, f"rank{rank}"), "-s", "64", "-r", "160", "-st", "all", "-dt", "0.55", "-b", "2"])
    with open(os.path.join(out, f"touched{rank}.txt"), "w") as f:
        f.write(str(len(touched)))


def test_cli_two_ranks_rank0_loads_and_broadcasts(tmp_path, device):
    from PIL import Image
    from face_crop_plus_amd import Cropper, weights as W
    src, ckpt, out = tmp_path / "src", tmp_path / "ckpt", tmp_path / "out"
    for d in (src, ckpt, out):
        d.mkdir()
    rng = np.random.default_rng(10)
    for i in range(7):
        Image.fromarray(rng.integers(0, 256, (140, 150, 3), dtype=np.uint8)).save(src / f"g{i}.png")
    sd = W.generate_state_dict("retinaface")
    torch.save(sd, ckpt / "retinaface_detector.pth")
    single = tmp_path / "single"
    Cropper(**dict(KW, weights={"retinaface": sd})).process_dir(str(src), str(single), desc=None)
    ref = {f: (single / f).read_bytes() for f in sorted(os.listdir(single))}
    mp.spawn(_cli_worker, args=(2, _free_port(), str(src), str(out), str(ckpt)), nprocs=2, join=True)
    got = {}
    for r in (0, 1):
        d = out / f"rank{r}"
        for f in (sorted(os.listdir(d)) if d.is_dir() else []):
            assert f not in got
            got[f] = (d / f).read_bytes()
    assert got == ref and len(ref) > 4
    assert (out / "touched1.txt").read_text() == "0", "rank 1 looked for a checkpoint"
