"""world_size-2 gloo tests of the multi-GPU plumbing (runs on CPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from face_crop_plus_amd import weights
    from face_crop_plus_amd import dist as D
    sd = weights.generate_state_dict("bisenet", seed=0 if rank == 0 else 99)   # ranks start different
    sd = D.broadcast_state_dict(sd)
    ref = weights.generate_state_dict("bisenet", seed=0)
    same = all(torch.equal(sd[k], ref[k]) for k in ref if not k.endswith("num_batches_tracked"))
    batches = [[f"{i}.jpg", f"{i}b.jpg"] for i in range(7)]
    mine = D.shard(batches)
    total = D.all_reduce_scalar(len(mine), "sum")
    tmax = D.all_reduce_scalar(1.0 + rank, "max")
    np.save(os.path.join(tmp, f"r{rank}.npy"), np.array([int(same), len(mine), int(total), int(tmax)] +
                                                       [int(b[0].split(".")[0]) for b in mine]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_broadcast_reduce_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npy")
    r1 = np.load(tmp_path / "r1.npy")
    assert r0[0] == 1 and r1[0] == 1                     # both ranks hold rank 0's weights
    assert r0[2] == 7 and r1[2] == 7 and r0[3] == 2      # sum of shard sizes, max over ranks
    assert sorted(r0[4:].tolist() + r1[4:].tolist()) == list(range(7))   # union of shards == all batches
    assert r0[4:].tolist() == [0, 2, 4, 6] and r1[4:].tolist() == [1, 3, 5]


def test_shard_is_identity_without_process_group():
    from face_crop_plus_amd import dist as D
    assert D.shard([1, 2, 3]) == [1, 2, 3]
    assert D.all_reduce_scalar(5) == 5


def _load_worker(rank, world, port, tmp):
    """The product's loader inside a process group: rank 0 reads the checkpoint, rank 1 must never look for it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      FCP_OFFLINE="1", TORCH_HOME=os.path.join(tmp, "th"))
    os.environ.pop("FCP_WEIGHTS", None)
    from face_crop_plus_amd import weights as W
    touched = []
    if rank == 0:
        os.environ["FCP_WEIGHTS_DIR"] = os.path.join(tmp, "ckpt")
    else:
        os.environ["FCP_WEIGHTS_DIR"] = os.path.join(tmp, "nowhere")
        W.find_checkpoint = lambda m: touched.append(("find", m))          # any lookup / read / download is recorded
        W._fetch_checkpoint = lambda m: touched.append(("fetch", m))
        torch.load = lambda *a, **k: touched.append(("torch.load", a))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = W.load_state_dict("bisenet")                                       # source None = "the real checkpoint"
    ref = W.generate_state_dict("bisenet", seed=7)
    same = all(torch.equal(torch.as_tensor(sd[k]), ref[k]) for k in ref if not k.endswith("num_batches_tracked"))
    # a failure on rank 0 is raised on EVERY rank (nobody hangs in the broadcast)
    try:
        W.load_state_dict("rrdb")
        raised = ""
    except RuntimeError as e:
        raised = str(e)
    # a state dict already in memory is used as it is, no collective
    mem = W.load_state_dict("bisenet", ref)
    np.save(os.path.join(tmp, f"load{rank}.npy"), np.array([int(same), len(touched), int("bsrgan_x4_enhancer.pth" in raised),
                                                            int(mem is ref)]))
    dist.barrier()
    dist.destroy_process_group()


def test_product_loader_reads_on_rank0_only_and_broadcasts(tmp_path):
    """north_star "RCCL broadcast of weights": inside a process group `load_state_dict` (what Cropper / the CLI call
    through the model objects' load()) resolves the checkpoint on rank 0 only."""
    from face_crop_plus_amd import weights as W
    os.makedirs(tmp_path / "ckpt")
    torch.save(W.generate_state_dict("bisenet", seed=7), tmp_path / "ckpt" / "bise_parser.pth")
    world, port = 2, _free_port()
    mp.spawn(_load_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "load0.npy"), np.load(tmp_path / "load1.npy")
    assert r0.tolist() == [1, 0, 1, 1]
    assert r1.tolist() == [1, 0, 1, 1], "rank 1 must receive rank 0's weights without touching any checkpoint source"
