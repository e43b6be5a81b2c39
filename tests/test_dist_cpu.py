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
