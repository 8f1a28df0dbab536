"""CPU: the multi-GPU sweep's only collective (all-gather of predicted corners) with world_size 2 on gloo."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from boxdreamer_amd import synth
from boxdreamer_amd.dist import gather_corners, gather_corners_ragged, shard_batch, shard_range


def _worker(rank, world, port, n_total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_total * 16, dtype=torch.float32).reshape(n_total, 8, 2)
        # equal shards
        lo, hi = shard_range(n_total - n_total % world, rank, world)
        eq = gather_corners(full[lo:hi].clone(), world)
        assert torch.equal(eq, full[: n_total - n_total % world])
        # ragged shards
        lo, hi = shard_range(n_total, rank, world)
        rg = gather_corners_ragged(full[lo:hi].clone(), n_total)
        assert torch.equal(rg, full)
        # a sharded batch dict keeps per-sample alignment
        data = synth.make_batch(5, B=4, T=2, size=56)
        mine = shard_batch(data, rank, world)
        marker = mine["bbox_proj_crop"][:, 0].contiguous()                 # (B_local, 8, 2)
        allm = gather_corners(marker, world)
        assert torch.equal(allm, data["bbox_proj_crop"][:, 0])
        # the per-Linear promotion state is rank 0's on every rank (each rank calibrates on its own first batch): stub modules,
        # host logic only
        from types import SimpleNamespace
        from boxdreamer_amd import calibrate
        enc = SimpleNamespace(model=SimpleNamespace(promote=[0, rank * 3], promote_misc=rank, feats_prec=0))
        dec = SimpleNamespace(hip_promote=[rank * 5, 2], hip_promote_misc=0)
        changed = calibrate.sync_state_across_ranks(enc, dec)
        assert changed == (rank != 0)
        assert calibrate.get_state(enc, dec) == {"enc": [0, 0], "enc_misc": 0, "dec": [0, 2], "dec_misc": 0}
        assert calibrate.has_state(enc, dec)
    finally:
        dist.destroy_process_group()


def test_corner_allgather_world2():
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, 7), nprocs=2, join=True)
