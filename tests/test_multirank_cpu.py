"""Two-rank CPU test (gloo) of the keyframe-pair sharding and the chunked gather.

Each rank computes its share of the pairs with the CPU oracle standing in for the
HIP kernel (the result format is the kernel's: distance << 16 | train index), the
shares are gathered to rank 0 exactly as bench.py does on the GPUs, and rank 0
compares with the single-process result.
"""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _pair_rows(oracle, desc, n_per, pair_list):
    from slam_amd import synth
    out = np.full((len(pair_list), n_per), -1, np.int32)
    kp = synth.random_keypoints(n_per, seed=1)
    for r, (i, j) in enumerate(pair_list):
        m = oracle.match_gated(desc[i], kp, desc[j], kp, 300.0, -1.0, -1)
        out[r, m["queryIdx"]] = (m["distance"].astype(np.int32) << 16) | m["trainIdx"]
    return out


def _worker(rank, world, port, n_kf, n_per, chunks, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import orb_oracle as oracle
    from slam_amd import pairs as pairs_mod, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc = synth.random_descriptors(n_kf * n_per, seed=3).reshape(n_kf, n_per, 32)
    plist = pairs_mod.all_pairs(n_kf)
    pg = pairs_mod.PairGather(torch, dist, len(plist), n_per, world, rank, torch.device("cpu"), chunks)

    def compute(a, b):
        rows = _pair_rows(oracle, desc, n_per, plist[pg.lo + a:pg.lo + b].tolist())
        pg.res[a:b] = torch.from_numpy(rows)

    pg.run(compute)
    if rank == 0:
        q.put(pg.result().numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_kf,chunks", [(2, 6, 3), (2, 5, 4), (3, 4, 2)])
def test_pair_gather_matches_single_process(oracle, world, n_kf, chunks):
    import torch.multiprocessing as mp
    from slam_amd import pairs as pairs_mod, synth
    n_per = 24
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_kf, n_per, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    desc = synth.random_descriptors(n_kf * n_per, seed=3).reshape(n_kf, n_per, 32)
    ref = _pair_rows(oracle, desc, n_per, pairs_mod.all_pairs(n_kf).tolist())
    assert got.shape == ref.shape and np.array_equal(got, ref)
