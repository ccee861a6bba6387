"""world_size-2 gloo test of the multi-GPU plumbing (sharding + the label-mask all-gather)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from groundgrid_amd.dist import (all_gather_label_masks, common_stride, owner_of, pack_label_masks, shard_range,
                                 unpack_label_masks)


def test_shard_range_covers_every_cloud_once():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                first, cnt = shard_range(n, r, world)
                seen += list(range(first, first + cnt))
            assert seen == list(range(n))
    assert shard_range(64, 3, 8) == (24, 8)
    assert owner_of(27, 64, 8) == (3, 3)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _expected_labels(cloud, stride):
    rng = np.random.default_rng(1000 + cloud)
    n = stride - (cloud % 5)
    lab = np.zeros(stride, dtype=np.uint8)
    lab[:n] = rng.choice(np.array([0, 49, 99], dtype=np.uint8), size=n)
    return lab, n


def _worker(rank, world, port, n_clouds, stride, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, cnt = shard_range(n_clouds, rank, world)
    labels = torch.zeros((cnt, stride), dtype=torch.uint8)
    counts = torch.zeros((cnt, 4), dtype=torch.int32)
    for i in range(cnt):
        lab, n = _expected_labels(first + i, stride)
        labels[i] = torch.from_numpy(lab)
        counts[i, 0] = n
    g, c = all_gather_label_masks(labels, counts)
    ok = True
    for b in range(n_clouds):
        lab, n = _expected_labels(b, stride)
        ok &= bool(np.array_equal(g[b].numpy(), lab)) and int(c[b, 0]) == n
    # what bench.py does at N > 1: every rank has its own largest cloud, the ranks agree on one stride, and the labels travel
    # as 2-bit masks (a quarter of the bytes)
    local_max = 250 + 7 * rank
    cs = common_stride(local_max)
    ok &= cs == 320 and cs % 4 == 0
    wide = torch.zeros((cnt, cs), dtype=torch.uint8)
    wide[:, :stride] = labels
    gm = all_gather_label_masks(pack_label_masks(wide))
    ok &= tuple(gm.shape) == (n_clouds, cs // 4)
    ok &= bool(torch.equal(unpack_label_masks(gm, stride), g))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, tuple(g.shape)))


def test_all_gather_of_label_masks_world2():
    world, n_clouds, stride = 2, 8, 257
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clouds, stride, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (n_clouds, stride), (rank, ok, shape)


def test_label_mask_packing_round_trip():
    rng = np.random.default_rng(3)
    lab = torch.from_numpy(rng.choice(np.array([0, 49, 99], dtype=np.uint8), size=(5, 64)))
    m = pack_label_masks(lab)
    assert m.shape == (5, 16) and m.dtype == torch.uint8
    assert torch.equal(unpack_label_masks(m), lab)
    assert int(m[0, 0]) == sum(({0: 0, 49: 1, 99: 2}[int(lab[0, k])]) << (2 * k) for k in range(4))


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` without a launcher (the command shape the driver uses for N = 1) must become two ranks that
    rendezvous; --dry-launch stops after the rendezvous + one all-reduce (gloo here, RCCL on a GPU box)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["dry_launch"] and r["ranks"] == 2 and r["all_reduce_ok"]
