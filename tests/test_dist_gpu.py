"""BASELINE configs[2] as far as one GPU allows: two processes share GPU 0, 32 of the 64 clouds each (shard_range), each runs
filter_batch(want_masks=True) on its fresh maps, the 2-bit label masks are all-gathered (gloo: RCCL refuses two ranks on one
device), and EVERY one of the 64 gathered masks is compared with the CPU oracle's labels for that cloud."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_CLOUDS, N_AZ = 64, 300  # ~17 k points per cloud: 64 oracle runs stay within seconds


def cloud_of(b):
    from groundgrid_amd import synth

    base = synth.hdl64_cloud(seed=500 + b % 8, n_az=N_AZ)
    ang = np.float32(0.37 * (b // 8))
    c, s = np.cos(ang), np.sin(ang)
    out = synth.clone_cloud(base)
    out["x"] = (c * base["x"] - s * base["y"]).astype(np.float32)
    out["y"] = (s * base["x"] + c * base["y"]).astype(np.float32)
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, stride, q):
    import torch
    import torch.distributed as dist

    from groundgrid_amd import api
    from groundgrid_amd.dist import all_gather_label_masks, shard_range, unpack_label_masks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    first, cnt = shard_range(N_CLOUDS, rank, world)
    clouds = [cloud_of(first + i) for i in range(cnt)]
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=cnt, max_points=stride, device=0)
    host = np.zeros((cnt, stride), dtype=api.POINT16_DTYPE)
    for i, c in enumerate(clouds):
        host[i, : len(c)] = api.pack16(c)
    pts = torch.from_numpy(host.view(np.uint8).reshape(cnt, stride, 16)).cuda()
    out = None
    for _ in range(2):  # second frame: the persistent map state of every slot is in play
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((cnt, 3), np.float32), np.full(cnt, -1.73), out=out, want_masks=True)
    torch.cuda.synchronize()
    gathered = all_gather_label_masks(out.label_masks.cpu())  # [64, stride / 4] on every rank
    labels = unpack_label_masks(gathered, stride).numpy()
    dist.barrier()
    dist.destroy_process_group()
    seg.close()
    q.put((rank, labels if rank == 0 else None, tuple(gathered.shape)))


def test_64_clouds_on_two_ranks_all_gather_of_masks_matches_the_oracle():
    import torch.multiprocessing as mp

    from oracle import oracle

    clouds = [cloud_of(b) for b in range(N_CLOUDS)]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, stride, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    labels = [r[1] for r in res if r[1] is not None][0]
    assert all(r[2] == (N_CLOUDS, stride // 4) for r in res)
    for b, c in enumerate(clouds):
        ref = oracle.OracleMap(120.0, 0.33)
        for _ in range(2):
            r = ref.filter_cloud(c, (0.0, 0.0, 0.0), -1.73)
        assert np.array_equal(labels[b, : len(c)], r["label"]), f"cloud {b}: gathered mask differs from the oracle's labels"
        assert not labels[b, len(c):].any()


# ---------------------------------------------------------------- more than one GPU: RCCL proper (skipped on a one-GPU box)

def _visible_gpus():
    try:
        import torch

        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def _rccl_worker(rank, world, port, stride, via, q):
    """One rank = one process = one GPU (device `rank`): its 64 / world clouds on fresh maps, two frames, then the all-gather of
    the 2-bit masks over RCCL -- through torch.distributed (backend nccl) or through the library's own C entry point."""
    import torch
    import torch.distributed as dist

    from groundgrid_amd import api
    from groundgrid_amd.dist import AbiLabelGather, shard_range, unpack_label_masks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    first, cnt = shard_range(N_CLOUDS, rank, world)
    clouds = [cloud_of(first + i) for i in range(cnt)]
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=max(cnt, 1), max_points=stride, device=rank)
    host = np.zeros((max(cnt, 1), stride), dtype=api.POINT16_DTYPE)
    for i, c in enumerate(clouds):
        host[i, : len(c)] = api.pack16(c)
    pts = torch.from_numpy(host.view(np.uint8).reshape(host.shape[0], stride, 16)).to(dev)[:cnt].contiguous()
    out = None
    for _ in range(2):
        out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((cnt, 3), np.float32), np.full(cnt, -1.73), out=out, want_masks=True)
    if via == "abi":
        gather = AbiLabelGather(seg, rank, world)
        gathered = gather.gather(out.label_masks)
        torch.cuda.synchronize(dev)
        gather.close()
    else:
        gathered = torch.empty((world * cnt, stride // 4), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, out.label_masks)
        torch.cuda.synchronize(dev)
    info = {"rank": rank, "backend": dist.get_backend(), "ranks": dist.get_world_size(), "device": torch.cuda.current_device(),
            "shape": tuple(gathered.shape)}
    labels = unpack_label_masks(gathered, stride).cpu().numpy()
    dist.barrier()
    dist.destroy_process_group()
    seg.close()
    q.put((info, labels if rank == world - 1 else None))  # (the LAST rank's copy: every rank must hold every cloud)


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs at least two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("world", sorted({2, min(8, max(2, _visible_gpus()))}))
@pytest.mark.parametrize("via", ["torch", "abi"])
def test_config3_sharded_over_rccl_ranks_one_gpu_each(world, via):
    """BASELINE configs[2] proper: 64 clouds sharded 64 / world per GPU (cloud b -> rank b // per_rank), one process per GPU, the
    all-gather of the label masks over RCCL / xGMI.  Lights up by itself wherever two or more GPUs are visible (2 ranks, and as
    many as there are up to 8); every one of the 64 gathered masks, as the LAST rank holds them, against the oracle."""
    import torch.multiprocessing as mp

    from oracle import oracle

    if N_CLOUDS % world:
        pytest.skip("64 clouds do not shard evenly")
    clouds = [cloud_of(b) for b in range(N_CLOUDS)]
    stride = (max(len(c) for c in clouds) + 63) // 64 * 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, stride, via, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for info, _ in res:
        assert info["backend"] == "nccl" and info["ranks"] == world and info["device"] == info["rank"], info
        assert info["shape"] == (N_CLOUDS, stride // 4), info
    labels = [r[1] for r in res if r[1] is not None][0]
    for b, c in enumerate(clouds):
        ref = oracle.OracleMap(120.0, 0.33)
        for _ in range(2):
            r = ref.filter_cloud(c, (0.0, 0.0, 0.0), -1.73)
        assert np.array_equal(labels[b, : len(c)], r["label"]), f"cloud {b}: gathered mask differs from the oracle's labels"
        assert not labels[b, len(c):].any()


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs at least two GPUs")
def test_bench_scales_to_two_gpus_over_rccl():
    """`python bench.py --gpus 2` (it launches itself under torch.distributed.run): RCCL sees two ranks, the per-step all-gather
    overlaps the next step, configs[2] is sharded 32 + 32, and the line carries n1_equivalent for comparison with the N = 1 run."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["collective"]["rccl_ranks"] == 2 and r["collective"]["backend"] == "nccl"
    assert r["config3"]["gathered_masks_match_labels"] is True and r["config3"]["clouds_per_gpu"] == 32
    assert abs(r["summary"]["n1_equivalent"] * 2 - r["value"]) < 1e-3 * r["value"]


def test_allgather_through_the_c_abi_single_rank():
    """gg_comm_unique_id -> gg_comm_init_rank -> gg_allgather_label_masks on a one-rank RCCL communicator (all this box has):
    the gathered buffer equals the masks k_label wrote, on the torch stream and on the context's own stream."""
    import numpy as np
    import torch

    from groundgrid_amd import api, synth
    from groundgrid_amd.dist import AbiLabelGather, unpack_label_masks

    clouds = [synth.hdl64_cloud(seed=700 + k, n_az=180 + 20 * k) for k in range(3)]
    B, stride = len(clouds), (max(len(c) for c in clouds) + 63) // 64 * 64
    seg = api.GroundSegmentation().init(120.0, 0.33, n_slots=B, max_points=stride)
    host = np.zeros((B, stride), dtype=api.POINT16_DTYPE)
    for b, c in enumerate(clouds):
        host[b, : len(c)] = api.pack16(c)
    pts = torch.from_numpy(host.view(np.uint8).reshape(B, stride, 16)).cuda()
    gather = AbiLabelGather(seg, rank=0, world=1)
    out = seg.filter_batch(pts, [len(c) for c in clouds], np.zeros((B, 3), np.float32), np.full(B, -1.73), want_masks=True)
    got = gather.gather(out.label_masks)
    torch.cuda.synchronize()
    assert torch.equal(got, out.label_masks)
    lab = unpack_label_masks(got, stride)
    for b, c in enumerate(clouds):
        assert torch.equal(lab[b, : len(c)], out.labels[b, : len(c)])
    side = torch.cuda.Stream()
    got2 = gather.gather(out.label_masks, stream=side.cuda_stream)
    side.synchronize()
    assert torch.equal(got2, out.label_masks)
    gather.close()
    seg.close()


@pytest.mark.parametrize("abi", [False, True])
def test_bench_multi_gpu_code_path_with_one_rank(abi):
    """bench.py's N > 1 code path on the one GPU this box has (--force-dist: an RCCL process group of one rank): the per-step
    all-gather of the label masks overlapped with the next step (torch.distributed, or the library's own C entry point with
    --abi-collective), the configs[2] leg with the gathered masks held against the labels, and the timed batch's oracle check --
    everything the 8-GPU run executes except a second rank."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--batch", "48", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1",
           "--config4-batch", "2"] + (["--abi-collective"] if abi else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["collective"]["rccl_ranks"] == 1 and r["collective"]["backend"] == "nccl"
    assert ("C ABI" in r["collective"]["issued_by"]) == abi
    assert r["parity_checked_in_run"] is True and r["warm_map"]["parity_checked_in_run"] is True
    assert r["config3"]["gathered_masks_match_labels"] is True and r["config3"]["parity_checked_in_run"] is True
    assert r["config4"]["parity_checked_in_run"] is True and r["config4"]["single_cloud"]["parity_checked_in_run"] is True
