"""The N>1 path on CPU: world_size-2 gloo processes shard an image stream, exchange their mAP
records with the single all-gather of fasterrcnn_amd/evaluate.py, and every rank must obtain the
single-process (reference) mAP bit for bit."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from fasterrcnn_amd.datasets.training_sample import Box
from fasterrcnn_amd.evaluate import ImageRecords, merged_calculator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream():
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_ops.npz"))
    gts, preds = g["map_stream_gt"], g["map_stream_pred"]
    out = []
    for i in range(int(gts[:, 0].max()) + 1):
        gt = [Box(int(r[1]), "x", r[2:6].astype(np.float32)) for r in gts[gts[:, 0] == i]]
        p = {c: preds[(preds[:, 0] == i) & (preds[:, 1] == c)][:, 2:7] for c in range(1, 21)}
        out.append((i, p, gt))
    return out, float(g["map_stream_value"])


def _worker(rank, world, port, result_queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stream, _ = _stream()
    rec = ImageRecords()
    for pos, (idx, p, gt) in enumerate(stream):
        if pos % world == rank:
            rec.add(idx, p, gt)
    calc = merged_calculator(rec)
    result_queue.put((rank, float(calc.compute_mean_average_precision())))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_sharded_map_equals_single_process():
    stream, expected = _stream()
    # single process, no process group
    rec = ImageRecords()
    for idx, p, gt in stream:
        rec.add(idx, p, gt)
    assert float(merged_calculator(rec).compute_mean_average_precision()) == expected

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: expected, 1: expected}


# ---- data-parallel training: the gradient exchange of fasterrcnn_amd/training.py ---------------------------------
def _grad_worker(rank, world, port, result_queue):
    import torch
    from fasterrcnn_amd import training
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    shapes = [("head", (16, 32)), ("fc2", (64, 64)), ("fc1", (64, 100)), ("rpn_head", (16, 8)), ("rpn_conv", (9, 4, 4)), ("conv12", (9, 8, 4)),
              ("conv11", (9, 8, 8))]
    mine = {k: torch.randn(shp, generator=g) for k, shp in shapes}
    # (a) the overlapped form train_step uses: tensors handed over one by one in production order; fc1 / fc2 (>= direct_bytes) are
    #     reduced in place, the rest coalesced into 2 KB buckets
    avg = training.GradientAverager(bucket_bytes=2 * 1024, direct_bytes=16 * 1024)
    grads = avg.track()
    for k, _ in shapes:
        grads[k] = mine[k].clone()
    avg.finish()
    nmsg = avg.messages
    # (b) the one-shot form
    again = {k: v.clone() for k, v in mine.items()}
    training.GradientAverager(bucket_bytes=8 * 1024, direct_bytes=1 << 30)(again)
    result_queue.put((rank, nmsg, {k: v.numpy() for k, v in mine.items()}, {k: v.numpy() for k, v in grads.items()},
                      {k: v.numpy() for k, v in again.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_averager_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, nmsg, mine, out, again = q.get(timeout=120)
        res[rank] = (nmsg, mine, out, again)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] and res[0][0] >= 4                 # same messages on both ranks: 2 in place + several buckets
    for k in res[0][1]:
        want = (res[0][1][k] + res[1][1][k]) / 2.0
        for form in (2, 3):
            assert np.array_equal(res[0][form][k], res[1][form][k])  # every rank ends with the same gradients
            assert np.allclose(res[0][form][k], want, rtol=1e-6, atol=1e-7)
