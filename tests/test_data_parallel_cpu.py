"""world_size-2 gloo test of the bucketed gradient reducer: after backward +
finish() every rank holds mean_r(g_r), parameters were broadcast from rank 0,
and several buckets are exercised (the all-reduce is checked as
mean_r(g_r))."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(20, 64), nn.GELU(), nn.Linear(64, 64), nn.GELU(), nn.Linear(64, 5))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(8, 20, generator=g), torch.randint(0, 5, (8,), generator=g)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basd_amd.training.data_parallel import GradientReducer
    from basd_amd.training.optim import FlatParams
    model = _model(seed=rank)                       # different init per rank: broadcast must fix it
    flat = FlatParams(list(model.parameters()))
    reducer = GradientReducer(flat, bucket_bytes=1024)   # several buckets
    assert len(reducer.buckets) >= 3
    reducer.broadcast_parameters(0)
    for _ in range(2):                              # two steps: pending counters must reset
        flat.zero_grad()
        x, y = _data(rank)
        nn.functional.cross_entropy(model(x), y).backward()
        reducer.finish()
    torch.save({"grad": flat.grad.clone(), "data": flat.data.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_mean(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    torch.testing.assert_close(r0["grad"], r1["grad"], atol=0, rtol=0)
    torch.testing.assert_close(r0["data"], r1["data"], atol=0, rtol=0)
    # single-process reference: same weights (rank 0's), mean of the two shard gradients
    from basd_amd.training.optim import FlatParams
    model = _model(seed=0)
    flat = FlatParams(list(model.parameters()))
    torch.testing.assert_close(flat.data, r0["data"])
    want = torch.zeros_like(flat.grad)
    for rank in range(2):
        flat.zero_grad()
        x, y = _data(rank)
        nn.functional.cross_entropy(model(x), y).backward()
        want += flat.grad / 2
    torch.testing.assert_close(r0["grad"], want, atol=1e-7, rtol=1e-6)


def test_reducer_is_a_noop_without_process_group():
    from basd_amd.training.data_parallel import GradientReducer
    from basd_amd.training.optim import FlatParams
    model = _model(0)
    flat = FlatParams(list(model.parameters()))
    red = GradientReducer(flat)
    assert not red.enabled
    x, y = _data(0)
    nn.functional.cross_entropy(model(x), y).backward()
    g = flat.grad.clone()
    red.finish()
    torch.testing.assert_close(flat.grad, g, atol=0, rtol=0)
