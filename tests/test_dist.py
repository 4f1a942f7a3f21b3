"""CPU, world_size 2 over gloo: the multi-GPU driver's sharding and loss gather (the only collective on the path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from topo4d_amd import dist as t4d_dist


def test_shard_units_partition():
    for n, w in ((24, 8), (24, 5), (64, 8), (3, 8), (1536, 8)):
        seen = []
        for r in range(w):
            s = t4d_dist.shard_units(n, r, w)
            assert s == list(range(r, n, w))
            seen += s
        assert sorted(seen) == list(range(n))
        assert t4d_dist.shard_sizes(n, w) == [len(t4d_dist.shard_units(n, r, w)) for r in range(w)]
    assert t4d_dist.shard_sizes(24, 8) == [3] * 8
    with pytest.raises(ValueError):
        t4d_dist.shard_units(24, 8, 8)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_units):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = t4d_dist.shard_units(n_units, rank, world)
        # the "loss" of unit u is a known function of u, so every rank can check the gathered vector
        local = torch.tensor([float(u) * 0.5 + 1.0 for u in mine])
        allv = t4d_dist.gather_losses(local, n_units=n_units)
        assert torch.equal(allv, torch.arange(n_units).float() * 0.5 + 1.0), (rank, allv)
        if n_units % world == 0:
            allv2 = t4d_dist.gather_losses(local)                  # equal-shard fast path
            assert torch.equal(allv2, allv)
        g = [torch.full((5, 3), float(rank + 1)), torch.full((7,), float(rank + 1))]
        t4d_dist.all_reduce_grads(g)
        tot = float(sum(range(1, world + 1)))
        assert torch.all(g[0] == tot) and torch.all(g[1] == tot)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [24, 7])
def test_gather_losses_world2_gloo(n_units):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_units), nprocs=2, join=True)


def test_single_process_gather_is_identity():
    x = torch.arange(5.0)
    assert t4d_dist.gather_losses(x) is x


def _worker_async(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = torch.tensor([rank * 10.0 + k for k in range(3)])
        out = torch.empty(3 * world)
        out, work = t4d_dist.gather_losses_async(local, out)
        if work is not None:
            work.wait()
        assert torch.equal(out.view(world, -1), torch.tensor([[r * 10.0 + k for k in range(3)] for r in range(world)]))
    finally:
        dist.destroy_process_group()


def test_async_gather_world2_gloo():
    mp.spawn(_worker_async, args=(2, _free_port()), nprocs=2, join=True)


def test_band_bounds_partition_the_rows():
    for n_rows, world in ((8192, 8), (1024, 3), (7, 8), (5, 1)):
        bounds = [t4d_dist.band_bounds(n_rows, r, world) for r in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == n_rows
        assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
        heights = [b - a for a, b in bounds]
        assert max(heights) - min(heights) <= 1
    with pytest.raises(ValueError):
        t4d_dist.band_bounds(10, 2, 2)


def _worker_bands(rank, world, port, n_rows):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_rows * 4 * 3, dtype=torch.float32).reshape(n_rows, 4, 3)      # what a 1-GPU bake would produce
        b0, b1 = t4d_dist.band_bounds(n_rows, rank, world)
        got = t4d_dist.gather_bands(full[b0:b1].contiguous(), n_rows)                       # texture.bake_texture_sharded's exchange
        assert torch.equal(got, full), rank
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [16, 7])
def test_gather_bands_world2_gloo(n_rows):
    mp.spawn(_worker_bands, args=(2, _free_port(), n_rows), nprocs=2, join=True)
