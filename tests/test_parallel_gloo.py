"""CPU tier: the N>1 path (voice sharding + one bus sum) with world_size 2 over gloo.
The per-rank buses come from the oracle here (no GPU in this tier); the sharding and
reduce logic is the product's (groove_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, frames, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from groove_amd import parallel, patches as P
    from oracle import oracle as O
    lo, hi = parallel.voice_range(total, rank, world)
    bank = O.Bank.welsh(P.welsh_voices(hi - lo, lo))
    bank.note_events(P.note_on_all(hi - lo, lo))
    bus = bank.render_bus(frames).astype(np.float32)
    red = parallel.reduce_bus_host(dist, bus, root=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), red)
    dist.barrier()
    dist.destroy_process_group()


def test_voice_range_partition():
    from groove_amd.parallel import voice_range
    for total in (1, 7, 256, 1_000_000, 131072):
        for world in (1, 2, 3, 4, 8):
            edges = [voice_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    assert voice_range(131072, 3, 8) == (49152, 65536)  # config #5: 16,384 per GPU


def test_two_rank_sharded_render_matches_single_rank(tmp_path, oracle):
    import torch.multiprocessing as mp
    from groove_amd import patches as P
    total, frames, world = 96, 512, 2
    mp.spawn(_worker, args=(world, _free_port(), total, frames, str(tmp_path)), nprocs=world, join=True)
    red = np.load(tmp_path / "reduced.npy").astype(np.float64)
    full = oracle.Bank.welsh(P.welsh_voices(total))
    full.note_events(P.note_on_all(total))
    want = full.render_bus(frames)
    assert np.max(np.abs(red - want)) / total <= 1e-6  # fp32 partial buses, different sum order
