"""Image-tile sharding: partition arithmetic for 1/2/4/8 ranks and the gather over torch.distributed
(gloo, world_size 2, on CPU).  The N-GPU image must equal the 1-GPU image bit for bit because the RNG is
keyed by the GLOBAL pixel index — checked here with the CPU oracle standing in for the per-rank renderers."""
import os
import socket

import numpy as np
import pytest

from adapt_amd.tiles import TilePlan, assemble, gather_tiles


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("w,h,bw", [(512, 512, 32), (800, 800, 32), (1280, 720, 32), (100, 7, 16), (37, 5, 8), (64, 64, 64)])
def test_partition_is_a_disjoint_cover(world, w, h, bw):
    plan = TilePlan(w, h, bw, world)
    cols = [plan.columns(r) for r in range(world)]
    allc = np.concatenate(cols)
    assert sorted(allc.tolist()) == list(range(w))                       # union = image, no column twice
    for r, c in enumerate(cols):
        assert all(plan.owner(int(x)) == r for x in c)
        # the device-side formula (csrc/stages.hpp local_to_global) enumerates exactly these columns, in this order
        assert [plan.local_to_global(r, lc) for lc in range(len(c))] == c.tolist()
        assert np.all(np.diff(c) > 0)
    if w % (bw * world) == 0:
        assert len({len(c) for c in cols}) == 1                          # balanced when the film divides evenly
    assert plan.max_cols() == max(len(c) for c in cols)


def test_assemble_round_trip():
    plan = TilePlan(96, 5, 16, 4)
    img = np.random.RandomState(0).rand(96, 5, 3).astype(np.float32)
    tiles = [np.concatenate([img[plan.columns(r)], np.full((3, 5, 3), np.nan, np.float32)]) for r in range(4)]   # padded tiles
    assert np.array_equal(assemble(plan, tiles), img)
    with pytest.raises(ValueError):
        TilePlan(0, 5, 16, 4)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, w, h, bw, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
        from adapt_amd.parsers import scene_parsing
        from adapt_amd.scene_pack import make_config, pack_scene
        from oracle import binding as ob
        root = os.path.join(os.path.dirname(__file__), "..", "scenes", "cbox")
        parsed = scene_parsing(root, "c2_cbox.xml")
        rc = make_config(parsed[3], width=w, height=h, max_bounce=3)
        full, _, _ = ob.OracleScene(pack_scene(*parsed), rc.cam_t).render(rc, 2, threads=1)
        plan = TilePlan(w, h, bw, world)
        tile = full[plan.columns(rank)]           # what rank `rank` renders: its own columns, global-pixel RNG keys
        img = gather_tiles(tile, plan, rank, world)
        np.save(os.path.join(out_dir, f"img{rank}.npy"), img)
        if rank == 0:
            np.save(os.path.join(out_dir, "full.npy"), full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h,bw", [(40, 12, 8), (36, 10, 8)])
def test_gloo_gather_two_ranks(tmp_path, w, h, bw):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, w, h, bw, str(tmp_path)), nprocs=2, join=True)
    full = np.load(tmp_path / "full.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"img{r}.npy"), full)      # every rank ends with the 1-GPU image


def test_single_rank_gather_needs_no_process_group():
    plan = TilePlan(16, 4, 16, 1)
    tile = np.arange(16 * 4 * 3, dtype=np.float32).reshape(16, 4, 3)
    assert np.array_equal(gather_tiles(tile, plan, 0, 1), tile)
    with pytest.raises(RuntimeError):
        gather_tiles(tile[:8], TilePlan(16, 4, 8, 2), 0, 2)
