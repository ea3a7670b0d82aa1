"""Image-tile sharding across GPUs and the framebuffer gather.

The reference is single-device (no communication layer at all).  Pixels are
independent, so the N-GPU path shards the film and exchanges data exactly once per
readback:

  * partition: columns x are grouped in bands of `band_width`; band b belongs to rank
    b % world_size (interleaved for load balance: cost varies across the image).  Every rank
    holds the whole scene; the RNG is keyed by the GLOBAL pixel index, so the assembled
    image is identical to the single-GPU image bit for bit, whatever the partition.
  * gather: one `all_gather` of the per-rank (n_cols, H, 3) float32 tiles over
    torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests),
    ~1-2 MB per rank — latency-bound, not link-bound.  There is no other collective on the path.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

__all__ = ["TilePlan", "gather_tiles", "gather_tiles_device", "gather_image", "assemble", "device_tile"]


class TilePlan:
    """Pure index arithmetic; mirrored on the device side by `local_to_global` in csrc/stages.hpp."""

    def __init__(self, width: int, height: int, band_width: int, world_size: int):
        if width <= 0 or height <= 0 or band_width <= 0 or world_size <= 0:
            raise ValueError("TilePlan: all arguments must be positive")
        self.width, self.height, self.band_width, self.world_size = int(width), int(height), int(band_width), int(world_size)

    def owner(self, x: int) -> int:
        return (x // self.band_width) % self.world_size

    def columns(self, rank: int) -> np.ndarray:
        """Global column index of every local column of `rank`, in local order."""
        x = np.arange(self.width)
        return x[(x // self.band_width) % self.world_size == rank]

    def n_cols(self, rank: int) -> int:
        return int(len(self.columns(rank)))

    def max_cols(self) -> int:
        return max(self.n_cols(r) for r in range(self.world_size))

    def local_to_global(self, rank: int, lc: int) -> int:
        lb, w = divmod(lc, self.band_width)
        return (lb * self.world_size + rank) * self.band_width + w


def assemble(plan: TilePlan, tiles) -> np.ndarray:
    """tiles[r] = (>= n_cols(r), H, 3) array of rank r  ->  (W, H, 3) image."""
    img = np.zeros((plan.width, plan.height, 3), np.float32)
    for r in range(plan.world_size):
        cols = plan.columns(r)
        img[cols] = np.asarray(tiles[r])[:len(cols)]
    return img


class _DevView:
    """Zero-copy torch view of a raw device pointer through __cuda_array_interface__."""

    def __init__(self, ptr: int, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def gather_tiles_device(tile, plan: TilePlan, world_size: int, group=None):
    """The collective alone: all_gather of this rank's (n_cols, H, 3) torch tensor (padded to the widest rank) into a
    (world_size, max_cols, H, 3) tensor on the same device.  No host copy, no synchronisation beyond the collective itself:
    this is what a render loop calls per step; `assemble` turns the result into the (W, H, 3) image once, at the end."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("gather_tiles_device: torch.distributed is not initialised")
    mc = plan.max_cols()
    if tile.shape[0] == mc:
        padded = tile.contiguous()
    else:
        padded = torch.zeros((mc, plan.height, 3), dtype=torch.float32, device=tile.device)
        padded[:tile.shape[0]] = tile
    out = torch.empty((world_size * mc, plan.height, 3), dtype=torch.float32, device=tile.device)    # rank-major concatenation
    dist.all_gather_into_tensor(out, padded, group=group)
    return out.view(world_size, mc, plan.height, 3)


def gather_tiles(tile, plan: TilePlan, rank: int, world_size: int, group=None, force_collective: bool = False) -> np.ndarray:
    """All-gather the per-rank tiles and assemble the full image (returned on every rank).

    `tile` is this rank's (n_cols, H, 3) float32 data: a numpy array (CPU / gloo) or a torch
    tensor already on the GPU (nccl).  A single rank needs no collective unless `force_collective`
    (used to exercise the RCCL path on a one-GPU box)."""
    if world_size == 1 and not force_collective:
        t = tile.detach().cpu().numpy() if hasattr(tile, "detach") else np.asarray(tile)
        return assemble(plan, [t])
    import torch
    t = tile if isinstance(tile, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(tile, np.float32))
    return assemble(plan, gather_tiles_device(t, plan, world_size, group).cpu().numpy())


_warned_staged = [False]


def device_tile(rdr):
    """This rank's accumulation tile as a torch tensor on its GPU: a zero-copy view of the renderer's framebuffer
    (apt_device_ptr) through __cuda_array_interface__, cloned on the device.  If this torch build does not take a raw pointer
    that way (TypeError / ValueError / RuntimeError from `as_tensor`), the tile is staged through the host instead - slower by
    one D2H + H2D of the tile, and said so once, loudly, rather than silently."""
    import torch
    dev = torch.device(f"cuda:{rdr.device}")
    try:
        view = _DevView(rdr.device_accum_ptr(), (rdr.n_cols, rdr.h, 3))
        return torch.as_tensor(view, device=dev).clone()
    except (TypeError, ValueError, RuntimeError) as e:
        if not _warned_staged[0]:
            import warnings
            warnings.warn(f"adapt_amd.tiles.device_tile: zero-copy view of the device framebuffer failed ({type(e).__name__}: {e}); "
                          "staging the tile through host memory", RuntimeWarning)
            _warned_staged[0] = True
        return torch.from_numpy(rdr.tile_accum()).to(dev)


def gather_image(rdr, normalised: bool = True, group=None, force_collective: bool = False) -> np.ndarray:
    """Full (W, H, 3) image from a sharded `Renderer` (every rank must call)."""
    rdr.synchronize()
    tile: Optional[object] = None
    if rdr.world_size > 1 or force_collective:
        try:
            import torch
            if torch.cuda.is_available():
                tile = device_tile(rdr)
        except ImportError:
            tile = None
    if tile is None:
        tile = rdr.tile_accum()
    img = gather_tiles(tile, rdr.plan, rdr.rank, rdr.world_size, group, force_collective)
    if normalised and rdr._cnt > 0:
        img = img / np.float32(rdr._cnt)
    return img
