"""Multi-GPU plumbing for the sharded renderer.

The reference has no distributed code at all; this is the MI355X-side design:
  * the world LOD chain is replicated on every GPU (read-only, < 0.5 GB at 2048^3),
  * every frame's 64-ray tiles are dealt round-robin to the ranks (cvx_set_shard),
  * after rendering, the tiles of frame f are sent to its display rank f % N.
The exchange is a set of direct peer-to-peer sends (torch.distributed batch_isend_irecv =
grouped ncclSend/ncclRecv on RCCL), one message per (source, destination) pair and raybuffer kind, so on
MI355X each pair rides its own xGMI link and no ring is formed.  torch is used for device memory and the
collective only; tensors may be CPU tensors with the gloo backend (tests).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist

TILE_RAYS = 64


@dataclass
class Pools:
    """Tile pools of all buffers of one context: td[G * tilesTD, H * 64], lr[G * tilesLR, W * 64] (int32 = ARGB32 bits)."""

    td: torch.Tensor
    lr: torch.Tensor
    tiles_td: int
    tiles_lr: int


def tile_capacity(width: int, height: int):
    """Tiles per buffer as cvx_set_resolution allocates them (RenderManager.cs:35-36 capacities in whole tiles + 2)."""
    return (width + 2 * height + TILE_RAYS - 1) // TILE_RAYS + 2, (2 * width + height + TILE_RAYS - 1) // TILE_RAYS + 2


def allocate_pools(buffer_count: int, lay_td, lay_lr, device) -> Pools:
    td = torch.zeros((buffer_count * lay_td.tileCapacity, lay_td.width * TILE_RAYS), dtype=torch.int32, device=device)
    lr = torch.zeros((buffer_count * lay_lr.tileCapacity, lay_lr.width * TILE_RAYS), dtype=torch.int32, device=device)
    return Pools(td, lr, lay_td.tileCapacity, lay_lr.tileCapacity)


def frame_tiles(ray_counts):
    """Tiles of one frame in the order libcpuvox_gpu numbers them (segment-major): list of (kind, tile index in
    the buffer's pool); kind 0 = top-down pool (segments 0,1), 1 = left-right pool (segments 2,3)."""
    out = []
    base = [0, 0]
    for s in range(4):
        kind = 0 if s < 2 else 1
        if s == 2:
            base[1] = 0
        n = (max(0, ray_counts[s]) + TILE_RAYS - 1) // TILE_RAYS
        first = base[kind]
        for t in range(n):
            out.append((kind, first + t))
        base[kind] = first + n
    return out


class TileExchange:
    """Precomputed exchange for one batch of frames: rank r rendered the tiles t of every frame with
    t % N == r into buffer b = frame index; afterwards frame f must be complete on rank f % N."""

    def __init__(self, frames, width: int, height: int, rank: int, world_size: int, pools: Pools, device):
        self.rank, self.N, self.pools = rank, world_size, pools
        N = world_size
        send = [[[], []] for _ in range(N)]  # [dest][kind] -> pool rows
        recv = [[[], []] for _ in range(N)]  # [src][kind]
        caps = (pools.tiles_td, pools.tiles_lr)
        for b, fr in enumerate(frames):
            rc = [s.RayCount for s in fr.segments] if hasattr(fr, "segments") else list(fr)
            root = b % N
            for t, (kind, tile) in enumerate(frame_tiles(rc)):
                owner = t % N
                row = b * caps[kind] + tile
                if owner == rank and root != rank:
                    send[root][kind].append(row)
                elif owner != rank and root == rank:
                    recv[owner][kind].append(row)
        self.ops_spec = []
        for kind, pool in ((0, pools.td), (1, pools.lr)):
            for peer in range(N):
                if peer == rank:
                    continue
                s_rows = torch.tensor(send[peer][kind], dtype=torch.long, device=device)
                r_rows = torch.tensor(recv[peer][kind], dtype=torch.long, device=device)
                s_buf = torch.empty((len(send[peer][kind]), pool.shape[1]), dtype=pool.dtype, device=device) if len(send[peer][kind]) else None
                r_buf = torch.empty((len(recv[peer][kind]), pool.shape[1]), dtype=pool.dtype, device=device) if len(recv[peer][kind]) else None
                self.ops_spec.append((pool, peer, s_rows, s_buf, r_rows, r_buf))
        self.sent_rows = sum(len(send[p][k]) for p in range(N) for k in range(2))
        self.recv_rows = sum(len(recv[p][k]) for p in range(N) for k in range(2))

    def run(self) -> None:
        """Pack my tiles per destination, exchange, scatter the received tiles into their pool rows."""
        ops = []
        for pool, peer, s_rows, s_buf, r_rows, r_buf in self.ops_spec:
            if s_buf is not None:
                torch.index_select(pool, 0, s_rows, out=s_buf)
                ops.append(dist.P2POp(dist.isend, s_buf, peer))
            if r_buf is not None:
                ops.append(dist.P2POp(dist.irecv, r_buf, peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for pool, peer, s_rows, s_buf, r_rows, r_buf in self.ops_spec:
            if r_buf is not None:
                pool.index_copy_(0, r_rows, r_buf)
