"""Multi-GPU plumbing for the sharded renderer.

The reference has no distributed code at all; this is the MI355X-side design:
  * the world LOD chain is replicated on every GPU (read-only, < 1 GB at 2048^3),
  * every frame's 64-ray tiles are dealt round-robin to the ranks (cvx_set_shard),
  * after rendering, the tiles of frame f are sent to its display rank f % N.
The exchange is a set of direct peer-to-peer transfers (torch.distributed batch_isend_irecv = grouped
ncclSend/ncclRecv on RCCL), one message per (source, destination) pair, so on MI355X each pair rides its own
xGMI link and no ring is formed.  Only the pixel rows a segment can write ([origMin, origMax] of every tile) travel:
they are packed into one staging buffer per step by a HIP kernel of libcpuvox_gpu (cvx_copy_rows) and unpacked on
the receiving side.  torch is used for device memory and the collective only; with CPU tensors (gloo, tests) the
pack / unpack falls back to numpy.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

TILE_RAYS = 64

# == cvx_row_span (include/cpuvox_gpu.h)
SPAN_DTYPE = np.dtype([("poolRow", "<i8"), ("packedRow", "<i8"), ("rows", "<i4"), ("kind", "<i4")])


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
    the buffer's pool, segment); kind 0 = top-down pool (segments 0,1), 1 = left-right pool (segments 2,3)."""
    out = []
    base = [0, 0]
    for s in range(4):
        kind = 0 if s < 2 else 1
        n = (max(0, ray_counts[s]) + TILE_RAYS - 1) // TILE_RAYS
        first = base[kind]
        for t in range(n):
            out.append((kind, first + t, s))
        base[kind] = first + n
    return out


def segment_pixel_ranges(vp, width: int, height: int):
    """[originalNextFreePixelMin, Max] of the four segments (RenderManager.cs:298-316; Mathf.RoundToInt = half to even)."""
    def rnd(v, hi):
        r = float(np.rint(np.float32(v)))
        if not np.isfinite(r):
            r = -2147483648.0
        return int(min(max(r, 0), hi))

    vx, vy = rnd(vp[0], width - 1), rnd(vp[1], height - 1)
    return [(vy, height - 1), (0, vy), (vx, width - 1), (0, vx)]


class TileExchange:
    """Precomputed exchange for one batch of frames: rank r rendered the tiles t of every frame with
    t % N == r into buffer b = frame index; afterwards frame f must be complete on rank f % N."""

    def __init__(self, frames, width: int, height: int, rank: int, world_size: int, pools: Pools, device, ctx=None):
        self.rank, self.N, self.pools, self.ctx = rank, world_size, pools, ctx
        self.device = torch.device(device)
        N = world_size
        caps = (pools.tiles_td, pools.tiles_lr)
        col = (height, width)
        send = [[] for _ in range(N)]  # per peer: (kind, poolRow, rows)
        recv = [[] for _ in range(N)]
        for b, fr in enumerate(frames):
            if hasattr(fr, "segments"):
                rc = [s.RayCount for s in fr.segments]
                ranges = segment_pixel_ranges(fr.vanishingPointScreenSpace, width, height)
            else:  # (rayCounts, vanishingPoint) tuples
                rc, vp = fr
                ranges = segment_pixel_ranges(vp, width, height)
            root = b % N
            for t, (kind, tile, seg) in enumerate(frame_tiles(rc)):
                owner = t % N
                if owner == root or (owner != rank and root != rank):
                    continue
                lo, hi = ranges[seg]
                item = (kind, (b * caps[kind] + tile) * col[kind] + lo, hi - lo + 1)
                if owner == rank:
                    send[root].append(item)
                else:
                    recv[owner].append(item)

        def layout(lists):
            spans = []
            offsets = [0]
            for peer in range(N):
                row = offsets[-1]
                for kind, pool_row, rows in lists[peer]:
                    spans.append((pool_row, row, rows, kind))
                    row += rows
                offsets.append(row)
            arr = np.array(spans, dtype=SPAN_DTYPE) if spans else np.zeros(0, dtype=SPAN_DTYPE)
            return arr, offsets

        self.send_spans, self.send_off = layout(send)
        self.recv_spans, self.recv_off = layout(recv)
        self.sent_rows, self.recv_rows = self.send_off[-1], self.recv_off[-1]
        self.payload_bytes = (self.sent_rows + self.recv_rows) * TILE_RAYS * 4
        if self.device.type == "cuda":
            self._send_spans_dev = torch.from_numpy(self.send_spans.view(np.uint8).copy()).to(self.device)
            self._recv_spans_dev = torch.from_numpy(self.recv_spans.view(np.uint8).copy()).to(self.device)

    # staging buffers are shared by all exchanges of a run (sized for the largest)
    @staticmethod
    def allocate_staging(exchanges, device):
        rows_s = max((e.sent_rows for e in exchanges), default=0)
        rows_r = max((e.recv_rows for e in exchanges), default=0)
        send = torch.empty((max(1, rows_s), TILE_RAYS), dtype=torch.int32, device=device)
        recv = torch.empty((max(1, rows_r), TILE_RAYS), dtype=torch.int32, device=device)
        for e in exchanges:
            e.send_buf, e.recv_buf = send, recv
        return send, recv

    def _copy_rows(self, spans, spans_dev, staging, to_packed: bool) -> None:
        if len(spans) == 0:
            return
        if self.device.type == "cuda":
            if self.ctx is None:
                raise RuntimeError("TileExchange on GPU tensors needs the cpuvox_amd.gpu.Context that owns the pools")
            self.ctx.copy_rows(None, to_packed, len(spans), spans_dev.data_ptr(), staging.data_ptr())
            self.ctx.synchronize()
            return
        rows = (self.pools.td.view(-1, TILE_RAYS).numpy(), self.pools.lr.view(-1, TILE_RAYS).numpy())
        stage = staging.numpy()
        for sp in spans:
            a, p, n, k = int(sp["poolRow"]), int(sp["packedRow"]), int(sp["rows"]), int(sp["kind"])
            if to_packed:
                stage[p:p + n] = rows[k][a:a + n]
            else:
                rows[k][a:a + n] = stage[p:p + n]

    def run(self) -> None:
        """Pack my tiles' pixel rows per destination, exchange, unpack the received rows into their pool places.
        The caller orders this after the render (ctx.synchronize()) and before the next one."""
        if not hasattr(self, "send_buf"):
            TileExchange.allocate_staging([self], self.device)
        self._copy_rows(self.send_spans, getattr(self, "_send_spans_dev", None), self.send_buf, True)
        ops = []
        for peer in range(self.N):
            if peer == self.rank:
                continue
            s0, s1 = self.send_off[peer], self.send_off[peer + 1]
            r0, r1 = self.recv_off[peer], self.recv_off[peer + 1]
            if s1 > s0:
                ops.append(dist.P2POp(dist.isend, self.send_buf[s0:s1], peer))
            if r1 > r0:
                ops.append(dist.P2POp(dist.irecv, self.recv_buf[r0:r1], peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._copy_rows(self.recv_spans, getattr(self, "_recv_spans_dev", None), self.recv_buf, False)


# ---------------------------------------------------------------------------------------------------------------
# Zero-copy sharding: the render kernel writes every tile straight into the buffer it has to end up in
# (cvx_draw_segments_placed), so the exchange is nothing but one send and one receive per peer.
# ---------------------------------------------------------------------------------------------------------------
class ShardPlan:
    """Placement of one batch of frames on one rank.

    Tile t of frame b (canonical order of libcpuvox_gpu) is rendered by rank t % N and must end up on the frame's
    display rank b % N.  Per rank and step there are two areas of 256-byte pixel rows:
      send[dest]      rows of my tiles of frames displayed elsewhere, grouped by destination, in (frame, tile) order;
      display[owner]  rows of all tiles of the frames I display, grouped by the rank that renders them, in (frame,
                      tile) order -- the section of owner == me is written by my own kernel, the others arrive by
                      P2P and are, by construction, exactly the peer's send[me] section.
    Only rows [origMin, origMax] of a tile exist in either area.
    """

    def __init__(self, frames, width: int, height: int, rank: int, world_size: int):
        N = world_size
        self.rank, self.N = rank, N
        send_rows = [0] * N
        disp_rows = [0] * N
        self.my_tiles = []       # (canonical index, area 0=send/1=display, section, row offset in section, omin)
        self.display_tiles = {}  # frame b -> list of (kind, tile, seg, owner, row offset in owner's section, omin, rows)
        index = 0
        for b, fr in enumerate(frames):
            if hasattr(fr, "segments"):
                rc = [s.RayCount for s in fr.segments]
                ranges = segment_pixel_ranges(fr.vanishingPointScreenSpace, width, height)
            else:
                rc, vp = fr
                ranges = segment_pixel_ranges(vp, width, height)
            root = b % N
            for t, (kind, tile, seg) in enumerate(frame_tiles(rc)):
                owner = t % N
                lo, hi = ranges[seg]
                rows = hi - lo + 1
                if root == rank:
                    self.display_tiles.setdefault(b, []).append((kind, tile, seg, owner, disp_rows[owner], lo, rows))
                    if owner == rank:
                        self.my_tiles.append((index, 1, rank, disp_rows[rank], lo))
                    disp_rows[owner] += rows
                elif owner == rank:
                    self.my_tiles.append((index, 0, root, send_rows[root], lo))
                    send_rows[root] += rows
                index += 1
        self.tile_count = index
        self.send_start = np.concatenate([[0], np.cumsum(send_rows)]).astype(np.int64)
        self.disp_start = np.concatenate([[0], np.cumsum(disp_rows)]).astype(np.int64)
        self.send_total, self.disp_total = int(self.send_start[-1]), int(self.disp_start[-1])

    def tile_out(self, send_ptr: int, disp_ptr: int) -> np.ndarray:
        """uint64 device address per tile for cvx_draw_segments_placed (0 = not rendered by this rank)."""
        out = np.zeros(self.tile_count, dtype=np.uint64)
        base = (send_ptr, disp_ptr)
        start = (self.send_start, self.disp_start)
        for index, area, section, off, lo in self.my_tiles:
            out[index] = (base[area] + (int(start[area][section]) + off - lo) * (TILE_RAYS * 4)) & 0xFFFFFFFFFFFFFFFF
        return out

    def exchange(self, send: torch.Tensor, disp: torch.Tensor):
        """One isend / irecv per peer on the current stream; returns the requests (wait() orders the stream).
        With the gloo backend and device tensors (single-GPU rehearsals of the multi-GPU path) the transfer is staged
        through host memory and has completed on return."""
        via_host = send.is_cuda and dist.get_backend() == "gloo"
        if via_host:
            torch.cuda.current_stream().synchronize()
        ops = []
        received = []
        for peer in range(self.N):
            if peer == self.rank:
                continue
            s0, s1 = int(self.send_start[peer]), int(self.send_start[peer + 1])
            r0, r1 = int(self.disp_start[peer]), int(self.disp_start[peer + 1])
            if s1 > s0:
                ops.append(dist.P2POp(dist.isend, send[s0:s1].cpu() if via_host else send[s0:s1], peer))
            if r1 > r0:
                if via_host:
                    buf = torch.empty((r1 - r0, TILE_RAYS), dtype=disp.dtype)
                    received.append((r0, r1, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, peer))
                else:
                    ops.append(dist.P2POp(dist.irecv, disp[r0:r1], peer))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if via_host:
            for req in reqs:
                req.wait()
            for r0, r1, buf in received:
                disp[r0:r1].copy_(buf)
            torch.cuda.current_stream().synchronize()
            return []
        return reqs

    def assemble(self, disp: torch.Tensor, frame: int, ray_counts, width: int, height: int):
        """Logical ray-major raybuffers (the reference layout, RayBuffer.cs:121-128; zeros where nothing is written)
        of a frame this rank displays -- for verification / read-back, on the host."""
        td = np.zeros((width + 2 * height, height), dtype=np.uint32)
        lr = np.zeros((2 * width + height, width), dtype=np.uint32)
        rc = [max(0, int(c)) for c in ray_counts]
        seg_tile0 = [0, (rc[0] + TILE_RAYS - 1) // TILE_RAYS, 0, (rc[2] + TILE_RAYS - 1) // TILE_RAYS]
        seg_row0 = [0, rc[0], 0, rc[2]]
        for kind, tile, seg, lo, rows in self.display_rows(disp, frame):
            block = rows.cpu().numpy().view(np.uint32)
            plane0 = (tile - seg_tile0[seg]) * TILE_RAYS
            lanes = min(TILE_RAYS, rc[seg] - plane0)
            buf = td if kind == 0 else lr
            buf[seg_row0[seg] + plane0: seg_row0[seg] + plane0 + lanes, lo: lo + block.shape[0]] = block[:, :lanes].T
        return td, lr

    def display_rows(self, disp: torch.Tensor, frame: int):
        """(kind, tile, seg, omin, rows tensor [n, 64]) of every tile of a frame this rank displays."""
        out = []
        for kind, tile, seg, owner, off, lo, rows in self.display_tiles[frame]:
            a = int(self.disp_start[owner]) + off
            out.append((kind, tile, seg, lo, disp[a:a + rows]))
        return out
