"""Multi-GPU frame assembly: interleaved row-block tiles -> one frame on the destination rank.

The render path shards by independent pixels (RNG streams are keyed by the global pixel index), so the only
exchange step of a frame is this gather at its end: every rank contributes its rows, the destination places
them.  torch.distributed is used as plumbing only (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the
CPU tests); tile payloads are uniform-size (padded to the largest rank) so ONE gather call moves each image.
"""
import numpy as np
import torch

import ptmi


def row_maps(height, world, row_block):
    """Global row indices owned by each rank (same arithmetic as the C ABI's ptmi_host_local_row_map)."""
    return [ptmi.host_local_row_map(height, world, k, row_block).astype(np.int64) for k in range(world)]


def dist_init_from_torch(renderer, dist):
    """ptmi_dist_init for a process that already has a torch.distributed group: rank 0 draws the ncclUniqueId
    (ptmi_dist_unique_id) and the 128 bytes travel to every rank through the group; the frame exchange itself then runs
    inside libptmi.so (ptmi_gather_frame), not through torch."""
    world, rank = dist.get_world_size(), dist.get_rank()
    box = [ptmi.Renderer.dist_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    renderer.dist_init(box[0], world, rank)


class FrameGather:
    """torch.distributed gather of padded tiles + row placement with tensor indexing: the CPU (gloo) rehearsal of the frame
    exchange used by the tests of the row arithmetic.  On GPUs the exchange is ptmi_gather_frame (csrc/dist.hip)."""

    def __init__(self, dist, width, height, world, rank, row_block, device, dst=0, n_send=1):
        self.dist, self.world, self.rank, self.dst = dist, world, rank, dst
        self.width, self.height = width, height
        self.maps = row_maps(height, world, row_block)
        self.max_rows = max(len(m) for m in self.maps)
        self.n_local = len(self.maps[rank])
        # one payload per rank: [float32 radiance | uint8 rgb] so a frame needs exactly ONE collective
        n_px = self.max_rows * width
        self._n_rad_bytes = n_px * 12
        # n_send > 1: a ring of send buffers, so that frame k + 1 can be rendered and staged while the gather of frame k
        # is still reading its buffer (the caller waits on its own event before reusing a slot)
        self.sends = [torch.zeros(n_px * 15, dtype=torch.uint8, device=device) for _ in range(max(1, n_send))]
        self.sends_rad = [b[: self._n_rad_bytes].view(torch.float32).view(self.max_rows, width, 3) for b in self.sends]
        self.sends_rgb = [b[self._n_rad_bytes:].view(self.max_rows, width, 3) for b in self.sends]
        self.send, self.send_rad, self.send_rgb = self.sends[0], self.sends_rad[0], self.sends_rgb[0]
        if rank == dst:
            self.recv = [torch.empty_like(self.sends[0]) for _ in range(world)]
            self.maps_t = [torch.from_numpy(m).to(device) for m in self.maps]
            self.frame_rad = torch.empty((height, width, 3), dtype=torch.float32, device=device)
            self.frame_rgb = torch.empty((height, width, 3), dtype=torch.uint8, device=device)

    def gather(self, slot=0):
        """sends_rad / sends_rgb[slot][:n_local] must hold this rank's rows.  Returns (frame_rad, frame_rgb) on dst, else
        None.  With the nccl backend the call only enqueues work (collective + row placement) on the current stream."""
        if self.world == 1 and self.dist is None:
            return self.sends_rad[slot][: self.n_local], self.sends_rgb[slot][: self.n_local]
        is_dst = self.rank == self.dst
        self.dist.gather(self.sends[slot], self.recv if is_dst else None, dst=self.dst)
        if not is_dst:
            return None
        for k in range(self.world):
            m = self.maps_t[k]
            rad = self.recv[k][: self._n_rad_bytes].view(torch.float32).view(self.max_rows, self.width, 3)
            rgb = self.recv[k][self._n_rad_bytes:].view(self.max_rows, self.width, 3)
            self.frame_rad[m] = rad[: len(m)]
            self.frame_rgb[m] = rgb[: len(m)]
        return self.frame_rad, self.frame_rgb
