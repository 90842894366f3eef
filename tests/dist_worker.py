"""Worker for tests/test_distributed_cpu.py: one rank of a gloo world.  Each rank produces ITS rows of the frame
with the CPU oracle (standing in for a GPU - this test is about the sharding arithmetic and the gather), the frame
is assembled with the product's FrameGather, and rank 0 checks it bit-for-bit against an unsharded render."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi  # noqa: E402
import ptmi_dist  # noqa: E402
from oracle_binding import OracleScene, SCENES, default_camera  # noqa: E402


def main():
    W, H, spp, row_block = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fg = ptmi_dist.FrameGather(dist, W, H, world, rank, row_block, torch.device("cpu"))
    o = OracleScene.load(os.path.join(SCENES, "cbox.obj"))
    rows = ptmi_dist.row_maps(H, world, row_block)[rank]
    assert (rows == ptmi.host_local_row_map(H, world, rank, row_block)).all()
    if len(sys.argv) > 5 and sys.argv[5] == "gpu":
        # the product path: every rank renders ITS tile on the GPU (all ranks share device 0 here, so RCCL itself cannot
        # carry them: it refuses two ranks on one device), two frames; the tiles travel by gloo in the exact sizes and order
        # ptmi_gather_frame receives them, rank 0 places them with the product's kernel and checks the second frame
        # against the unsharded oracle frame
        r = ptmi.Renderer(0)
        r.load_scene(os.path.join(SCENES, "cbox.obj"))
        r.update_resolution(W, H, n_ranks=world, rank=rank, row_block=row_block)
        r.set_config(spp=spp, max_depth=5)
        state = np.zeros((H * W, 6), np.uint32)
        out = None
        for frame in range(2):
            r.render_frame()
            rgb, rad = r.read_image()
            # exact tile sizes, rank order: what ptmi_gather_frame's ncclRecv calls deliver into the staging buffers
            tiles = [None] * world if rank == 0 else None
            dist.gather_object((rgb, rad), tiles, dst=0)
            if rank == 0:
                # the product's own placement kernel (csrc/dist.hip: ptmi_place_tiles) assembles the frame
                out = r.debug_place_tiles(W, H, world, row_block, np.concatenate([t[0].reshape(-1) for t in tiles]),
                                          np.concatenate([t[1].reshape(-1) for t in tiles]))
        r.close()
        ok = True
        if rank == 0:
            frgb, frad = out
            for frame in range(2):
                orgb, orad, _ = o.render(default_camera(), W, H, spp, n_threads=2, rng_state=state, reset_rng=(frame == 0))
            ok = bool((frad.view(np.uint32) == orad.view(np.uint32)).all() and (frgb == orgb).all())
            print("dist-gather-gpu", "OK" if ok else "MISMATCH", W, H, world, row_block, flush=True)
        flag = torch.tensor([1 if ok else 0])
        dist.broadcast(flag, src=0)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if int(flag.item()) == 1 else 1)
    # contiguous runs of rows -> one oracle call each (rows [y0, y1) of the full frame)
    rad = np.zeros((H, W, 3), np.float32); rgb = np.zeros((H, W, 3), np.uint8)
    y = 0
    while y < len(rows):
        y1 = y
        while y1 + 1 < len(rows) and rows[y1 + 1] == rows[y1] + 1:
            y1 += 1
        a, b, _ = o.render(default_camera(), W, H, spp, y0=int(rows[y]), y1=int(rows[y1]) + 1, n_threads=2)
        rgb[rows[y]:rows[y1] + 1] = a[rows[y]:rows[y1] + 1]; rad[rows[y]:rows[y1] + 1] = b[rows[y]:rows[y1] + 1]
        y = y1 + 1
    fg.send_rad[: len(rows)] = torch.from_numpy(rad[rows]); fg.send_rgb[: len(rows)] = torch.from_numpy(rgb[rows])
    out = fg.gather()
    ok = True
    if rank == 0:
        frad, frgb = out
        orgb, orad, _ = o.render(default_camera(), W, H, spp, n_threads=2)
        ok = bool((frad.numpy().view(np.uint32) == orad.view(np.uint32)).all() and (frgb.numpy() == orgb).all())
        print("dist-gather", "OK" if ok else "MISMATCH", W, H, world, row_block, flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
