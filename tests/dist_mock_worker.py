"""One rank of tests/test_distributed_cpu.py::test_gather_frame_with_several_ranks_over_a_mock_transport.

The whole multi-rank exchange of the product through the C ABI - ptmi_dist_init, per frame ptmi_render_frame +
ptmi_gather_frame (ncclSend on the peers, N - 1 ncclRecv on the destination, exact tile sizes), ptmi_read_frame,
ptmi_dist_barrier, ptmi_dist_allreduce_max - with every rank a process of its own on the ONE GPU of the test box.  RCCL
refuses that, so "librccl.so.1" is tests/mock_rccl.cpp here (first on LD_LIBRARY_PATH): only the wire is a stand-in.
    argv: W H spp row_block world rank dst what id_file"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi  # noqa: E402
from oracle_binding import OracleScene, SCENES, default_camera  # noqa: E402


def main():
    W, H, spp, row_block, world, rank, dst, what = (int(a) for a in sys.argv[1:9])
    id_file = sys.argv[9]
    if rank == 0:
        uid = ptmi.Renderer.dist_unique_id()
        with open(id_file + ".tmp", "wb") as f: f.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 120
            time.sleep(0.01)
        uid = open(id_file, "rb").read()
    r = ptmi.Renderer(0)
    r.load_scene(os.path.join(SCENES, "cbox.obj"))
    r.update_resolution(W, H, n_ranks=world, rank=rank, row_block=row_block)
    r.set_config(spp=spp, max_depth=5)
    r.dist_init(uid, world, rank)
    for frame in range(2):                                       # the second gather is enqueued behind the first
        r.render_frame()
        r.gather_frame(dst, what)
    r.dist_barrier()
    assert r.dist_allreduce_max(1.5 * rank) == 1.5 * (world - 1)
    ok = True
    if rank == dst:
        frgb, frad = r.read_frame(rgb8=bool(what & 1), radiance=bool(what & 2))
        o = OracleScene.load(os.path.join(SCENES, "cbox.obj"))
        state = np.zeros((H * W, 6), np.uint32)
        for frame in range(2):
            orgb, orad, _ = o.render(default_camera(), W, H, spp, n_threads=2, rng_state=state, reset_rng=(frame == 0))
        if what & 1: ok = ok and bool((frgb == orgb).all())
        if what & 2: ok = ok and bool((frad.view(np.uint32) == orad.view(np.uint32)).all())
        print("mock-rccl-gather", "OK" if ok else "MISMATCH", W, H, world, row_block, dst, what, flush=True)
    r.dist_barrier()
    r.close()
    sys.exit(0 if ok else 1)


main()
