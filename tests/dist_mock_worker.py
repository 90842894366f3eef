"""One rank of tests/test_distributed_cpu.py::test_gather_frame_with_several_ranks_over_a_mock_transport.

The whole multi-rank exchange of the product through the C ABI - ptmi_dist_init, per frame ptmi_render_frame +
ptmi_gather_frame (ncclSend on the peers, N - 1 ncclRecv on the destination, exact tile sizes), ptmi_read_frame,
ptmi_dist_barrier, ptmi_dist_allreduce_max - with every rank a process of its own on the ONE GPU of the test box.  RCCL
refuses that, so "librccl.so.1" is tests/mock_rccl.cpp here (first on LD_LIBRARY_PATH): only the wire is a stand-in.
    argv: W H spp row_block world rank dst what id_file [mode]
mode "async": the asynchronous behaviour the product's own ordering must survive (the mock's worker delays every transfer,
    PTMI_MOCK_RCCL_DELAY_MS): frame k is rendered and its gather ENQUEUED, the camera changes and frame k + 1 is rendered at once,
    no wait in between - frame k must still arrive bit-identical on the destination (the resolve of frame k + 1 waits on the
    device for the gather that still reads the tile: resolve_gate), then frame k + 1 is gathered and checked, and a third frame
    queues two gathers back to back on the exchange stream (staging buffers reused in stream order).
mode "fail": PTMI_MOCK_RCCL_FAIL makes an nccl call inside the group fail on the destination: ptmi_gather_frame must report
    PTMI_E_DIST, and so must the next ptmi_dist_barrier (a communicator whose group was abandoned is unusable, not silently
    out of step)."""
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
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("PTMI_WORKER_DEADLINE_S", "240")), exit=True)      # a hung exchange ends with a traceback, not a silent timeout
    W, H, spp, row_block, world, rank, dst, what = (int(a) for a in sys.argv[1:9])
    id_file = sys.argv[9]
    mode = sys.argv[10] if len(sys.argv) > 10 else ""
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
    assert r.dist_comm_count() == world
    if mode == "fail":
        r.render_frame()
        if rank == dst:
            for call in (lambda: r.gather_frame(dst, what), r.dist_barrier, lambda: r.gather_frame(dst, what), lambda: r.dist_allreduce_max(1.0)):
                try:
                    call()
                    print("mock-rccl-fail NOT REPORTED", flush=True); sys.exit(1)
                except ptmi.PtmiError as e:
                    assert e.code == -6, e                      # PTMI_E_DIST
            r.dist_finalize()
            print("mock-rccl-fail OK", flush=True)
        else:
            r.gather_frame(dst, what)                            # its send is simply never received
        r.close()
        sys.exit(0)
    if mode == "async":
        import ctypes as C
        o = OracleScene.load(os.path.join(SCENES, "cbox.obj"))
        state = np.zeros((H * W, 6), np.uint32)
        yaws = (90.0, 60.0, 130.0, 75.0)
        def cam(y):
            c = ptmi.default_camera(); c.yaw_deg = y; return c
        def ocam(y):
            c = default_camera(); c.yaw_deg = y; return c
        want = []
        if rank == dst:
            for k, y in enumerate(yaws):
                want.append(o.render(ocam(y), W, H, spp, n_threads=2, rng_state=state, reset_rng=(k == 0))[:2])
        ok = True
        def check(k, tag):
            if rank != dst: return True
            frgb, frad = r.read_frame(rgb8=bool(what & 1), radiance=bool(what & 2))
            good = True
            if what & 1: good = good and bool((frgb == want[k][0]).all())
            if what & 2: good = good and bool((frad.view(np.uint32) == want[k][1].view(np.uint32)).all())
            if not good: print(f"mock-rccl-async frame {k} ({tag}) CORRUPTED", flush=True)
            return good
        r.set_camera(cam(yaws[0])); r.render_frame(); r.gather_frame(dst, what)       # frame 0: only enqueued
        r.set_camera(cam(yaws[1])); r.render_frame()                                  # frame 1 rendered while frame 0 is still on its way
        ok = check(0, "read after the next frame was rendered") and ok
        r.gather_frame(dst, what)
        ok = check(1, "second gather") and ok
        r.set_camera(cam(yaws[2])); r.render_frame(); r.gather_frame(dst, what)       # two gathers back to back on the exchange stream,
        r.set_camera(cam(yaws[3])); r.render_frame(); r.gather_frame(dst, what)       # nothing waited for in between
        ok = check(3, "last of two queued gathers") and ok
        r.dist_barrier()
        if rank == dst: print("mock-rccl-async", "OK" if ok else "MISMATCH", flush=True)
        r.close()
        sys.exit(0 if ok else 1)
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
