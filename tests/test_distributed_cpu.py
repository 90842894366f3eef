"""N > 1 path on CPU: world_size-2 and -3 gloo processes shard a frame by interleaved row blocks, gather it with
the product's FrameGather and must reproduce the unsharded frame bit-for-bit (rendering by the oracle)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,W,H,row_block", [(2, 40, 37, 8), (2, 32, 16, 4), (3, 24, 50, 8)])
def test_gloo_tile_gather_reassembles_the_frame(world, W, H, row_block):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_worker.py"), str(W), str(H), "3", str(row_block)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "dist-gather OK" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,W,H,row_block", [(2, 96, 70, 2), (4, 64, 64, 8), (3, 50, 33, 1)])
def test_gloo_ranks_render_their_tiles_on_the_gpu(world, W, H, row_block):
    """the product path under N > 1: every rank drives its own ptmi_ctx (sharing the one GPU of the test box), renders its
    interleaved row blocks, and the gathered second frame equals the unsharded oracle frame bit for bit"""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_worker.py"), str(W), str(H), "3", str(row_block), "gpu"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "dist-gather-gpu OK" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,W,H,row_block,dst,what", [(2, 96, 70, 2, 0, 3), (4, 64, 64, 8, 0, 1), (3, 50, 33, 1, 2, 2), (4, 40, 21, 4, 1, 3), (4, 16, 8, 4, 0, 3)])
def test_gather_frame_with_several_ranks_over_a_mock_transport(tmp_path, world, W, H, row_block, dst, what):
    """ptmi_dist_init / ptmi_gather_frame / ptmi_read_frame / barrier / max-reduction with 2 - 4 ranks (the GPU box allows six
    processes on its card, this one included), every rank a process
    with its own ptmi_ctx on the test box's one GPU.  RCCL refuses two ranks on a device, so librccl.so.1 is
    tests/mock_rccl.cpp for these processes (files as the wire): the product's send / receive group, exact tile sizes and
    offsets, ragged tilings (a rank without rows included), 8-bit / float / both payloads, a destination other than rank 0,
    two gathers in flight and the placement kernel all run as they would over xGMI."""
    root = os.path.dirname(HERE)
    lib_dir = tmp_path / "lib"; lib_dir.mkdir()
    wire = tmp_path / "wire"; wire.mkdir()
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-o", str(lib_dir / "librccl.so.1"),
                    os.path.join(HERE, "mock_rccl.cpp")], check=True, timeout=600)
    env = dict(os.environ, OMP_NUM_THREADS="2", PTMI_MOCK_RCCL_DIR=str(wire), GPU_MAX_HW_QUEUES="8",
               LD_LIBRARY_PATH=str(lib_dir) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_mock_worker.py"), str(W), str(H), "3", str(row_block), str(world), str(k),
                               str(dst), str(what), str(tmp_path / "id.bin")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for k, p in enumerate(procs):
        assert p.returncode == 0, f"rank {k}: " + outs[k][0][-1500:] + outs[k][1][-1500:]
    assert "mock-rccl-gather OK" in outs[dst][0]


def _mock_env(tmp_path, **extra):
    lib_dir = tmp_path / "lib"; lib_dir.mkdir()
    wire = tmp_path / "wire"; wire.mkdir()
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-o", str(lib_dir / "librccl.so.1"),
                    os.path.join(HERE, "mock_rccl.cpp")], check=True, timeout=600)
    # GPU_MAX_HW_QUEUES=8: every stream of a rank on a hardware queue of its own (tests/mock_rccl.cpp says why)
    return dict(os.environ, OMP_NUM_THREADS="2", PTMI_MOCK_RCCL_DIR=str(wire), GPU_MAX_HW_QUEUES="8",
                LD_LIBRARY_PATH=str(lib_dir) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), **extra)


@pytest.mark.gpu
@pytest.mark.parametrize("world,W,H,row_block,dst,what", [(2, 96, 70, 2, 0, 3), (3, 64, 48, 8, 1, 1)])
def test_frames_survive_an_asynchronous_exchange(tmp_path, world, W, H, row_block, dst, what):
    """The stand-in transport is asynchronous and stream-ordered like RCCL (tests/mock_rccl.cpp) and here delays every transfer
    by 300 ms: frame k is gathered while frame k + 1 (another camera) is already being rendered, nothing is waited for in
    between, two gathers queue up on the exchange stream.  What keeps frame k intact is the product's own ordering: the
    resolve of frame k + 1 waits on the device for the gather that still reads the tile (checked once by hand with that wait
    compiled out: frame 0 then arrives with frame 1's rows and this test fails - DESIGN.md 6)."""
    env = _mock_env(tmp_path, PTMI_MOCK_RCCL_DELAY_MS="300")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_mock_worker.py"), str(W), str(H), "3", str(row_block), str(world), str(k),
                               str(dst), str(what), str(tmp_path / "id.bin"), "async"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for k, p in enumerate(procs):
        assert p.returncode == 0, f"rank {k}: " + outs[k][0][-1500:] + outs[k][1][-1500:]
    assert "mock-rccl-async OK" in outs[dst][0]


@pytest.mark.gpu
def test_a_failed_exchange_is_reported_and_poisons_the_communicator(tmp_path):
    """An nccl call that fails between ncclGroupStart and ncclGroupEnd (the stand-in's ncclRecv, by injection): the error comes
    back as PTMI_E_DIST, the abandoned group is closed, and every later ptmi_dist_* call fails too instead of queueing into a
    dead group and returning as if it had synchronised."""
    env = _mock_env(tmp_path, PTMI_MOCK_RCCL_FAIL="recv")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_mock_worker.py"), "48", "32", "2", "4", "2", str(k), "0", "3",
                               str(tmp_path / "id.bin"), "fail"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for k, p in enumerate(procs):
        assert p.returncode == 0, f"rank {k}: " + outs[k][0][-1500:] + outs[k][1][-1500:]
    assert "mock-rccl-fail OK" in outs[0][0]


@pytest.mark.gpu
def test_bench_with_three_ranks_over_the_mock_transport(tmp_path):
    """bench.py --gpus 3 exactly as the driver launches it (torch.distributed.run, one process per rank), the frame exchange
    through ptmi_gather_frame - with the three ranks sharing the test box's GPU, librccl.so.1 = tests/mock_rccl.cpp and torch's
    own group on gloo.  The line must be a valid weak-scaling line of three ranks."""
    import json
    root = os.path.dirname(HERE)
    lib_dir = tmp_path / "lib"; lib_dir.mkdir()
    wire = tmp_path / "wire"; wire.mkdir()
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-o", str(lib_dir / "librccl.so.1"),
                    os.path.join(HERE, "mock_rccl.cpp")], check=True, timeout=600)
    env = dict(os.environ, OMP_NUM_THREADS="2", PTMI_MOCK_RCCL_DIR=str(wire), PTMI_RCCL_LIB=str(lib_dir / "librccl.so.1"), GPU_MAX_HW_QUEUES="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "3", "--warmup", "1",
           "--rehearse-shared-gpu", "--no-cpu", "--spp", "4", "--side", "192"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert 0.0 < line["roofline"]["frac"] <= 1.0
    assert any(f.startswith("mock-") for f in os.listdir(wire))                      # the exchange really went over the mock


@pytest.mark.gpu
def test_bench_exchange_path_through_rccl_with_one_rank():
    """bench.py's N > 1 code path (process group "nccl" = RCCL, device-to-device staging, pipelined gather) with a world of
    one rank - all that a one-GPU box can run of it - must produce a valid line."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-dist",
                        "--no-cpu", "--spp", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 100 and line["unit"] == "Msamples/s"
    assert line["roofline"]["bound"] == "hbm" and "cpu_baseline" not in line
    assert 0.0 < line["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
def test_rccl_gather_through_the_c_abi_one_rank_and_placement_kernel():
    """ptmi_dist_init / ptmi_gather_frame / ptmi_read_frame with a world of one rank (all a one-GPU box can run of RCCL) and
    the destination's placement kernel on 2-, 3-, 5- and 8-rank tilings (tiles cut on the host, ragged sizes included)."""
    import numpy as np
    import ptmi
    from oracle_binding import SCENES
    r = ptmi.Renderer(0)
    try:
        r.load_scene(os.path.join(SCENES, "cbox.obj"))
        r.update_resolution(70, 45); r.set_config(spp=4, max_depth=5)
        with pytest.raises(ptmi.PtmiError):
            r.gather_frame(0, 3)                                   # before ptmi_dist_init
        r.dist_init(ptmi.Renderer.dist_unique_id(), 1, 0)
        for what in (1, 2, 3, 3):
            r.render_frame()
            r.gather_frame(0, what)                                # enqueues; the next frame may start at once
        rgb, rad = r.read_image()
        frgb, frad = r.read_frame()
        assert (frgb == rgb).all() and (frad.view(np.uint32) == rad.view(np.uint32)).all()
        r.dist_barrier()
        assert r.dist_allreduce_max(3.25) == 3.25
        with pytest.raises(ptmi.PtmiError):
            r.gather_frame(1, 3)                                   # dst out of range
        r.update_resolution(70, 45, n_ranks=2, rank=0, row_block=8)
        with pytest.raises(ptmi.PtmiError):
            r.gather_frame(0, 3)                                   # tiling disagrees with the communicator
        r.dist_finalize()
        rng = np.random.default_rng(5)
        for (W, H, n, rb) in ((64, 64, 2, 8), (33, 50, 3, 8), (17, 9, 5, 2), (40, 37, 8, 1), (8, 3, 4, 8)):
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8); rad = rng.random((H, W, 3), dtype=np.float32)
            maps = [ptmi.host_local_row_map(H, n, k, rb) for k in range(n)]
            t_rgb = np.concatenate([rgb[m].reshape(-1) for m in maps]); t_rad = np.concatenate([rad[m].reshape(-1) for m in maps])
            o_rgb, o_rad = r.debug_place_tiles(W, H, n, rb, t_rgb, t_rad)
            assert (o_rgb == rgb).all() and (o_rad.view(np.uint32) == rad.view(np.uint32)).all(), (W, H, n, rb)
    finally:
        r.close()


@pytest.mark.gpu
def test_bench_line_carries_every_contract_field():
    import json
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--spp", "16"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                    # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "Msamples/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0.0 < d["roofline"]["frac"] <= 1.0          # a fraction of the peak, never above it
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert "profile_stale" in d["roofline"] and "algorithmic_GBps" in d["roofline"] and "served_from" in d["roofline"]
    assert d["value_incl_d2h"] > 0 and d["config"]["frames"].startswith("one ptmi_render_frame call per step")
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 3` with NO launcher around it (WORLD_SIZE unset), as the driver may run it: bench.py starts the three
    ranks itself before anything touches a GPU, relays rank 0's line and exits with the worst child's code.  Default N > 1 line:
    weak-scaled c2 as `value` plus the strong-scaled BASELINE configurations (c4; c5 by the default walk and through both trees; shrunk here by the
    rehearsal switch) and the rank count RCCL itself reports.  The ranks share the test box's GPU over tests/mock_rccl.cpp."""
    import json
    root = os.path.dirname(HERE)
    env = _mock_env(tmp_path)
    env["PTMI_RCCL_LIB"] = str(tmp_path / "lib" / "librccl.so.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1", "--rehearse-shared-gpu", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["scaling"] == "weak" and line["value"] > 0 and line["rccl_ranks"] == 3
    extras = {e["name"]: e for e in line["extra_configs"]}
    assert set(extras) == {"c4", "c5strong", "c5strong_packed", "c5strong_fast", "c5"} and all("error" not in e and e["value"] > 0 and e["scaling"] == "strong" for e in extras.values()), extras
    # a rank that fails must fail the whole run: an unknown flag makes every child exit non-zero
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-such-flag"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_bench_keeps_torch_on_gloo():
    """the only RCCL communicator of a bench.py process is libptmi's own; torch.distributed carries the 128-byte id
    (and the rehearsals) on gloo, and the timed region's barrier / max over ranks go through ptmi_dist_barrier / ptmi_dist_allreduce_max."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    assert 'init_process_group("nccl"' not in src and "init_process_group('nccl'" not in src
    assert 'init_process_group("gloo"' in src and "r.dist_barrier()" in src and "r.dist_allreduce_max(" in src
    assert 'assert out["rccl_ranks"] == world' in src
