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
    assert d["roofline"]["bound"] in ("hbm", "mfma") and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
