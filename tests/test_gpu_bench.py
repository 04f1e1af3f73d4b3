"""bench.py prints the one JSON line the driver reads: keys, types and the internal checks, on a small batch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("pipeline", [1, 0])
def test_bench_line_contract(pipeline):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "6", "--distinct", "3", "--cpu-sample", "1", "--pipeline", str(pipeline)],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mpixels/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 6 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    assert "workload" in d["config"] and d["config"]["images_per_gpu"] == 6 and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    assert (r["kernel_alone"] is not None) == bool(pipeline)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "Mpixels/s" and "sample" in c
    chk = d["check"]
    assert chk["nbits_all_equal_budget"] and chk["stream_bit_exact_vs_oracle"] and chk["decoded_image_bit_exact_vs_oracle"]
    assert chk["images_checked_vs_oracle"] == 3 and chk["cycled_copies_equal"]  # every distinct image, not image 0 only
    assert d["config"]["distinct_images_per_gpu"] == 3 and d["config"]["gather"] is None
    sl = d["single_image_latency"]
    assert sl["host_api_encode_ms"] > 0 and sl["host_api_decode_ms"] > 0 and sl["host_api_stream_equals_batch"]


def test_bench_under_a_launcher_gathers_through_rccl():
    """With the launcher's environment set (one rank here: a one-GPU box) bench.py takes the multi-GPU path: the
    library's own RCCL communicator (spiht_comm_create), the stream gather queued on the list-coding stream
    (spiht_gather_streams) and the decoder reading this rank's rows of the GATHERED buffers.  No torch is imported."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for pipeline in ("1", "0"):
        p = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
                            "--warmup", "1", "--batch", "4", "--cpu-sample", "0", "--pipeline", pipeline],
                           cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        assert " torch" not in p.stderr  # -X importtime lists every imported module on stderr
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
        assert d["n_gpus"] == 1 and d["check"]["gather_rows_match"] is True
        assert d["config"]["gather"]["world"] == 1 and d["config"]["gather"]["rccl_version"] > 0
        assert d["check"]["nbits_all_equal_budget"] and d["cpu_baseline"] is None
    # should RCCL not come up, the job still runs (host channel for the barrier and the times) and says so
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--cpu-sample", "0"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(env, SPIHT_BENCH_NO_RCCL="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["check"]["gather_rows_match"] is None and "failed" in d["config"]["gather"] and d["value"] > 0


def test_two_ranks_on_one_gpu_without_rccl():
    """`python bench.py --gpus 2` from a bare shell: the process starts its two rank processes itself (before anything
    touches a GPU), they find each other over the host channel (HostGroup: barrier, maximum of the times), each codes ITS
    shard -- images with seeds 1000 + rank * B + i -- and rank 0 prints the one line with n_gpus = 2.  RCCL refuses two
    ranks on one device, so the rehearsal on this one-GPU box runs with the gather switched off (SPIHT_BENCH_NO_RCCL) and
    both ranks on device 0: the line must say that no gather took place.  What the driver's 8-GPU run adds to this is
    ncclCommInitRank and the all-gather themselves, rehearsed with one rank in the test above."""
    env = dict(os.environ, SPIHT_BENCH_NO_RCCL="1", SPIHT_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--cpu-sample", "0"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]   # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["images_per_gpu"] == 4
    # whole-job throughput: both ranks' images over the slower rank's time
    assert abs(d["value"] - 2 * 4 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    g = d["config"]["gather"]
    assert "failed" in g and "SPIHT_BENCH_NO_RCCL" in g["failed"] and "no stream gather" in g["note"]
    assert d["check"]["gather_rows_match"] is None and d["check"]["nbits_all_equal_budget"] and d["cpu_baseline"] is None
    assert "RCCL communicator not available" in p.stderr  # every rank says so on stderr
