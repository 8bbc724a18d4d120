"""bench.py as the driver runs it: the JSON line's contract, the N > 1 launch path on one device, and the N = 1 path
under torch.distributed.run against the bare one (the 8-GPU run must not fail for a trivial reason)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--steps", "6", "--warmup", "2", "--batch", "64", "--no-cpu-baseline", "--rotate", "2"]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _line(p):
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    return json.loads(lines[0])


def _check_contract(d, n_gpus):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "roofline_by_direction"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["world_size_seen_by_backend"] == n_gpus and len(d["per_rank_ms_per_step"]) == n_gpus
    assert d["round_trip_bit_exact"] and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = whole-job uncompressed bytes through encode + decode / the MAX-over-ranks step time
    assert abs(d["value"] - n_gpus * 2 * d["config"]["per_gpu_batch_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_us"] * 1e-6) / 1e9) < 0.02 * r["achieved"]
    # the headline is the cache-cold loop; the one-buffer-set figure is a named extra
    assert "rotating" in d["headline_loop"] and d["rotating_sets"] >= 2 and "ms_per_step_one_buffer_set" in d
    bd = d["roofline_by_direction"]
    assert bd["cold"]["compress"]["frac"] > 0 and bd["cold"]["decompress"]["frac"] > 0 and bd["warm_one_buffer_set"]["compress"]["frac"] > 0


def test_default_style_line_and_cpu_baseline():
    p = subprocess.run([sys.executable, BENCH, "--steps", "6", "--warmup", "2", "--batch", "64", "--rotate", "2"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    d = _line(p)
    _check_contract(d, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "GB/s" and c["cores"] >= 1 and c["value"] > 0 and c["single_thread"]["value"] > 0
    # persistent threads, one per CPU the host GRANTS (cgroup quota): the figure scales with the CPU time the run actually
    # got (cpu_seconds_per_wall_second: other processes of this test session -- pytest-xdist workers -- share the quota)
    got = min(c["cores"], max(1.0, c["host"]["cpu_seconds_per_wall_second"]))
    assert c["cores"] <= c["host"]["visible_cpus"] and c["value"] > 0.5 * got * c["single_thread"]["value"], c


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_one_rank_under_torchrun_equals_the_bare_path():
    bare = _line(subprocess.run([sys.executable, BENCH, "--gpus", "1"] + SMALL, capture_output=True, text=True, timeout=600,
                                env=_clean_env()))
    run = _line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                                "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "1"] + SMALL,
                               capture_output=True, text=True, timeout=600, env=_clean_env()))
    _check_contract(bare, 1)
    _check_contract(run, 1)
    assert set(bare) == set(run) and bare["config"] == run["config"] and bare["metric"] == run["metric"]
    # (six timed steps each, other tests of the session may share the device: same order of magnitude is the claim)
    assert 0.5 < bare["ms_per_step"] / run["ms_per_step"] < 2.0, (bare["ms_per_step"], run["ms_per_step"])


def test_two_ranks_on_one_device():
    # python bench.py --gpus 2 starts its own two ranks (gloo here, both on GPU 0: the box has one GPU)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo"] + SMALL, capture_output=True, text=True,
                       timeout=900, env=_clean_env(DGPU_BENCH_ONE_DEVICE="1"))
    d = _line(p)
    _check_contract(d, 2)
    assert d["dist_backend"] == "gloo"


def test_wrong_world_size_is_refused():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=300,
                       env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stdout + p.stderr)


def test_eight_ranks_on_one_device():
    # BASELINE config 5's launch path at its real world size: python bench.py --gpus 8 starts eight ranks (gloo, all on
    # GPU 0 -- the box has one GPU); rank 0's line must account for all eight, and the job must stay small enough that
    # an 8-rank run on a shared device is safe (peak memory reported by rank 0's process group below 40 GB)
    torch_free0 = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.mem_get_info(0)[0])"], capture_output=True,
                                 text=True, timeout=300)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dist-backend", "gloo", "--steps", "4", "--warmup", "1", "--batch", "32",
                        "--no-cpu-baseline", "--rotate", "2", "--quick"], capture_output=True, text=True, timeout=1200,
                       env=_clean_env(DGPU_BENCH_ONE_DEVICE="1"))
    d = _line(p)
    _check_contract(d, 8)
    assert d["dist_backend"] == "gloo" and d["n_gpus"] == d["world_size_seen_by_backend"] == 8
    assert len(d["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert "8 ranks x 32 independent tensors" in d["config"]["sharding"]
    # per rank: 2 buffer sets x 32 rows x (1 MiB in + 1 MiB out + <= 1.3 MiB archive) + temp: far below 40 GB in all
    assert int(torch_free0.stdout.strip()) > 0


def test_compressed_all_gather_eight_ranks_on_one_device():
    # the compressed collective (bench.py --collective) with eight gloo ranks on GPU 0: bit-exact payload (asserted
    # inside bench.py against the plain all-gather), wire bytes below the raw bytes, one line from rank 0
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dist-backend", "gloo", "--collective", "--steps", "3", "--warmup", "1",
                        "--batch", "16", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200,
                       env=_clean_env(DGPU_BENCH_ONE_DEVICE="1"))
    d = _line(p)
    assert d["n_gpus"] == 8 and d["bit_exact"] is True and d["dist_backend"] == "gloo"
    assert d["config"]["wire_bytes_per_rank"] < d["config"]["per_rank_bytes"]
    assert d["config"]["rows_sent_uncompressed"] == 0
