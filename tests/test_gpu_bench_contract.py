"""bench.py prints ONE JSON line with the contract's keys (tiny workload so the check is quick)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nodes", "200000", "--walks", "16384",
           "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "pairs/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["finite"] is True
    r = d["roofline"]
    # the bench graph is trained in resident cells: its kernel is bound by the L2 atomic units
    # (the central rows' gradients), and `frac` is a fraction of THAT ceiling (never above 1);
    # SURVEY 8d's byte model is reported beside it under its own name
    assert r["kernel"] == "gn2v::sgns_resident_v2_kernel" and r["bound"] == "l2_atomic"
    assert r["unit"] == "G f32 atomic adds/s" and r["peak"] == 331.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - 128 * r["kernel_pairs_per_s"] / 1e9) < 1e-6
    assert r["valu_floor_per_pair"] == 60.5 and 0 < r["valu_issue_frac"] < 1 and "traffic" in r
    assert r["work_over_hbm_peak"] > 0 and abs(r["work_over_hbm_peak"] - r["work_bytes_per_s_gb"] / 8000.0) < 1e-12
    # bandwidth as bandwidth: either the PMC-based figure or the reason it is missing (this
    # tiny workload has no committed profile)
    assert "frac_hbm" in r and (r["traffic"] is not None or "no committed" in r["traffic_missing_reason"])
    assert r["mean_centre_run"] >= 1
    assert "traffic_key" in d["config"]
    # round 6: the counters behind the ceiling (committed TCC passes of kernel and probe), the
    # vector instructions REALLY issued, and what `value` leaves out (the first fit of a handle)
    busy = r["atomic_unit_busy"]
    assert 0.9 < busy["tcc_busy"] <= 1 and busy["of_probe_per_cycle"] > 0.95, busy
    assert 0 < r["valu_issued_frac"] < 1 and r["valu_issued_per_pair"] > r["valu_floor_per_pair"]
    assert d["first_fit_s"] > 0 and d["first_fit"]["steps"] == 1
    assert "max_neighbours 100" in d["config"]["workload"]
    pairs = 2 * 16384 * 1250
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - pairs) < 1e-3 * pairs
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "pairs/s"
    assert isinstance(c["sample"], str)


def test_blocks_mode_line_on_one_gpu():
    d = _run("--parallelism", "blocks", "--no-cpu-baseline", "--round-walks", "8192")
    assert "travelling parts" in d["config"]["parallelism"] and d["finite"] is True
    assert "cpu_baseline" not in d and d["value"] > 0
    # the bench graph (10 M nodes, d = 128) is planned into resident cells since round 4
    assert d["roofline"]["kernel"] == "gn2v::sgns_resident_v2_kernel"
    assert "resident cells" in d["config"]["parallelism"]
    pairs = 2 * 16384 * 1250
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - pairs) < 1e-3 * pairs


def test_cbow_line_on_one_gpu():
    d = _run("--model", "cbow")
    assert d["unit"] == "centres/s" and d["finite"] is True and "CBOW" in d["config"]["workload"]
    centres = 2 * 16384 * 128
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - centres) < 1e-3 * centres
    assert d["roofline"]["kernel"].startswith("gn2v::cbow") and d["roofline"]["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "centres/s" and c["value"] > 0


def test_two_rank_line_through_torch_distributed_run():
    """The N > 1 launch line of the contract (python -m torch.distributed.run ... bench.py --gpus 2)
    on a one-GPU box: both ranks share GPU 0 and exchange over gloo (testing flags of bench.py);
    the collectives, barriers, max-over-ranks timing and the single JSON line are the real code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--nodes", "200000", "--walks", "8192", "--round-walks", "8192", "--backend", "gloo",
           "--share-device"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and res.stdout.strip().splitlines()[-1] == lines[0]  # JSON comes last
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["finite"] is True
    assert "2 GPU(s)" in d["config"]["parallelism"] and "cpu_baseline" not in d
    dd = d["distributed"]
    assert dd["backend"] == "gloo" and dd["world_size"] == 2 and len(dd["per_rank_pairs"]) == 2
    assert sum(dd["per_rank_pairs"]) == 2 * 2 * 8192 * 1250 and min(dd["per_rank_pairs"]) > 0
    assert dd["walk_allgather_ms_alone"] > 0 and dd["half_partition_hop_ms_alone"] > 0
    # what a reader needs to judge a scaling record: round size, groups, memory and exposed waits
    assert dd["walks_per_round_per_rank"] == 8192 and dd["parts_per_group"] >= 1
    assert len(dd["per_rank_hbm_peak_gb"]) == 2 and min(dd["per_rank_hbm_peak_gb"]) > 0
    waits = dd["exposed_hop_wait_ms"]
    assert waits["hops_per_rank"] > 0 and len(waits["mean_per_rank"]) == 2
    pairs = 2 * 2 * 8192 * 1250  # steps x ranks x walks x pairs per walk: the whole-job aggregate
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - pairs) < 1e-3 * pairs


def test_bench_gpus_two_starts_its_own_job():
    """`python bench.py --gpus 2 ...` with no torch.distributed environment: the parent (which
    never touches the GPU) launches the two workers, relays the one JSON line and their status."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--nodes", "200000", "--walks", "8192", "--round-walks", "8192",
           "--backend", "gloo", "--share-device"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout[-500:]  # nothing but the line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["finite"] is True and d["distributed"]["world_size"] == 2
    assert sum(d["distributed"]["per_rank_pairs"]) == 2 * 2 * 8192 * 1250
    # a failing job is reported as failing
    bad = subprocess.run(cmd + ["--parallelism", "single"], capture_output=True, text=True,
                         timeout=900, env=env)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_cbow_runs_as_independent_replicas_on_two_ranks():
    """CBOW does not shard: `bench.py --model cbow --gpus N` is N independent fits,
    the line adds their centres and says "replicas"."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--model", "cbow", "--nodes", "200000", "--walks", "8192", "--backend", "gloo",
           "--share-device"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["unit"] == "centres/s" and d["finite"] is True
    assert "2 independent replicas" in d["config"]["parallelism"]
    assert d["distributed"]["world_size"] == 2 and "replicas only" in d["distributed"]["note"]
    centres = 2 * 2 * 8192 * 128  # steps x ranks x walks x positions
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - centres) < 1e-3 * centres
