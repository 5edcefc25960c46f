"""`python bench.py --gpus N` starts its own ranks (gencomm_amd/launch.py): spawn, rank environment, JSON collection from rank 0,
non-zero exit propagation -- exercised on the CPU with two gloo ranks THROUGH bench.py's own entry point (`--workload
launch_selftest` replaces the GPU step by a sleep; every other line of the flow is the benchmark's), and under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

from gencomm_amd import launch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", launch.ENV_LAUNCHED, launch.ENV_SHARE_DEVICE)}


def _check_line(out, world, steps, batch):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                                   # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    ranks = d["ranks"]
    assert d["n_gpus"] == world and [r["rank"] for r in ranks] == list(range(world))
    assert len({r["pid"] for r in ranks}) == world                # one process per rank
    assert all(r["scenes"] == steps * batch for r in ranks)
    total, slowest = sum(r["scenes"] for r in ranks), max(r["elapsed"] for r in ranks)
    assert d["total_scenes"] == total
    assert d["value"] == pytest.approx(total / slowest, rel=1e-9)  # aggregate = all ranks' scenes / the slowest rank's time
    assert d["ms_per_step"] == pytest.approx(1e3 * slowest / steps, rel=1e-9)
    assert slowest >= max(r["busy"] for r in ranks)                # the barrier makes everyone wait for the slowest
    assert ranks[-1]["busy"] > 1.5 * ranks[0]["busy"]              # rank 1 really was the slow one
    return d


def test_bench_self_launches_two_ranks_and_aggregates():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "launch_selftest", "--steps", "6", "--batch", "3"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    d = _check_line(p.stdout, 2, 6, 3)
    for r in d["ranks"]:
        assert r["env"] == {"RANK": str(r["rank"]), "LOCAL_RANK": str(r["rank"]), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "GENCOMM_LAUNCHED": "1"}


def test_bench_self_launches_eight_ranks():
    """world size 8 -- the node BASELINE.json names -- through the same launcher and aggregation (gloo on the CPU; reference
    multi_gpu_utils.py:16-38 reads the same RANK / WORLD_SIZE / LOCAL_RANK environment)"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--workload", "launch_selftest", "--steps", "3", "--batch", "2"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    d = _check_line(p.stdout, 8, 3, 2)
    assert [r["env"]["LOCAL_RANK"] for r in d["ranks"]] == [str(i) for i in range(8)]
    assert all(r["env"]["WORLD_SIZE"] == "8" for r in d["ranks"])


def test_a_failing_rank_stops_the_job_with_its_exit_code():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "launch_selftest", "--steps", "3"],
                       env=dict(_clean_env(), GENCOMM_SELFTEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr)
    assert "rank 1 exited with 3" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]   # no result line from a failed job


def test_torchrun_environment_is_used_as_is():
    port = str(launch.free_port())
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, BENCH, "--gpus", "2", "--workload", "launch_selftest", "--steps", "4", "--batch", "1"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    d = _check_line(p.stdout, 2, 4, 1)
    assert all(r["env"]["GENCOMM_LAUNCHED"] is None for r in d["ranks"])   # not re-launched by bench.py


def test_world_size_mismatch_is_refused():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "launch_selftest"],
                       env=dict(_clean_env(), RANK="0", WORLD_SIZE="3", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "does not match --gpus" in p.stderr


def test_launcher_helpers():
    assert launch.needs_launch(2, {}) and not launch.needs_launch(1, {})
    assert not launch.needs_launch(4, {"RANK": "1", "WORLD_SIZE": "4"}) and not launch.needs_launch(4, {launch.ENV_LAUNCHED: "1"})
    e = launch.rank_env(3, 8, 1234, share_device=True, base={})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_PORT"], e["MASTER_ADDR"]) == ("3", "3", "8", "1234", "127.0.0.1")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert launch.device_index(3, e) == 0 and launch.backend(e) == "gloo"          # share-device rehearsal
    assert launch.device_index(3, {}) == 3 and launch.backend({}) == "nccl"
    rc, out = launch.spawn_ranks([sys.executable, "-c", "import os; print('r' + os.environ['RANK'])"], 3, echo=False)
    assert rc == 0 and out == "r0\n"
    rc, _ = launch.spawn_ranks([sys.executable, "-c", "import time; time.sleep(30)"], 2, timeout=0.5, echo=False)
    assert rc == 124
