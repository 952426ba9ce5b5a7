"""CPU: `python bench.py --gpus N` without a launcher starts N ranks of itself (fresh child processes, parent only
collects) - the N-rank rendezvous, barrier and reductions run here over gloo with world size 2, no GPU, no engine."""
import json
import os
import subprocess
import sys

import pytest

from tests import _harness as H


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    d = _run(["--gpus", "2", "--launch-check"])
    assert d["launch_check"] and d["n_gpus"] == 2
    assert d["streams_total"] == 2 * 65536           # SUM over ranks of the contiguous shards = the whole job
    assert abs(d["max_time"] - 0.002) < 1e-12        # MAX over ranks
    assert d["per_rank_streams"] == [65536.0, 65536.0]       # every rank's own figure, gathered


def test_single_rank_needs_no_rendezvous():
    d = _run(["--gpus", "1", "--launch-check"])
    assert d["n_gpus"] == 1 and d["streams_total"] == 65536


def test_under_an_external_launcher_nothing_is_spawned():
    """With WORLD_SIZE in the environment (torch.distributed.run) the script is a rank, not a launcher."""
    d = _run(["--gpus", "2", "--launch-check"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert d["n_gpus"] == 1


def test_eight_ranks_plan_the_8m_stream_job():
    """BASELINE configs[4]: 8M streams over 8 ranks.  Eight gloo ranks (CPU) rendezvous, take their contiguous shards and
    plan the HBM of their share of the transcode (sharding.plan_transcode_bytes: I/O and per-stream state grow with the
    shard, the engine's workspaces stop at ac3mi_set_tile_frames' default): every rank fits 288 GB with room to spare."""
    d = _run(["--gpus", "8", "--launch-check", "--job-streams", str(8 * (1 << 20))])
    assert d["n_gpus"] == 8 and d["streams_total"] == 8 * (1 << 20)
    assert d["ranks_that_fit_288GB"] == 8
    assert 20e9 < d["max_rank_hbm_plan_bytes"] < 60e9


@pytest.mark.gpu
def test_two_rank_line_carries_every_ranks_rate_and_the_cpu_baseline():
    """The N-rank code path of the real bench on the one-GPU box: both ranks on device 0 over gloo (AC3MI_BENCH_REHEARSE=1;
    RCCL refuses two ranks on one device).  The line must name BASELINE's metric, carry one rate per rank - a slow GPU
    cannot hide behind the MAX-reduced time - and keep its cpu_baseline (rank 0 measures it after the barrier)."""
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "2048", "--no-extra"],
             env_extra={"AC3MI_BENCH_REHEARSE": "1", "AC3MI_BENCH_MILLION": "0"})
    assert d["n_gpus"] == 2 and d["metric"] == json.load(open(os.path.join(H.ROOT, "BASELINE.json")))["metric"]
    assert len(d["per_rank_frames_per_s"]) == 2 and all(x > 0 for x in d["per_rank_frames_per_s"])
    assert abs(d["value"] - 2 * 2048 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["all_frames_ok"] and d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0
    assert d["roofline"]["hbm"]["frac"] > 0
