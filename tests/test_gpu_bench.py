"""bench.py end to end on the GPU box: the JSON contract, and the N > 1 launcher on one device
(two ranks over gloo share cuda:0 -- the control flow of the multi-GPU run without a second GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_single_gpu_contract(cuda):
    r = _run(["--steps", "20", "--warmup", "8", "--no-cpu-baseline", "--scenes", "4"])
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["warmup"] == 8
    # the same replay structure at every N: 8 steps per graph, no untimed steps beyond --warmup; round
    # 4's schedule (one K-step graph behind 400 pre-warm steps) rides along as a side figure
    assert r["config"]["steps_per_replay"] == 8 and r["config"]["prewarm_steps"] == 0
    assert r["round4_protocol"]["steps_per_replay"] == 20 and r["round4_protocol"]["value"] > 0
    assert r["roofline"]["traffic"] and r["roofline"]["traffic_over_algorithmic"] > 0
    assert r["unit"] == "voxels/s" and r["value"] > 0 and r["scaling"] == "weak"
    for key in ("roofline", "roofline_cold", "roofline_warm"):
        assert r[key]["bound"] == "hbm" and 0 < r[key]["frac"] < 1.0
    assert r["config"]["scenes_rotated"] == 4
    # the other BASELINE configurations ride along in the same line (compact, N = 1 only)
    assert set(r["also"]) == {"2b", "5", "3", "4", "4i"}
    assert r["also"]["4i"]["live_rows_identical_to_eager"] is True
    for cfg, c in r["also"].items():
        assert "error" not in c, (cfg, c)
        assert c["value"] > 0 and c["ms_per_step"] > 0 and 0 < c["roofline"]["frac"] < 1.0


def test_bench_gpus_2_spawns_two_ranks(cuda):
    r = _run(["--gpus", "2", "--steps", "20", "--warmup", "8", "--scenes", "2"],
             {"BENCH_DIST_BACKEND": "gloo", "BENCH_ONE_DEVICE": "1", "BENCH_CHECK_EXCHANGE": "1"})
    assert r["n_gpus"] == 2
    # the overlapped exchange is numerically what it claims: bucket row u == mean over the ranks of step u's dW
    chk = r["config"]["gradient_exchange_check"]
    assert chk["ranks_have_distinct_gradients"] and chk["buckets_checked"] >= 3 and chk["max_rel_err"] < 2e-3, chk
    # the gradient exchange is off the critical path: 8 steps per replay also at N > 1, one bucket per replay
    assert r["config"]["steps_per_replay"] == 8 and "side stream" in r["config"]["gradient_exchange"]
    assert r["config"]["ranks_seen"] == 2 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["dist_backend"] == "gloo"


def test_bench_gpus_8_dry_run_on_one_device(cuda):
    """What the driver launches on an 8-GPU node (`--gpus 8`), as far as one device can take it: eight ranks over gloo
    share cuda:0 -- launcher, per-rank scenes, the two alternating 8-step graphs, eight-way flat-bucket averages, the
    max-over-ranks timing -- for the headline layer and for the backbone."""
    env = {"BENCH_DIST_BACKEND": "gloo", "BENCH_ONE_DEVICE": "1"}
    r = _run(["--gpus", "8", "--steps", "16", "--warmup", "8", "--scenes", "1", "--voxels", "30000"], env, timeout=900)
    assert r["n_gpus"] == 8 and r["config"]["ranks_seen"] == 8 and r["config"]["parallelism"] == "dp8"
    assert r["config"]["steps_per_replay"] == 8 and r["value"] > 0
    r = _run(["--gpus", "8", "--config", "4", "--steps", "2", "--warmup", "1", "--voxels", "8000", "--scenes", "1"],
             env, timeout=900)
    assert r["n_gpus"] == 8 and r["config"]["ranks_seen"] == 8 and len(r["config"]["layer_voxels"]) == 12


def test_bench_config4_two_ranks(cuda):
    r = _run(["--gpus", "2", "--config", "4", "--steps", "2", "--warmup", "1", "--voxels", "20000",
              "--scenes", "1"], {"BENCH_DIST_BACKEND": "gloo", "BENCH_ONE_DEVICE": "1"})
    assert r["n_gpus"] == 2 and r["config"]["ranks_seen"] == 2
    assert len(r["config"]["layer_voxels"]) == 12


def test_rccl_path_with_a_world_of_one_rank(cuda):
    """BENCH_SOLO_DIST=1: the N > 1 control flow -- process group brought up AFTER graph capture with
    device_id, flat gradient bucket all-reduced by the real RCCL backend on the side stream under the next replay,
    barriers, the reductions of the timing -- with a single rank (a one-GPU box cannot host two RCCL ranks)."""
    r = _run(["--steps", "24", "--warmup", "8", "--no-cpu-baseline", "--no-also", "--scenes", "2"],
             {"BENCH_SOLO_DIST": "1"})
    assert r["n_gpus"] == 1 and r["config"]["dist_backend"] == "nccl"
    assert r["config"]["steps_per_replay"] == 8 and "side stream" in r["config"]["gradient_exchange"]
    assert r["value"] > 0
    r = _run(["--config", "4", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--scenes", "2"],
             {"BENCH_SOLO_DIST": "1"})
    assert r["config"]["dist_backend"] == "nccl" and "all-reduce" in r["config"]["workload"] and r["value"] > 0
