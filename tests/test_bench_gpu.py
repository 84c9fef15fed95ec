"""bench.py on the GPU box: the JSON contract of every mode, and the multi-rank code path (torch.distributed over RCCL,
world size 1 forced through FRP_BENCH_FORCE_DIST) with the HIP solve inside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, dist=False):
    env = dict(os.environ)
    if dist:
        env.update(FRP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--repeats", "2", "--no-cpu", *flags],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_is_the_serial_single_stream_rate():
    d = _bench("--batch", "512")
    assert d["unit"] == "solves/s" and d["scaling"] == "weak" and d["n_gpus"] == 1 and d["dtype"] == "f64"
    assert d["config"]["converged_frac"] == 1.0 and d["config"]["baseline_config"] == 2
    # serial launches: a step cannot be shorter than the kernel it consists of (the pipelined rate is reported separately)
    assert d["ms_per_step"] >= d["roofline"]["kernel_ms"] > 0.5 * d["ms_per_step"]
    assert d["roofline"]["kernel_launches_timed"] == 2 * 2  # every launch of the timed regions (--steps 2 --repeats 2)
    assert d["config"]["pipelined_solves_per_s"] >= 0.8 * d["value"]
    assert d["roofline"]["bound"] == "fp64-issue" and 0.0 < d["roofline"]["frac"] < 1.0


@pytest.mark.parametrize("cfg", [2, 3])
def test_strong_scaling_step_scatters_solves_and_gathers_over_rccl(cfg):
    d = _bench("--config", str(cfg), "--scaling", "strong", "--batch", "96", dist=True)
    assert d["scaling"] == "strong" and d["config"]["batch_total"] == 96 and d["config"]["batch_per_gpu"] == 96
    ph = d["config"]["strong_scaling_phases"]
    assert ph["solve_ms"] > 0 and ph["scatter_ms"] >= 0 and ph["gather_ms"] >= 0
    assert d["config"]["converged_frac"] > (0.99 if cfg == 2 else 0.7)


def test_monte_carlo_receding_horizon_mode():
    d = _bench("--config", "4", "--batch", "256", dist=True)
    assert d["config"]["baseline_config"] == 4 and d["config"]["converged_frac"] == 1.0
    assert d["config"]["mean_ipm_iterations"] < 6.0  # warm-started ticks
