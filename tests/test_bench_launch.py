"""`bench.py --gpus N` starts N ranks itself (the flag used to be dead).  CPU test of the
launch path: the parent must spawn the ranks before anything touches a GPU, rank 0 prints ONE JSON line with n_gpus = N, a
mismatch between --gpus and a torchrun environment fails loudly.  The engine is replaced by a sleeping stand-in (hidden
--stub-engine flag, gloo instead of RCCL): what is exercised is argument handling, torch.distributed.run, barriers,
max-over-ranks and the output contract -- not a measurement."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_2_launches_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-engine"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["stub"] is True and "STUB" in out["config"]["workload"]
    assert out["config"]["ranks_seen"] == 2 and out["value"] > 0 and out["higher_is_better"] is True
    for key in ("metric", "unit", "ms_per_step", "vs_baseline", "dtype", "data", "config"):
        assert key in out


def test_gpus_mismatch_with_torchrun_environment_fails_loudly():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--stub-engine"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_parent_does_not_touch_the_gpu_before_spawning():
    """the launcher branch sits above load_package() / torch imports in main()"""
    src = (ROOT / "bench.py").read_text()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args)") < main.index("ge.load_package()") < main.index("capi.Engine") if "capi.Engine" in main else True
    assert main.index("launch_ranks(args)") < main.index("import torch")
    launcher = src[src.index("def launch_ranks"):src.index("class _StubEngine")]
    assert "os.exec" not in launcher and "subprocess.run" in launcher
