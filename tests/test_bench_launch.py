"""bench.py --gpus N starts its own ranks: the parent makes no GPU call, spawns the driver's own
torch.distributed.run command line as a child, relays its exit code."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_argv_is_the_drivers_command_line():
    b = _bench()
    argv = b.launch_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29617)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29617"
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_parent_spawns_children_and_relays_their_exit_code(monkeypatch):
    import torch
    b = _bench()
    calls = []

    class _Done:
        returncode = 7

    def fake_run(argv, env=None, cwd=None):
        calls.append((argv, env, cwd))
        return _Done()

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("the parent must not touch the GPU")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 7 and len(calls) == 1
    argv, env, cwd = calls[0]
    assert "--nproc-per-node=4" in argv and argv[-4:] == ["--gpus", "4", "--steps", "3"] and cwd == ROOT
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "WORLD_SIZE" not in env


def test_parent_refuses_more_ranks_than_gpus(monkeypatch, capsys):
    import torch
    b = _bench()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert b.self_launch(8, ["--gpus", "8"]) == 2
    assert "--gpus 8 but this node has 1 GPU" in capsys.readouterr().err


@pytest.mark.gpu
def test_gpus_1_through_the_spawn_path_prints_the_contract_line():
    env = dict(os.environ, MURCL_BENCH_SPAWN="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--stat-steps", "0", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["roofline"]["frac"] > 0
