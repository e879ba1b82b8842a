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

    monkeypatch.setattr(b, "count_gpus", lambda: 8)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("the parent counts GPUs from sysfs")))
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
    monkeypatch.setattr(b, "count_gpus", lambda: 1)
    assert b.self_launch(8, ["--gpus", "8"]) == 2
    assert "--gpus 8 but this node has 1 GPU" in capsys.readouterr().err


def test_gpus_are_counted_from_the_kfd_topology_without_a_runtime_call(tmp_path, monkeypatch):
    """count_gpus reads /sys/class/kfd/kfd/topology/nodes/*/properties: nodes with simd_count > 0 are GPUs (CPU nodes report 0);
    a visibility mask caps the count; only a node without readable sysfs falls back to torch."""
    import torch
    b = _bench()
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("no runtime call when sysfs is readable")))
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert b.count_gpus(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert b.count_gpus(str(tmp_path)) == 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 5)
    assert b.count_gpus(str(tmp_path / "missing")) == 5


@pytest.mark.gpu
def test_forced_distributed_run_carries_the_comm_block():
    """MURCL_FORCE_DIST=1: the RCCL code path with one rank; the bench line then carries `comm`: HIP-event time
    of the z all-gather and of each gradient all-reduce, and the exposed (non-overlapped) part of a step."""
    env = dict(os.environ, MURCL_FORCE_DIST="1", MASTER_PORT="29533")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--stat-steps", "0",
                        "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = out["comm"]
    assert "error" not in c, c
    assert c["ranks"] == 1 and c["z_all_gather_us"] > 0 and c["z_all_gather_bytes_per_rank"] == 2 * 64 * 128 * 4
    assert set(c["grad_all_reduce_us"]) == {"group0", "group1"} and all(v["us"] > 0 and v["bytes"] > 0 for v in c["grad_all_reduce_us"].values())
    assert c["step_us_with_collectives"] > 0 and c["step_us_local_only"] > 0
    assert abs(c["exposed_us_per_step"] - (c["step_us_with_collectives"] - c["step_us_local_only"])) < 0.2
    assert "env" in c["rccl"]


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
    # round 5: the bound comes from SURVEY 8(d) (the encoder GEMMs are MFMA-bound), both fractions are reported, the step as a whole
    # is priced against the MFMA peak and the fused minimum of HBM bytes, K2 includes its combine launch, traffic names its run
    r, k2 = out["roofline"], out["roofline_k2"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_flops", "frac_layer_bytes", "traffic_source", "step"):
        assert k in r, k
    if r["kernel"].startswith(("panel_gemm", "gemm_tn")):
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["frac_flops"]) < 1e-9
    st = r["step"]
    assert st["flops"] == pytest.approx(2 * 64 * 9.41e9) and 0 < st["frac_of_mfma"] < 1
    assert st["fused_min_bytes"] == 2 * 64 * 2048 * 512 * 2
    assert st["hbm_bytes_pmc"] is None or (st["ratio_vs_fused_min"] > 1 and st["pmc_source"])
    # round 6: two timed regions of the same K steps - eager (host-paced when the host is loaded) and hipGraph replays with LIVE
    # optimizer state (GPU-paced); `value` is the graph region when it exists, the eager one is reported beside it
    assert out["timing_mode"].startswith("hipgraph") and out["graph"]["replays"] * out["graph"]["steps_per_graph"] == 3
    assert out["graph"]["value_graph"] == out["value"] and out["ms_per_step_graph"] == out["graph"]["ms_per_step_graph"]
    assert abs(out["ms_per_step_graph"] - out["ms_per_step"]) < 1e-3
    assert out["eager"]["value"] > 0 and out["ms_per_step_eager"] == out["eager"]["ms_per_step"]
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 / 64 - 1) < 1e-3
    # round 6: the box is calibrated in the same process (copy rate + bf16 MFMA rate of THIS box), fractions against it beside the
    # fractions against the data-sheet peaks
    box = out["box"]
    assert 2000 < box["copy_GBps"] < 8000 and 800 < box["mfma_bf16_TFLOPs"] < 2600, box
    assert 0 < r["frac_of_box_copy"] < 1.5 and 0 < k2["frac_of_box_copy"] < 1.2
    # round 6: the K2 row is ONE launch - its per-bag merge runs inside the decoder launch (no abmil_pool_combine in the step)
    assert k2["bound"] == "hbm" and k2["launches_of_the_row"] == ["abmil_pool_fwd<bf16>"]
    assert len(k2["avg_ms_each_untimed_pass"]) == 1 and k2["avg_launch_ms"] > 0
    assert k2["merge_launch"]["kernel"] == "abmil_pool_decoder" and k2["merge_launch"]["avg_ms_untimed_pass"] > 0
    assert "abmil_pool_combine" not in out["kernel_ms_per_step"] and "abmil_pool_decoder" in out["kernel_ms_per_step"]
