"""Drop-in command line: the entry scripts accept every flag of the reference's parsers with the same
types / defaults / choices (tests/golden/g13_cli_flags.json, produced from the reference by oracle/gen_goldens.py) and
parse the invocations of the reference's launch scripts (runs/pretrain.sh, scratch.sh, finetune.sh, linear.sh) into the
same namespaces.  Host logic around them: run directories, schedulers, early stop, rank sharding."""
import json
import math
import os

import pytest
import torch

from oracle.recipes import run_script_argvs

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g13():
    with open(os.path.join(HERE, "golden", "g13_cli_flags.json")) as f:
        return json.load(f)


def _parser(entry):
    import importlib
    return importlib.import_module(f"murcl_amd.{entry}").build_parser()


@pytest.mark.parametrize("entry", ["train_MuRCL", "train_RLMIL"])
def test_every_reference_flag_exists_with_same_type_default_choices(g13, entry):
    mine = {a.dest: a for a in _parser(entry)._actions}
    for row in g13["flags"][entry]:
        a = mine.get(row["dest"])
        assert a is not None, f"{entry}: flag {row['options']} missing"
        assert list(a.option_strings) == row["options"]
        assert getattr(a.type, "__name__", None) == row["type"], row["dest"]
        assert a.default == row["default"] and type(a.default) is type(row["default"]), (row["dest"], a.default, row["default"])
        assert (list(a.choices) if a.choices is not None else None) == row["choices"], row["dest"]
        assert type(a).__name__ == row["action"] and a.nargs == row["nargs"], row["dest"]
    extras = set(mine) - {r["dest"] for r in g13["flags"][entry]} - {"help"}
    assert extras <= {"synthetic", "num_clusters", "dtype", "no_batched_stage1", "no_resident", "global_mixup", "dist_backend"}, extras


@pytest.mark.parametrize("script", ["pretrain.sh", "scratch.sh", "finetune.sh", "linear.sh"])
def test_launch_script_invocations_parse_to_the_reference_namespaces(g13, script):
    entry, argvs = run_script_argvs()[script]
    assert g13["runs"][script]["entry"] == entry and len(argvs) == 3
    p = _parser(entry)
    for av, want in zip(argvs, g13["runs"][script]["namespaces"]):
        got = vars(p.parse_args(av))
        for k, v in want.items():
            assert got[k] == v and type(got[k]) is type(v), (script, k, got[k], v)


def test_run_directory_names_and_increment(tmp_path):
    """train_MuRCL.py:18-55 / train_RLMIL.py:20-58: the directory encodes the hyper-parameters; an existing one is
    re-used with --exist_ok and numbered otherwise (utils/general.py:42-53)."""
    from murcl_amd.utils import general as G
    a = _parser("train_MuRCL").parse_args(["--arch", "ABMIL", "--base_save_dir", str(tmp_path), "--train_stage", "2"])
    assert G.run_dir_name(a, "MuRCL") == str(tmp_path / "Camelyon16_np_1024" / "MuRCL" / "T6_pd128_as0.5_pg0.1_tau1.0_alpha0.9" /
                                             "ABMIL" / "L512_D128_dpt0.0" / "exp" / "seed985" / "stage_2")
    r = _parser("train_RLMIL").parse_args(["--base_save_dir", str(tmp_path), "--train_method", "linear", "--save_dir_flag", "x"])
    assert G.run_dir_name(r, "RLMIL") == str(tmp_path / "Camelyon16_np_1024" / "RLMIL" / "T6_as0.5_pg0.1_phd512_fhd1024" / "CLAM_SB" /
                                             "size_small_ks_8_bw_0.7" / "linear" / "exp_x" / "seed985" / "stage_1")
    d = G.prepare_run_dir(a, "MuRCL")
    assert os.path.isdir(d) and d.endswith("stage_2")
    a.save_dir = None
    assert G.prepare_run_dir(a, "MuRCL") == d + "_2"                     # exists and no --exist_ok: next free name
    a.save_dir, a.exist_ok = None, True
    assert G.prepare_run_dir(a, "MuRCL") == d
    a.save_dir = "given/stage_1"                                         # an explicit --save_dir lives under --base_save_dir
    assert G.prepare_run_dir(a, "MuRCL") == str(tmp_path / "given" / "stage_1")
    G.dump_args(a, a.save_dir)
    assert os.path.exists(os.path.join(a.save_dir, "args.yaml"))


def test_schedules_equal_torch_schedulers():
    """optim.LRSchedule (closed forms) == torch's StepLR(7, 0.1) / CosineAnnealingLR(T_max, 1e-6) (train_MuRCL.py:174-186)."""
    from murcl_amd.optim import LRSchedule

    class _Opt:
        def __init__(self, lrs):
            self.param_groups = [{"lr": v, "initial_lr": v} for v in lrs]

    for name, make in (("StepLR", lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=7, gamma=0.1)),
                       ("CosineAnnealingLR", lambda o: torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=40, eta_min=1e-6))):
        ps = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        topt = torch.optim.Adam([{"params": [ps[0]], "lr": 1e-4}, {"params": [ps[1]], "lr": 5e-5}])
        ts = make(topt)
        mine = _Opt([1e-4, 5e-5])
        ms = LRSchedule(mine, name, T_max=40)
        for _ in range(40):
            topt.step()
            ts.step()
            ms.step()
            for g, w in zip(mine.param_groups, topt.param_groups):
                assert math.isclose(g["lr"], w["lr"], rel_tol=1e-9, abs_tol=1e-15), (name, g["lr"], w["lr"])


def test_early_stop_best_and_mean_semantics():
    from murcl_amd.utils import general as G
    es = G.EarlyStop(3)
    for v, stop in ((1.0, False), (1.0, False), (1.0, True)):
        es.update(v)
        assert es.is_stop() == stop
    es.update(0.5)
    assert not es.is_stop()
    b = G.Best("min")
    assert b.compare(2.0, 1, inplace=True) and not b.compare(3.0, 2, inplace=True) and (b.best, b.epoch) == (2.0, 1)
    m = G.Mean()
    m.update(1.0, 3), m.update(5.0, 1)
    assert m.avg == 2.0


def test_rank_shards_are_equal_sized_and_disjoint():
    """ranks must run the same number of steps per epoch - shards are trimmed to len // world slides each."""
    from murcl_amd.train_MuRCL import shard_slides
    for n, w in ((1023, 8), (64, 8), (9, 2), (7, 7)):
        shards = [shard_slides(n, r, w) for r in range(w)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == n // w
        flat = [i for s in shards for i in s]
        assert len(set(flat)) == len(flat) and max(flat) < n
    with pytest.raises(ValueError):
        shard_slides(3, 0, 8)


def test_device_flag_rejects_cpu():
    from murcl_amd.utils import general as G
    with pytest.raises(RuntimeError, match="no CPU path"):
        G.pick_device("cpu")


def test_device_flag_never_seats_two_local_ranks_on_one_gpu(monkeypatch):
    """A --device list shorter than the local world (the inherited default "3" under an N-rank launcher) must not wrap around:
    rank r takes cuda:r; a full list is indexed by the local rank."""
    import torch
    from murcl_amd.utils import general as G
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert [G.pick_device("3", r).index for r in range(4)] == [0, 1, 2, 3]
    assert [G.pick_device("4,5,6,7", r).index for r in range(4)] == [4, 5, 6, 7]
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert G.pick_device("3", 0).index == 3
