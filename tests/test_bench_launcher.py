"""bench.py as the driver calls it: `python bench.py --gpus N` with no torchrun environment must start N fresh child
ranks (never exec / re-exec a process that has touched the GPU) and exit with their code.  CPU only."""
import importlib.util
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_command_is_the_drivers_torchrun_line():
    b = _bench()
    cmd = b.launcher_command(["--gpus", "4", "--steps", "7", "--warmup", "2"], 4, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_self_launch_propagates_the_childs_exit_code_and_sets_the_ipc_mode():
    b = _bench()
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    assert b.self_launch(["--gpus", "2"], 2, run=fake_run) == 7
    assert "--nproc-per-node=2" in seen["cmd"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert 1024 < int(seen["cmd"][seen["cmd"].index("--master-port") + 1]) < 65536


def test_main_self_launches_before_touching_the_gpu(monkeypatch):
    """--gpus 2 with WORLD_SIZE unset: main() must go to self_launch (and exit with its code) before any torch.cuda call."""
    b = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    calls = []
    monkeypatch.setattr(b, "self_launch", lambda argv, n: calls.append((argv, n)) or 5)
    monkeypatch.setattr(b.torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before launch")))
    try:
        b.main()
        raise AssertionError("main() returned")
    except SystemExit as e:
        assert e.code == 5
    assert calls == [(["--gpus", "2", "--steps", "1"], 2)]


def test_real_two_rank_launch_round_trip(tmp_path):
    """The launcher line really starts two ranks (gloo, CPU) and the exit code of a failing rank comes back non-zero."""
    b = _bench()
    ok = tmp_path / "ok.py"
    ok.write_text("import os, torch.distributed as d\nd.init_process_group('gloo')\n"
                  "assert d.get_world_size() == 2 and int(os.environ['WORLD_SIZE']) == 2\nd.barrier()\nd.destroy_process_group()\n")
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    assert subprocess.run(b.launcher_command([], 2, b.free_port(), script=str(ok)), env=env, timeout=300).returncode == 0
    assert subprocess.run(b.launcher_command([], 2, b.free_port(), script=str(bad)), env=env, timeout=300).returncode != 0
