"""CPU: the argument plumbing of bench.py -- `--gpus N` starts its own ranks.

The contract verb is `python bench.py --gpus N`; under a launcher (WORLD_SIZE set) the process is a rank, without one and
with N > 1 it starts `torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD (never an exec, and
before torch or HIP has been touched) and hands its exit code on.  No GPU here: the ranks themselves end with
"needs an MI355X", which is what the end-to-end case below looks for."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_a_single_rank_goes_on():
    assert bench.launch_ranks(1, [], environ={}) is None
    assert bench.launch_ranks(1, ["--steps", "3"], environ={"WORLD_SIZE": "1"}) is None


def test_under_a_launcher_the_process_is_a_rank():
    called = []
    assert bench.launch_ranks(8, ["--gpus", "8"], environ={"WORLD_SIZE": "8", "RANK": "3"}, run=lambda *a, **k: called.append(a)) is None
    assert not called


def test_world_size_and_gpus_must_agree(capsys):
    assert bench.launch_ranks(8, ["--gpus", "8"], environ={"WORLD_SIZE": "1"}) == 2
    assert bench.launch_ranks(1, [], environ={"WORLD_SIZE": "2"}) == 2
    assert "WORLD_SIZE" in capsys.readouterr().err


def test_gpus_n_without_a_launcher_starts_n_ranks():
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    rc = bench.launch_ranks(4, argv, environ={"PATH": os.environ.get("PATH", ""), "TDOA_BENCH_BACKEND": "gloo"}, run=fake_run)
    assert rc == 7                                                # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                    # the same arguments, --gpus included
    assert seen["env"]["TDOA_BENCH_BACKEND"] == "gloo" and "WORLD_SIZE" not in seen["env"]


def test_a_given_master_port_is_kept():
    seen = {}
    bench.launch_ranks(2, [], environ={"MASTER_PORT": "29777"}, run=lambda cmd, env=None: seen.setdefault("cmd", cmd) and 0)
    assert seen["cmd"][seen["cmd"].index("--master-port") + 1] == "29777"


def test_end_to_end_without_a_gpu_the_ranks_start_and_refuse():
    """`python bench.py --gpus 2` here: two ranks come up under torch.distributed.run, each finds no HIP device and says so;
    the parent hands the launcher's non-zero status on and prints no JSON line"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "needs an MI355X" in r.stderr
    assert r.stdout.strip() == ""
