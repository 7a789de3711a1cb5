"""bench.py's launcher logic, without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher environment must start N ranks itself (children of
    torch.distributed.run, rendezvous on 127.0.0.1) instead of quietly measuring one GPU; the parent
    decides before it imports torch or touches a GPU."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    import torch
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
        raise AssertionError("bench.main() should leave with the children's exit code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_under_a_launcher_does_not_spawn():
    """With WORLD_SIZE set (the driver's torch.distributed.run) bench.py is a rank, not a launcher: it goes on
    to the GPU check (and fails it here, loudly -- there is no CPU fallback)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591")
    p = subprocess.run([sys.executable, "-c",
                        "import sys; sys.argv=['bench.py','--gpus','2'];\n"
                        "import bench, subprocess\n"
                        "def no(*a, **k): raise AssertionError('spawned')\n"
                        "subprocess.call = no\n"
                        "import sina_amd.dist as d; d.init=lambda backend=None:(0,0,2,None)\n"
                        "bench.main()"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs a GPU" in p.stderr and "spawned" not in p.stderr


def test_bench_gpus_n_on_a_node_with_fewer_gpus_says_so(monkeypatch, capsys):
    """More ranks than the node has GPUs: a clear message and a non-zero exit before anything is spawned."""
    sys.path.insert(0, ROOT)
    import bench
    import torch

    def no_spawn(*a, **k):
        raise AssertionError("spawned")

    monkeypatch.setattr(subprocess, "call", no_spawn)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
        raise AssertionError("should have left")
    except SystemExit as e:
        assert e.code == 2
    assert "shows 1 GPU" in capsys.readouterr().err
