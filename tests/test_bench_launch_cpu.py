"""bench.py started from a plain shell with --gpus N > 1 (no WORLD_SIZE): the ranks are started as CHILD processes
through torch.distributed.run before anything touches the GPU, rank 0's JSON line and the launcher's exit code are
relayed.  Here, without a GPU: the command line that is built, that the parent imports neither
torch nor the HIP library, and that a failing rank's exit code and a rank's JSON line both come back."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_command_is_the_drivers_form():
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29555)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


def test_print_launch_touches_neither_torch_nor_the_library():
    """The parent of a self-launch must not have initialised the GPU: it imports neither torch nor ctypes-loads the HIP
    library before it starts the ranks."""
    code = textwrap.dedent("""
        import sys
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "2", "--print-launch"]
        import bench
        try:
            bench.main()
        except SystemExit as e:
            assert e.code in (0, None), e.code
        assert "torch" not in sys.modules, "the launching parent imported torch"
        assert "ptmi355" not in sys.modules, "the launching parent loaded the package"
    """)
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "--print-launch" not in d["launch"] and d["launch"][d["launch"].index("--nproc-per-node") + 1] == "4"


def test_self_launch_relays_line_and_exit_code(tmp_path, monkeypatch):
    """self_launch() with the launcher replaced by a stand-in child: the JSON line goes to stdout, other output to
    stderr, the exit code is the child's; no line and exit 0 is an error."""
    import bench

    class A:
        gpus, print_launch, launch_timeout = 2, False, 30.0

    def fake(lines, rc):
        script = tmp_path / "child.py"
        script.write_text("import sys\n" + "".join("print(%r)\n" % l for l in lines) + "sys.exit(%d)\n" % rc)
        monkeypatch.setattr(bench, "launch_command", lambda n, argv, port: [sys.executable, str(script)])

    fake(["noise", '{"value": 1.0}'], 0)
    assert bench.self_launch(A, []) == 0
    fake(['{"value": null, "watchdog": "x"}'], 3)
    assert bench.self_launch(A, []) == 3
    fake(["nothing"], 0)
    assert bench.self_launch(A, []) == 1
