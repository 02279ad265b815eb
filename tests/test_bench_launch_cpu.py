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


class Args:
    gpus, print_launch, launch_timeout, phase, one_launch = 2, False, 30.0, "all", False
    no_sub, shared_frame, sub_timeout = False, False, 5.0


def _stand_in(tmp_path, monkeypatch, body):
    """Replace the launcher by a child whose behaviour depends on the --phase it is started with."""
    import bench
    script = tmp_path / "child.py"
    script.write_text("import sys, time, json\nphase = sys.argv[sys.argv.index('--phase') + 1] if '--phase' in sys.argv else 'all'\n" + body)
    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port: [sys.executable, str(script)] + list(argv))


def test_self_launch_relays_line_and_exit_code(tmp_path, monkeypatch, capsys):
    """--one-launch / a named phase: ONE launch, its JSON line goes to stdout, other output to stderr, the exit code is the
    child's; no line and exit 0 is an error."""
    import bench

    class A(Args):
        one_launch = True

    def fake(lines, rc):
        _stand_in(tmp_path, monkeypatch, "".join("print(%r)\n" % l for l in lines) + "sys.exit(%d)\n" % rc)

    fake(["noise", '{"value": 1.0}'], 0)
    assert bench.self_launch(A, []) == 0
    assert json.loads(capsys.readouterr().out.splitlines()[-1]) == {"value": 1.0}
    fake(['{"value": null, "watchdog": "x"}'], 3)
    assert bench.self_launch(A, []) == 3
    fake(["nothing"], 0)
    assert bench.self_launch(A, []) == 1


def test_one_launch_per_phase_merged_into_one_line(tmp_path, monkeypatch, capsys):
    """The default for a plain `--gpus N`: the main timed pass in a launch of its own FIRST, then one launch per
    sub-measurement; their objects end up in one line, in the places rounds 3-5 had them."""
    import bench
    _stand_in(tmp_path, monkeypatch, textwrap.dedent("""
        open(sys.argv[0] + '.order', 'a').write(phase + '\\n')
        if phase == 'main':
            print(json.dumps({"metric": "m", "value": 7.0, "n_gpus": 2, "config": {"workload": "w"}, "ranks": {"world_size": 2}}))
        elif phase == 'exchange':
            print(json.dumps({"phase": phase, "per_iteration_exchange": {"mrays_per_s": 3.0, "ratio": 0.9}}))
        elif phase == 'strong':
            print(json.dumps({"phase": phase, "strong": {"mrays_per_s": 5.0}}))
    """))
    assert bench.self_launch(Args, ["--gpus", "2"]) == 0
    d = json.loads(capsys.readouterr().out.splitlines()[-1])
    assert d["value"] == 7.0 and d["ranks"]["world_size"] == 2
    assert d["per_iteration_exchange"] == d["config"]["per_iteration_exchange"] == {"mrays_per_s": 3.0, "ratio": 0.9}
    assert d["config"]["strong"] == {"mrays_per_s": 5.0} and d["phases"]["order"] == ["main", "exchange", "strong"]
    assert open(str(tmp_path / "child.py") + ".order").read().split() == ["main", "exchange", "strong"]


def test_a_phase_that_hangs_or_dies_does_not_lose_the_main_number(tmp_path, monkeypatch, capsys):
    """The per-iteration exchange never returns (its launch is ended by the time limit: exit code 124), strong scaling
    dies without a line: the line still carries the main pass's value, both failures are stated, exit code 0."""
    import bench

    class A(Args):
        launch_timeout = 3.0

    _stand_in(tmp_path, monkeypatch, textwrap.dedent("""
        if phase == 'main':
            print(json.dumps({"value": 9.0, "config": {}}), flush=True)
        elif phase == 'exchange':
            time.sleep(600)
        else:
            sys.exit(7)
    """))
    assert bench.self_launch(A, ["--gpus", "2"]) == 0
    d = json.loads(capsys.readouterr().out.splitlines()[-1])
    assert d["value"] == 9.0
    assert "time limit" in d["per_iteration_exchange"]["failed"] and "exit code 7" in d["config"]["strong"]["failed"]


def test_main_pass_lost_is_said_and_the_exit_code_is_the_launchers(tmp_path, monkeypatch, capsys):
    import bench
    _stand_in(tmp_path, monkeypatch, "sys.exit(3 if phase == 'main' else 0)\n")
    assert bench.self_launch(Args, ["--gpus", "2"]) == 3
    d = json.loads(capsys.readouterr().out.splitlines()[-1])
    assert d["value"] is None and "failed" in d and "failed" in d["config"]["strong"]


def test_phases_run_under_gloo_in_their_own_launches():
    """The real thing without a GPU as far as it goes: `--phase exchange` under torch.distributed.run with two gloo ranks
    must fail LOUDLY per rank (bench.py needs a GPU) -- exit code non-zero, no line -- and the parent turns that into a
    stated failure instead of hanging or printing nothing."""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "240", "--sub-timeout", "20"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(lines[0])
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and d["value"] is None and "failed" in d["per_iteration_exchange"] and "failed" in d["config"]["strong"]
    else:
        assert p.returncode == 0 and d["value"] > 0 and d["phases"]["order"] == ["main", "exchange", "strong"]


def test_watchdog_exit_code_follows_the_main_pass():
    """A rank's own watchdog (N > 1 under an external launcher, where the phases share one set of processes): a phase that
    hangs BEFORE the main timed pass has left `value` ends the rank with exit code 3; one that hangs after it, with 0 -- the
    line is printed either way, with `watchdog` naming the phase."""
    code = textwrap.dedent("""
        import sys, time, json
        import bench
        out = {"value": VALUE, "config": {}}
        g = bench.Watchdog(0, out)
        g.arm(0.2, "per-iteration exchange")
        time.sleep(30)
    """)
    for value, rc in (("None", 3), ("7.0", 0)):
        p = subprocess.run([sys.executable, "-c", code.replace("VALUE", value)], cwd=ROOT, capture_output=True, text=True, timeout=60)
        assert p.returncode == rc, (value, p.returncode, p.stderr[-500:])
        d = json.loads(p.stdout.strip().splitlines()[-1])
        assert "per-iteration exchange" in d["watchdog"] and d["value"] == (None if value == "None" else 7.0)

