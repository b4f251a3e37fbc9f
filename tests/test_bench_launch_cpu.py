"""bench.py --gpus N outside a launcher starts its own rank processes: the tensor-parallel degree
arrives as a bare integer and nothing upstream starts ranks
(/root/reference/benchmark-vllm/vllm_templates.py:55)."""
import json
import os
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_importing_bench_pulls_in_no_gpu_runtime():
    # the parent of the ranks must not have initialised a GPU: bench.py imports torch only inside main()
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; "
                          "print('torch' in sys.modules, 'mi355_decode' in sys.modules)" % ROOT],
                         capture_output=True, text=True, check=True).stdout.split()
    assert out == ["False", "False"]


def test_ranks_get_their_environment_and_rank0_is_relayed(tmp_path):
    cmd = _script(tmp_path, """
        import json, os, sys
        e = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                            "HSA_ENABLE_IPC_MODE_LEGACY")}
        open(os.path.join(os.path.dirname(__file__), "rank%s.json" % e["RANK"]), "w").write(json.dumps(e))
        print(json.dumps({"rank": int(e["RANK"])}))
    """)
    rc, out = bench.launch_ranks(3, cmd)
    assert rc == 0
    assert json.loads(out.strip()) == {"rank": 0}          # only rank 0's stdout comes back
    envs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0
    assert {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}


def test_a_failing_rank_ends_the_job_non_zero(tmp_path):
    cmd = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)            # a rank waiting in a collective for the one that died
    """)
    t0 = time.time()
    rc, _ = bench.launch_ranks(2, cmd)
    assert rc == 7 and time.time() - t0 < 20


def test_timeout_ends_the_job(tmp_path):
    cmd = _script(tmp_path, "import time; time.sleep(60)\n")
    t0 = time.time()
    rc, _ = bench.launch_ranks(2, cmd, timeout=1.0)
    assert rc == 124 and time.time() - t0 < 20


def test_gpus_n_without_launcher_spawns_and_relays(monkeypatch, capsys):
    calls = []

    def fake(n, cmd, env=None, timeout=None):
        calls.append((n, cmd))
        return 0, '[Gloo] chatter\n{"metric": "decode_tok_s"}\n'

    monkeypatch.setattr(bench, "launch_ranks", fake)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0
    (n, cmd), = calls
    assert n == 4 and cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
    cap = capsys.readouterr()
    assert cap.out == '{"metric": "decode_tok_s"}\n' and "[Gloo] chatter" in cap.err
