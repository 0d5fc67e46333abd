"""bench.py --gpus N starts its ranks itself (one process per GPU). The launcher must notice a dead rank:
a rank blocked in a collective whose partner died would otherwise wait for ever. CPU-only: the ranks here are stub scripts."""
import argparse
import os
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(tmp_path, body, gpus=2, timeout=60.0):
    script = tmp_path / "rank_stub.py"
    script.write_text("import json, os, sys, time\nrank = int(os.environ['RANK'])\n" + textwrap.dedent(body))
    args = argparse.Namespace(gpus=gpus, dist_timeout=timeout)
    t0 = time.time()
    rc = bench.launch_ranks(args, [], script=str(script), poll_s=0.05)
    return rc, time.time() - t0


@pytest.mark.timeout(120)
def test_dead_rank_ends_the_job(tmp_path, capfd):
    rc, dt = run(tmp_path, """
        if rank == 1:
            time.sleep(0.5)
            sys.stderr.write("rank 1 is going down\\n")
            sys.exit(3)
        time.sleep(300)   # rank 0: 'blocked in the all-gather'
    """)
    assert rc == 3 and dt < 30
    err = capfd.readouterr().err
    assert "rank 1 exited with code 3" in err and "rank 1 is going down" in err


@pytest.mark.timeout(120)
def test_timeout_terminates_all_ranks(tmp_path):
    rc, dt = run(tmp_path, "time.sleep(300)\n", timeout=1.5)
    assert rc == 124 and dt < 30


@pytest.mark.timeout(120)
def test_killed_rank_is_reported(tmp_path):
    rc, dt = run(tmp_path, """
        import signal
        if rank == 1:
            time.sleep(0.3)
            os.kill(os.getpid(), signal.SIGKILL)
        time.sleep(300)
    """)
    assert rc == 9 and dt < 30


@pytest.mark.timeout(120)
def test_json_line_is_checked(tmp_path, capfd):
    ok = """
        assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'
        if rank == 0:
            print(json.dumps({"value": 1.0, "config": {"ranks": 2, "collective_backend": %r, "devices_visible": %d}}))
    """
    assert run(tmp_path, ok % ("rccl", 8))[0] == 0
    assert '"ranks": 2' in capfd.readouterr().out
    assert run(tmp_path, ok % ("gloo", 1))[0] == 0          # ranks sharing one GPU: gloo is the only option
    assert run(tmp_path, ok % ("gloo", 8))[0] == 1          # a silent gloo fallback on a multi-GPU node is an error
    assert run(tmp_path, "pass\n")[0] == 1                   # no JSON line at all
