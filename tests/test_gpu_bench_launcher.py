"""The driver starts the multi-GPU bench as the BARE command `python bench.py --gpus N ...`
(no torchrun around it).  bench.py must then start its N ranks itself -- as child processes of a
parent that never touched the GPU -- forward rank 0's single JSON line and return the ranks' exit
code.  On the one-GPU box the ranks share cuda:0 and the exchange runs over gloo
(TC_BENCH_REHEARSAL=1: control flow only, never a measurement); the 8-GPU RCCL run is the driver's.
SURVEY.md 8(e); the only parallelism this replaces is FMIndex.hs:417-423."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.gpu
def test_bare_command_launches_two_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--n", str(1 << 24)], {"TC_BENCH_REHEARSAL": "1"})
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    g = out["gather"]
    assert g["ranks_in_communicator"] == 2 and g["containers_verified"] == 2 and len(g["container_bytes"]) == 2
    assert out["value"] > 0 and out["config"]["records"] == 2


@pytest.mark.gpu
def test_rank_failure_propagates():
    """a rank that dies must turn into a non-zero exit code of the bare command"""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--n", str(1 << 20)],
             {"TC_BENCH_REHEARSAL": "1", "TC_BENCH_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_stalled_first_exchange_falls_back_within_the_deadline():
    """a rank that never reaches its first exchange: the parent kills the ranks at its deadline, starts fresh ones with
    the conservative exchange, and the line says which path ran and why"""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--n", str(1 << 22)],
             {"TC_BENCH_REHEARSAL": "1", "TC_BENCH_STALL_RANK": "1", "TC_BENCH_DEADLINE_S": "60"}, timeout=400)
    took = time.time() - t0
    assert p.returncode == 0, p.stderr[-4000:]
    assert took < 60 + 60 + 30, took
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    g = json.loads(lines[0])["gather"]
    assert g["path"] == "torch" and "deadline" in g["fallback_reason"], g
    assert g["containers_verified"] == 2 and g["exchange_wait_ms"]["calls"] >= 1
    assert "starting fresh ranks" in p.stderr


def test_parent_does_not_touch_gpu_or_torch():
    """the launching parent imports neither torch nor the library before it spawns the ranks"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main()")]
    assert "import torch" not in head.replace("torch.distributed.run", "")
    body = src[src.index("def main()"):]
    assert body.index("self_launch(a)") < body.index("import torch")
    assert "os.exec" not in src


# ---- the parent's deadline / fallback logic on its own (no GPU, no torch: any child command) ----------------------------
_CHILD = ("import os, sys, time\n"
          "mode = os.environ['TC_TEST_CHILD']\n"
          "fb = 'TC_BENCH_FALLBACK_REASON' in os.environ\n"
          "if mode == 'stall_first' and not fb: time.sleep(1000)\n"
          "if mode == 'stall_always': time.sleep(1000)\n"
          "if mode == 'die_first' and not fb: sys.exit(7)\n"
          "print('{\"path\": \"%s\", \"gather\": \"%s\", \"cus\": \"%s\", \"why\": \"%s\"}' % ('fallback' if fb else 'first', "
          "os.environ.get('TC_BENCH_GATHER', ''), os.environ.get('TC_COMM_CUS', ''), os.environ.get('TC_BENCH_FALLBACK_REASON', '')))\n")


def _parent(mode, deadline):
    import time
    child = os.path.join(ROOT, "gpurun_out", "_launcher_child.py")
    os.makedirs(os.path.dirname(child), exist_ok=True)
    with open(child, "w") as f:
        f.write(_CHILD)
    t0 = time.time()
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"],
             {"TC_BENCH_CHILD_CMD": "%s %s" % (sys.executable, child), "TC_TEST_CHILD": mode, "TC_BENCH_DEADLINE_S": str(deadline)}, timeout=120)
    return p, time.time() - t0


def test_parent_falls_back_after_a_stall_cpu():
    p, took = _parent("stall_first", 3)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["path"] == "fallback" and out["gather"] == "torch" and out["cus"] == "0" and "deadline" in out["why"]
    assert took < 3 + 15 and "starting fresh ranks" in p.stderr


def test_parent_falls_back_after_a_dead_rank_cpu():
    p, _ = _parent("die_first", 20)
    assert p.returncode == 0
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["path"] == "fallback" and "exit code 7" in out["why"]


def test_parent_gives_up_after_two_stalls_cpu():
    p, took = _parent("stall_always", 2)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert took < 2 * (2 + 12) + 5 and "fallback ranks failed as well" in p.stderr


def test_first_attempt_line_is_forwarded_unchanged_cpu():
    p, _ = _parent("ok", 20)
    assert p.returncode == 0 and json.loads(p.stdout.strip())["path"] == "first" and "starting fresh ranks" not in p.stderr

