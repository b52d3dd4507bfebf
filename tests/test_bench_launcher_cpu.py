"""bench.py starts its own ranks: a plain ``python bench.py --gpus 2`` (no torchrun, WORLD_SIZE unset) must yield a
2-rank run -- rehearsed on CPU with the gloo backend through --launch-check (no GPU work; the rendezvous, the env the
children get and the relayed rank-0 line are what is under test).  Reference: the ranks come from ``accelerate launch``
(src/training/trainer.py:80-82)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_plain_invocation_with_two_gpus_starts_two_ranks():
    res = _run(["--gpus", "2", "--launch-check"])
    assert res.returncode == 0, res.stderr
    line = _json_line(res.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["all_reduce_of_ones"] == 2.0
    assert line["launcher"] == "self"


def test_single_gpu_invocation_does_not_spawn():
    res = _run(["--gpus", "1", "--launch-check"])
    assert res.returncode == 0, res.stderr
    line = _json_line(res.stdout)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1


def test_world_size_must_match_gpus_under_an_external_launcher():
    res = _run(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


def test_a_failing_rank_fails_the_launcher():
    # without a GPU every rank of the real benchmark exits with an error: the launcher must report it, not hang
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=300)
    assert res.returncode != 0
