"""bench.py launch contract (no GPU needed): `--gpus N` must never silently run fewer ranks than asked for."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    env["HIP_VISIBLE_DEVICES"] = ""  # this test is about the launcher: pretend there is no device even on a GPU box
    env["CUDA_VISIBLE_DEVICES"] = ""
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_refuses_too_few_devices():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "--gpus 2 but only 0 device(s) visible" in r.stderr and not r.stdout.strip()


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--steps", "1"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in r.stderr


def test_no_gpu_is_an_error_not_a_fallback():
    r = _run(["--gpus", "1", "--steps", "1"])
    assert r.returncode != 0 and "no GPU visible" in r.stderr and not r.stdout.strip()
