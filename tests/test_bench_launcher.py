"""bench.py --gpus N without a launcher around it must itself start N ranks: the parent never
imports torch or touches HIP, runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`, forwards rank 0's
one JSON line, fails if a rank fails, and refuses when fewer than N GPUs are visible."""
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
FAKE = os.path.join(ROOT, "tests", "fake_rank.py")


def _launch(n, have, env_extra=None, capsys=None):
    import bench
    old = dict(os.environ)
    os.environ["MVR_BENCH_LAUNCH_CMD"] = FAKE
    os.environ.update(env_extra or {})
    try:
        return bench.launch_ranks(n, ["--gpus", str(n), "--steps", "3"], count_gpus=lambda: have)
    finally:
        os.environ.clear(); os.environ.update(old)


def test_launcher_starts_n_ranks(capfd):
    rc = _launch(2, 2)
    out = capfd.readouterr().out
    assert rc == 0
    line = json.loads(out.strip().splitlines()[-1])
    assert line["world_env"] == 2 and line["ranks_counted"] == 2 and line["master"] == "127.0.0.1"
    assert line["argv"] == ["--gpus", "2", "--steps", "3"]


def test_launcher_three_ranks(capfd):
    assert _launch(3, 8) == 0
    assert json.loads(capfd.readouterr().out.strip().splitlines()[-1])["ranks_counted"] == 3


def test_launcher_refuses_without_enough_gpus(capfd):
    assert _launch(8, 1) == 2
    cap = capfd.readouterr()
    assert cap.out.strip() == "" and "refusing" in cap.err


def test_launcher_propagates_rank_failure(capfd):
    rc = _launch(2, 2, {"FAKE_RANK_FAIL": "1"})
    assert rc != 0 and capfd.readouterr().out.strip() == ""


def test_parent_process_never_loads_torch_or_hip():
    """python bench.py --gpus 2 on a box without GPUs: refused (exit 2) by a parent that has neither torch nor a HIP
    runtime mapped (the GPU count comes from a child process)."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    libs = [l for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libtorch' in l]\n"
            "    print('EXIT', e.code, 'TORCH' if 'torch' in sys.modules else 'clean', len(libs))\n" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
    assert "EXIT 2 clean 0" in r.stdout, (r.stdout, r.stderr)
    assert "refusing" in r.stderr


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, env=dict(os.environ, WORLD_SIZE="2", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
