"""The committed fixtures are what the committed generator produces (authoring container only: needs /root/reference).

`tests/golden/gen/gen_goldens.py --only <section>` draws every fixture's random inputs from a stream seeded by the fixture's file name,
so a section re-run must reproduce the committed arrays bit for bit.  The cheap sections (function vectors: ~2 s) run here on every
CPU test run; `APT_FRESHNESS_ALL=1` adds the whole-kernel `textured` section (~30 s).  Arrays are compared, not file bytes: the zip
container stores timestamps.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
GEN = os.path.join(GOLD, "gen", "gen_goldens.py")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/bxdf"), reason="the reference tree is only present in the authoring container")


def _regen(section, tmp_path):
    res = subprocess.run([sys.executable, GEN, "--only", section, "--out", str(tmp_path)], capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert res.returncode == 0, res.stderr[-2000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert made, "the section wrote nothing"
    return made


def _same(a_path, b_path):
    a, b = np.load(a_path, allow_pickle=False), np.load(b_path, allow_pickle=False)
    assert sorted(a.files) == sorted(b.files), (a_path, set(a.files) ^ set(b.files))
    for k in a.files:
        x, y = a[k], b[k]
        assert x.dtype == y.dtype and x.shape == y.shape, (a_path, k)
        assert x.tobytes() == y.tobytes(), f"{os.path.basename(a_path)}[{k}] differs from what the generator writes now"


SECTIONS = ["func"] + (["textured"] if os.environ.get("APT_FRESHNESS_ALL") == "1" else [])


@pytest.mark.parametrize("section", SECTIONS)
def test_section_reproduces_the_committed_fixtures(section, tmp_path):
    for f in _regen(section, tmp_path):
        _same(os.path.join(tmp_path, f), os.path.join(GOLD, f))
