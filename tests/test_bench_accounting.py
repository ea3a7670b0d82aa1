"""bench.py's byte model: a kernel that does another stage's work is credited with that stage's unit bytes, so that
the pipeline's algorithmic bytes per sample do not depend on how many kernels the work is spread over."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _stats(launch_extend, launch_shadow):
    return {"n_samples": 1000, "n_extend": 4480, "n_shadow_traced": 3110, "n_lit": 2600,
            "launches": {"generate": 1, "extend": launch_extend, "shade": 8, "shadow": launch_shadow, "finalize": 1}}


def test_the_pipeline_total_does_not_depend_on_which_kernel_does_the_work():
    staged = bench.kernel_bytes(_stats(8, 8))
    light_in_place = bench.kernel_bytes(_stats(8, 0))
    traced_in_place = bench.kernel_bytes(_stats(0, 0))
    total = sum(staged.values())
    assert sum(light_in_place.values()) == total and sum(traced_in_place.values()) == total
    # 140 N_samples + 184 N_shade + 88 N_shadow + 24 N_lit, with the model's own counts (one queue entry per traced ray)
    assert staged["extend"] == 40 * 4480 and staged["shadow"] == 44 * 3110 + 24 * 2600
    assert light_in_place["shadow"] == 0 and light_in_place["shade"] == staged["shade"] + staged["shadow"]
    assert traced_in_place["extend"] == 0 and traced_in_place["shadow"] == 0
    assert traced_in_place["generate"] == staged["generate"] + 40 * 1000                       # the camera rays are traced by k_generate
    assert traced_in_place["shade"] == staged["shade"] + staged["shadow"] + 40 * (4480 - 1000)   # the continuation rays by the shade kernel


def test_a_stage_that_is_not_launched_has_no_row():
    st = _stats(0, 0)
    st["kernel_ms"] = {"generate": 1.0, "extend": 0.0, "shade": 10.0, "shadow": 0.0, "finalize": 0.5}
    per = bench.region_roofline(st, None, 1024, 2400.0)
    assert set(per) == {"generate", "shade", "finalize"} and per["shade"]["launches"] == 8
    assert bench.pick_dominant(per) == "shade"


def test_plain_bench_with_gpus_n_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 4` with no launcher around it (no WORLD_SIZE / RANK in the environment) must not die on the rendezvous:
    it re-executes its own command line under torch.distributed.run, one process per GPU, 127.0.0.1 rendezvous."""
    import subprocess
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--config", "c2"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "2", "--config", "c2"]          # the ranks run this very command line
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_region_roofline_carries_traffic_and_valu_next_to_the_credited_bytes():
    """What the headline needs to say which resource binds: HBM-side bytes per launch from the PMC record and issued wave-instructions."""
    st = _stats(0, 0)
    st["kernel_ms"] = {"generate": 1.0, "extend": 0.0, "shade": 10.0, "shadow": 0.0, "finalize": 0.5}
    counters = {"kernels": {"shade": {"bytes_per_unit": 120.0, "valu_insts_per_unit": 22.0}}}
    per = bench.region_roofline(st, counters, 1024, 2400.0)
    sh = per["shade"]
    assert sh["traffic"] == int(120.0 * 4480 / 8) and sh["traffic_over_algorithmic"] < 1.0            # the fused kernel moves less than it is credited with
    assert abs(sh["valu"]["busy_frac"] - 22.0 * 4480 / 10e-3 / (1024 * 2400e6 / 4)) < 1e-4
