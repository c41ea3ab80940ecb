"""CPU tier: the parts of bench.py that need no GPU — which configurations the line measures (BASELINE.json's own sample counts), and how the
rocprofv3 counters become the line's traffic / vector-pipe figures (profiles/calibration.json: VERDICT r04 item 3b / 3c)."""
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_line_measures_baselines_own_configurations(bench):
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))["configs"]
    w = bench.WORKLOADS
    # configs[1] the headline; configs[2..4] beside it at the sample counts BASELINE names (configs[3]: a stated share of its 2048 passes)
    assert (w["cfg2"]["width"], w["cfg2"]["height"], w["cfg2"]["samples"], w["cfg2"]["bounces"]) == (1280, 720, 256, 8) and "1280x720, 256 spp, 8 bounces" in base[1]
    assert (w["cfg3"]["samples"], w["cfg3"]["bounces"]) == (1024, 32) and "1024 spp, 32 bounces" in base[2]
    assert w["cfg4"]["samples"] == 2048 and "2048 spp" in base[3]
    assert w["soup10m"]["samples"] == 512 and w["soup10m"]["triangles"] == 10_000_000 and "512 spp" in base[4]
    measured = {k: spp for k, spp, _, _ in bench.OTHER_WORKLOADS}
    assert all(measured[k] == w[k]["samples"] for k in ("cfg3", "cfg4", "soup", "soup10m")), "every other workload is timed at BASELINE's own sample count (round 6: configs[3] too)"
    counted = {k: c for k, _, c, _ in bench.OTHER_WORKLOADS}
    assert counted["cfg4"] == 256 and all(counted[k] == measured[k] for k in ("cfg3", "soup", "soup10m"))          # (configs[3]'s counters: a 256-pass dispatch, also timed: at_256_spp)
    assert set(bench.WIDE4_WORKLOADS) == {"cfg3", "cfg4", "soup"}
    # the multi-GPU objects: long enough that a 1 / 8 share is not a drain test (>= 128 passes), the soup among them
    assert set(bench.SCALING_SPP) == {"cfg4", "soup10m"} and all(v >= 128 for v in bench.SCALING_SPP.values())


def test_traffic_uses_the_calibrated_factors_not_the_blanket_doubling(bench):
    cal = bench.calibration()
    if not cal:
        pytest.skip("profiles/calibration.json absent")
    # what tools/ubench_calib.hip measured on one MI355X: FETCH_SIZE counts a request at 64 bytes whatever its size
    assert 0.9 < cal["k_gather64"]["read_factor"] < 1.05          # divergent 64-byte gathers: face value
    assert 1.8 < cal["k_gather128"]["read_factor"] < 2.05         # 128-byte requests: half
    assert 1.95 < cal["k_stream"]["read_factor"] < 2.05           # the guide's case
    assert 0.98 < cal["k_write128"]["write_factor"] < 1.02
    rec = {"fetch_bytes_raw": 100e9, "write_bytes": 10e9}
    lo, hi = bench.calibrated_traffic(rec)
    assert lo == pytest.approx(100e9 * cal["k_gather64"]["read_factor"] + 10e9 * cal["k_write128"]["write_factor"])
    assert hi == pytest.approx(100e9 * cal["k_gather128"]["read_factor"] + 10e9 * cal["k_write128"]["write_factor"]) and hi > 1.8 * lo - 10e9
    assert bench.calibrated_traffic({}) == (None, None)
    f = bench.fractions(500e9, lo, 67.5, hi)
    assert f["frac_algorithmic"] == pytest.approx(500e9 / 67.5 / 1e6 / 8000.0, abs=1e-4)
    assert f["frac_measured_traffic"] < f["frac_measured_traffic_upper"] < 1.0
    # the vector pipe's formula reads well above 1 at known saturation: a reading of 0.9 is about half of the pipe
    assert all(1.4 < cal[k]["pipe_busy_at_saturation"] < 2.6 for k in ("k_fma", "k_add", "k_mix"))
    assert cal["k_fma"]["counters"]["SQ_ACTIVE_INST_VALU"] == pytest.approx(cal["k_fma"]["counters"]["SQ_INSTS_VALU"], rel=1e-3)


def test_committed_traffic_records_carry_what_the_line_needs(bench):
    """profiles/hbm_traffic*.json as tools/parse_prof.py / tools/traffic_table.py write them since round 5: raw counter bytes + the calibrated figure + its upper bound."""
    import glob
    seen = 0
    for path in glob.glob(os.path.join(REPO, "profiles", "hbm_traffic*.json")):
        t = json.load(open(path))
        if t.get("tag", "").startswith(("r01", "r02", "r03", "r04")):
            continue
        seen += 1
        lo, hi = bench.calibrated_traffic(t)
        assert t["hbm_bytes_per_launch"] == pytest.approx(lo) and t["hbm_bytes_upper"] == pytest.approx(hi), path
        assert t["source_md5"] and t["write_bytes"] > 0 and t["fetch_bytes_raw"] > 0
    if not seen:
        pytest.skip("no round-5 traffic record yet")
