"""bench.py only reports hardware counters (roofline.traffic, valu_*) collected on the build it runs:
profiles/counters.json entries carry the vcy_version() of the library they were measured on -- a string that
ends in a hash of the library's sources -- and a single differing byte makes bench.py report null."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_library_build_names_a_source_hash():
    b = bench.library_build()
    assert b.startswith("vacancy_amd ") and " src:" in b
    h = b.split(" src:")[1]
    assert len(h) == 16 and all(c in "0123456789abcdef" for c in h)


def test_stale_counters_are_refused(tmp_path):
    build = bench.library_build()
    p = tmp_path / "counters.json"
    entry = {"hbm_bytes_per_launch": 123, "SQ_INSTS_VALU": 4.0, "build": build}
    json.dump({"default_1024_32_b1_c1": entry}, open(p, "w"))
    ctr, note = bench.load_counters("default_1024_32_b1_c1", build, str(p))
    assert note is None and ctr["hbm_bytes_per_launch"] == 123
    # one byte of the stamp flipped: the same entry is now another build's
    flipped = build[:-1] + ("0" if build[-1] != "0" else "1")
    ctr, note = bench.load_counters("default_1024_32_b1_c1", flipped, str(p))
    assert ctr is None and "another build" in note
    # unstamped entries (written before the stamp existed) and missing keys are refused as well
    json.dump({"k": {"hbm_bytes_per_launch": 1}}, open(p, "w"))
    assert bench.load_counters("k", build, str(p))[0] is None
    assert bench.load_counters("absent", build, str(p))[0] is None
    assert bench.load_counters("k", build, str(tmp_path / "nope.json"))[0] is None


def test_committed_counters_are_stamped():
    """Whatever profiles/counters.json holds is either stamped or will be ignored; nothing in between."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(path):
        return
    for key, entry in json.load(open(path)).items():
        assert isinstance(entry, dict), key
        if "build" in entry:
            assert " src:" in entry["build"], key
