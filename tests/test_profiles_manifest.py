"""profiles/rNN/MANIFEST.json (round 5 on): every committed profile file names the commit, box and command it came from, and the
recorded source hashes really are that commit's files -- so a number quoted from profiles/ can be tied to the kernels that produced
it, and bench.py's committed-profile fallback for `roofline.traffic` can refuse a stale summary (CPU only; needs the git history)."""
import glob
import hashlib
import json
import os
import subprocess

import pytest

from conftest import ROOT


def _manifests():
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "MANIFEST.json")))


def test_manifest_tool_records_sources(tmp_path):
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_manifest

    rels = make_manifest.tracked_sources()
    assert "haghighatshoarmuir2024_amd/csrc/beamform.hip" in rels and "bench.py" in rels
    make_manifest.record_sources(str(tmp_path))
    src = json.load(open(tmp_path / "SOURCES.json"))
    assert src["bench.py"] == hashlib.sha256(open(os.path.join(ROOT, "bench.py"), "rb").read()).hexdigest()
    assert "host" in json.load(open(tmp_path / "BOX.json"))


def test_bench_refuses_a_profile_without_a_matching_manifest_entry():
    import sys

    sys.path.insert(0, ROOT)
    import bench

    # round 4's summary has no manifest: its traffic figure may be read, but bench.py's fallback does not accept it
    assert bench.manifest_entry("profiles/r4/pmc_summary.csv") is None
    assert bench.source_sha256("haghighatshoarmuir2024_amd/csrc/beamform.hip") is not None
    assert bench.source_sha256("no/such/file") is None


@pytest.mark.parametrize("mpath", _manifests() or [None])
def test_every_profile_file_is_tied_to_a_commit(mpath):
    if mpath is None:
        pytest.skip("no profiles/r*/MANIFEST.json yet")
    man = json.load(open(mpath))
    d = os.path.dirname(mpath)
    listed = set(man["files"])
    present = {f for f in os.listdir(d) if os.path.isfile(os.path.join(d, f)) and f != "MANIFEST.json"}
    assert present <= listed, f"files without a manifest entry: {sorted(present - listed)}"
    have_git = os.path.isdir(os.path.join(ROOT, ".git"))
    for f, e in man["files"].items():
        assert e["command"] and e["box"].get("host"), f
        assert e["git_sha"], f"{f}: taken on uncommitted sources {e.get('uncommitted_sources')}"
        if have_git:
            rc = subprocess.run(["git", "-C", ROOT, "merge-base", "--is-ancestor", e["git_sha"], "HEAD"]).returncode
            assert rc == 0, f"{f}: {e['git_sha']} is not an ancestor of HEAD"
    if have_git:
        # spot-check one entry per distinct commit: the recorded hashes are that commit's files
        seen = set()
        for f, e in man["files"].items():
            if e["git_sha"] in seen:
                continue
            seen.add(e["git_sha"])
            for rel, h in e["sources_sha256"].items():
                blob = subprocess.run(["git", "-C", ROOT, "show", f"{e['git_sha']}:{rel}"], stdout=subprocess.PIPE).stdout
                assert hashlib.sha256(blob).hexdigest() == h, (f, rel)


HOT_SOURCES = tuple(f"haghighatshoarmuir2024_amd/csrc/{f}" for f in ("beamform.hip", "stht.hip", "rzcc.hip"))


def _round_of(mpath):
    name = os.path.basename(os.path.dirname(mpath))
    return int("".join(c for c in name if c.isdigit()) or 0)


def test_newest_profiles_belong_to_the_kernels_in_the_tree():
    """From round 6 on: the NEWEST round's committed profiles were taken on exactly the hot-path kernel sources of the
    working tree -- csrc/beamform.hip, stht.hip, rzcc.hip hash to what every entry of that round's MANIFEST recorded on the GPU box.
    Touching one of them after the profiles were taken fails this test until the profiles are re-taken (tools/profile_round.sh).
    Consequence checked too: bench.py's committed fallback for `roofline.traffic` accepts the newest pmc_summary.csv."""
    import sys

    ms = [m for m in _manifests() if _round_of(m) >= 6]
    if not ms:
        pytest.skip("no manifest of round 6 or later yet")
    mpath = max(ms, key=_round_of)
    man = json.load(open(mpath))
    now = {rel: hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest() for rel in HOT_SOURCES}
    stale = sorted({(f, rel) for f, e in man["files"].items() for rel in HOT_SOURCES if e["sources_sha256"].get(rel) != now[rel]})
    assert not stale, f"{os.path.relpath(mpath, ROOT)}: taken on other kernel sources than the tree's: {stale[:6]}"
    sys.path.insert(0, ROOT)
    import bench

    rel = os.path.relpath(os.path.join(os.path.dirname(mpath), "pmc_summary.csv"), ROOT)
    ent = bench.manifest_entry(rel)
    assert ent and ent["sources_sha256"]["haghighatshoarmuir2024_amd/csrc/beamform.hip"] == bench.source_sha256("haghighatshoarmuir2024_amd/csrc/beamform.hip")
    t, src = bench.traffic_from_profiles("beamform_ws_kernel", -(-4799 // 256) * 512 * 1100)
    assert src == rel and t and 1.0e8 < t < 2.0e8  # 2 x FETCH_SIZE + WRITE_SIZE of the headline's launch: about 136 MB
