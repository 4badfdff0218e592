#!/usr/bin/env python3
"""profiles/rNN/MANIFEST.json: which commit, box and command every committed profile file came from.

On the GPU box (no .git there) tools/profile_round.sh records, next to the raw output, `BOX.json` (host, device UUID, ROCm),
`SOURCES.json` (SHA-256 of every kernel source, header and bench.py AS RUN) and `COMMANDS.tsv` (output file <TAB> command).
Here, in the build container:

    python tools/make_manifest.py gpurun_out/r5 profiles/r5 [file ...]

copies the named files (default: every regular file of the run directory except the three records) into profiles/r5/ and writes /
updates MANIFEST.json: per file {git_sha, box, command, sources_sha256}.  `git_sha` is HEAD -- but only if every recorded source
hash equals the file at HEAD (otherwise the run was made on an uncommitted tree and the entry says so); bench.py's
`roofline.traffic` fallback and tests/test_profiles_manifest.py check these entries.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = ("BOX.json", "SOURCES.json", "COMMANDS.tsv")


def tracked_sources():
    out = []
    d = os.path.join(ROOT, "haghighatshoarmuir2024_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            out.append(os.path.join("haghighatshoarmuir2024_amd", "csrc", f))
    return out + ["include/micloc_hip.h", "bench.py"]


def sha256_of(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def record_sources(out_dir):
    """(GPU box) SOURCES.json + BOX.json of the tree that is running."""
    src = {rel: sha256_of(os.path.join(ROOT, rel)) for rel in tracked_sources()}
    json.dump(src, open(os.path.join(out_dir, "SOURCES.json"), "w"), indent=1, sort_keys=True)
    box = {"host": os.uname().nodename}
    try:
        import torch

        pr = torch.cuda.get_device_properties(0)
        box.update(device=pr.name, gcn_arch=getattr(pr, "gcnArchName", ""), compute_units=pr.multi_processor_count, uuid=str(getattr(pr, "uuid", "")),
                   hip=str(torch.version.hip))
    except Exception as e:  # noqa: BLE001
        box["device_error"] = str(e)[:200]
    json.dump(box, open(os.path.join(out_dir, "BOX.json"), "w"), indent=1, sort_keys=True)


def git(*a):
    return subprocess.run(["git", "-C", ROOT] + list(a), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip()


def main(argv):
    if argv and argv[0] == "--record":
        os.makedirs(argv[1], exist_ok=True)
        record_sources(argv[1])
        return 0
    run_dir, prof_dir = argv[0], argv[1]
    files = argv[2:] or [f for f in sorted(os.listdir(run_dir)) if os.path.isfile(os.path.join(run_dir, f)) and f not in RECORDS]
    src = json.load(open(os.path.join(run_dir, "SOURCES.json")))
    box = json.load(open(os.path.join(run_dir, "BOX.json")))
    cmds = {}
    for line in open(os.path.join(run_dir, "COMMANDS.tsv")):
        if "\t" in line:
            k, v = line.rstrip("\n").split("\t", 1)
            cmds[k] = v
    head = git("rev-parse", "HEAD")
    dirty = []
    for rel, h in src.items():
        blob = subprocess.run(["git", "-C", ROOT, "show", f"{head}:{rel}"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        if hashlib.sha256(blob).hexdigest() != h:
            dirty.append(rel)
    os.makedirs(prof_dir, exist_ok=True)
    mpath = os.path.join(prof_dir, "MANIFEST.json")
    man = json.load(open(mpath)) if os.path.exists(mpath) else {"files": {}}
    man["note"] = ("file -> the commit whose sources produced it (every hash of sources_sha256 equals `git show <git_sha>:<path>`), the GPU box "
                   "and the command; written by tools/make_manifest.py from the records tools/profile_round.sh leaves on the box")
    for f in files:
        shutil.copy2(os.path.join(run_dir, f), os.path.join(prof_dir, f))
        man["files"][f] = {"git_sha": head if not dirty else None, "uncommitted_sources": dirty or None, "box": box,
                           "command": cmds.get(f, cmds.get("*", "see tools/profile_round.sh")), "sources_sha256": src}
    json.dump(man, open(mpath, "w"), indent=1, sort_keys=True)
    print(f"{mpath}: {len(files)} file(s) at {head[:10]}" + (f"  [UNCOMMITTED: {dirty}]" if dirty else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
