"""Are the committed golden fixtures reproducible from HEAD?  (build container only: needs /root/reference)

Regenerates every section of oracle/gen/gen_golden.py into a scratch directory (one child process per section, a few at a
time) and compares each regenerated file with tests/golden/<file> by CONTENT (parsed JSON: key order does not count).
Exit 0 = every fixture is what the generator produces today; a differing or missing file is listed and the exit code is 1.
Skips (exit 0, says so) where the reference tree is absent, e.g. on the GPU box.

    python tools/check_golden.py [--jobs 6] [section ...]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "oracle", "gen", "gen_golden.py")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def sections():
    src = open(GEN).read()
    start = src.index("SECTIONS = OrderedDict(")
    end = src.index("\n\n", start)
    import re
    return re.findall(r"(\w+)=section_", src[start:end])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=max(1, min(6, (os.cpu_count() or 2) - 1)))
    ap.add_argument("sections", nargs="*")
    args = ap.parse_args()
    ref = os.environ.get("MD_REFERENCE_ROOT", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "metadrive")):
        print("check_golden: no reference tree at %s -- skipped" % ref)
        return 0
    todo = args.sections or sections()
    out_dir = tempfile.mkdtemp(prefix="md_golden_")
    env = dict(os.environ, MD_GOLDEN_OUT=out_dir, PYTHONDONTWRITEBYTECODE="1", MD_BUILD_WORKERS="1")
    running, failed, t0 = {}, [], time.time()
    queue = list(todo)
    while queue or running:
        while queue and len(running) < args.jobs:
            name = queue.pop(0)
            log = open(os.path.join(out_dir, name + ".log"), "w")
            running[name] = (subprocess.Popen([sys.executable, GEN, name], env=env, stdout=log, stderr=subprocess.STDOUT), log, time.time())
        time.sleep(1.0)
        for name, (p, log, ts) in list(running.items()):
            if p.poll() is not None:
                log.close()
                del running[name]
                print("  section %-18s %s in %.0f s" % (name, "ok" if p.returncode == 0 else "FAILED (rc %d)" % p.returncode, time.time() - ts), flush=True)
                if p.returncode != 0:
                    failed.append(name)
    bad = []
    made = sorted(f for f in os.listdir(out_dir) if f.endswith(".json"))
    for f in made:
        committed = os.path.join(GOLDEN, f)
        if not os.path.exists(committed):
            bad.append((f, "not committed"))
            continue
        with open(os.path.join(out_dir, f)) as a, open(committed) as b:
            if json.load(a) != json.load(b):
                bad.append((f, "content differs"))
    print("check_golden: %d sections, %d files regenerated into %s in %.0f s" % (len(todo), len(made), out_dir, time.time() - t0))
    for name in failed:
        print("  generator section failed: %s (see %s)" % (name, os.path.join(out_dir, name + ".log")))
    for f, why in bad:
        print("  %s: %s" % (f, why))
    if not failed and not bad:
        print("check_golden: every regenerated fixture equals the committed one")
    return 1 if (failed or bad) else 0


if __name__ == "__main__":
    sys.exit(main())
