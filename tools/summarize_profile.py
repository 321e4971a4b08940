"""Condense the rocprofv3 outputs of tools/profile_round.sh into a small text summary (what gets
committed under profiles/)."""
import csv
import glob
import json
import os
import re
import sys


def kernel_stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "trace", "*", "*kernel_stats.csv")):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    return rows


def pmc_avg(d, sub, counter):
    """Average counter value per dispatch of each kernel."""
    acc = {}
    for f in glob.glob(os.path.join(d, sub, "*", "*counter_collection.csv")):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = r["Kernel_Name"]
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in acc.items()}


def main():
    d = sys.argv[1]
    print("== bench line")
    try:
        print(open(os.path.join(d, "bench.json")).read().strip())
    except OSError:
        pass
    print("== rocprofv3 --kernel-trace --stats (top kernels)")
    for r in kernel_stats(d)[:6]:
        print("%-70s calls %6s  avg %10.1f us  min %9.1f  max %9.1f  %6.2f%%" %
              (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
               float(r["Percentage"])))
    fetch = pmc_avg(d, "pmc_fetch", "FETCH_SIZE")
    write = pmc_avg(d, "pmc_write", "WRITE_SIZE")
    print("== HBM traffic per dispatch from PMC (FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md: on gfx950")
    print("   FETCH_SIZE under-reports wide coalesced reads by 2x -> corrected = 2 x FETCH_SIZE; WRITE_SIZE exact)")
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        if "env_kernel" not in k and "step_kernel" not in k:
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        m = re.search(r"env_kernel<(\d+)", k)
        if m:
            traffic["env_kernel<%s>" % m.group(1)] = int((2 * f + w) * 1024)
            if m.group(1) == "511":
                traffic["step_kernel"] = traffic["env_kernel<511>"]
        elif "step_kernel" in k:
            traffic["step_kernel"] = int((2 * f + w) * 1024)
        print("%-60s FETCH_SIZE %10.1f KiB (corrected %10.1f KiB)  WRITE_SIZE %10.1f KiB  => %.2f MB/launch" %
              (k[:60], f, 2 * f, w, (2 * f + w) * 1024 / 1e6))
    insts = {c: pmc_avg(d, "pmc_insts", c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVES")}
    stats = {r["Name"]: float(r["AverageNs"]) for r in kernel_stats(d)}
    if insts["SQ_INSTS_VALU"]:
        print("== instruction mix per dispatch (wave-instructions; VALU issue = 4 cycles per wave64 instruction on one of the")
        print("   1024 SIMDs, shader clock taken as 2.4 GHz)")
        for k in sorted(insts["SQ_INSTS_VALU"]):
            if "env_kernel" not in k and "step_kernel" not in k:
                continue
            v = {c: insts[c].get(k, 0.0) for c in insts}
            line = "%-60s waves %7.0f  VALU %10.0f (%5.0f/wave)  SALU %10.0f  LDS %9.0f  SMEM %9.0f" % (
                k[:60], v["SQ_WAVES"], v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), v["SQ_INSTS_SALU"],
                v["SQ_INSTS_LDS"], v["SQ_INSTS_SMEM"])
            if k in stats:
                line += "  VALU issue %.0f %% of the %.1f us" % (100.0 * v["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e3 / (stats[k] / 1e3) , stats[k] / 1e3)
            print(line)
    waits = {c: pmc_avg(d, "pmc_wait", c) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                     "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")}
    if waits["SQ_WAVE_CYCLES"]:
        print("== where the wave-cycles go (SQ counters in quad-cycles, per dispatch)")
        for k in sorted(waits["SQ_WAVE_CYCLES"]):
            if "env_kernel" not in k and "step_kernel" not in k:
                continue
            wc = waits["SQ_WAVE_CYCLES"].get(k, 0.0)
            if wc <= 0:
                continue
            print("%-60s wave-cycles %12.0f  waiting (s_waitcnt / barrier) %4.1f %%  issue stalls %4.1f %%  issuing %4.1f %% (VALU %4.1f %%)" % (
                k[:60], wc, 100 * waits["SQ_WAIT_ANY"].get(k, 0) / wc, 100 * waits["SQ_WAIT_INST_ANY"].get(k, 0) / wc,
                100 * waits["SQ_ACTIVE_INST_ANY"].get(k, 0) / wc, 100 * waits["SQ_ACTIVE_INST_VALU"].get(k, 0) / wc))
    if len(sys.argv) > 2:
        # machine-readable copy for bench.py's roofline.traffic (bytes per launch, corrected as above)
        with open(sys.argv[2], "w") as fh:
            json.dump(dict(source=os.path.basename(os.path.normpath(d)), workload="metadrive",
                           bytes_per_launch=traffic), fh, indent=1)


if __name__ == "__main__":
    main()
