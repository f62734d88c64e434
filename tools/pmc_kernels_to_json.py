"""Per-kernel summary of a set of rocprofv3 --pmc passes (one counter group per pass directory).

usage: pmc_kernels_to_json.py <gpurun_out dir> <pass directory prefix> [kernel-name substring]
Prints JSON: for every kernel (name cut at the first '(' or '<'-balanced template tail kept), the
number of launches seen and the PER-LAUNCH mean of every counter, plus hbm_bytes_per_launch =
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 where both were collected (gfx950: FETCH_SIZE reports half of the
bytes of a wide streaming read, MI355X_MICROARCH.md HBM section; WRITE_SIZE is taken as read), and
the sha256 of the library the passes ran on.
"""
import csv
import glob
import hashlib
import json
import os
import sys


def main():
    out, prefix = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else ""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get("GPDLA_LIB_PATH") or os.path.join(root, "gp_dla_detection_amd", "csrc", "libgpdla.so")
    with open(lib, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    acc = {}
    for path in glob.glob(os.path.join(out, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpdla::", "")
                if want and want not in name:
                    continue
                acc.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    kernels = {}
    for name, cs in sorted(acc.items()):
        per = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
        entry = {"launches": {c: len(v) for c, v in sorted(cs.items())}, "per_launch": per}
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            entry["hbm_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
        if per.get("SQ_INSTS_MFMA"):
            entry["mfma_busy_cycles_per_mfma"] = per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / per["SQ_INSTS_MFMA"]
        kernels[name] = entry
    print(json.dumps({"lib_sha256": sha, "passes": prefix, "kernels": kernels,
                      "note": "separate --pmc passes; FETCH_SIZE doubled for gfx950, WRITE_SIZE as read (KB per "
                              "launch); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, "
                              "SQ_VALU_MFMA_BUSY_CYCLES in cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs"}, indent=1))


if __name__ == "__main__":
    main()
