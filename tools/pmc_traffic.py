"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output to bytes per launch of mg_step_kernel.

Usage (on the GPU box, two separate passes as MI355X_MICROARCH.md prescribes):
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  python3 tools/pmc_traffic.py out/fetch out/write > profiles/r01_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are in KB.  gfx950 correction (guide §HBM): FETCH_SIZE reads exactly half the bytes
of a wide coalesced (16 B/lane) streaming read -> doubled; WRITE_SIZE is exact for 16-B/lane stores, and is
left uncorrected for the kernel's 4-B/lane observation stores (uncalibrated width, stated in the output).
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_counter(d, counter, kernel="mg_step_kernel<false"):
    vals = []
    names.clear()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"])
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


names = set()
from bench import kernel_source_hash          # noqa: E402  (ties the numbers to the kernel source they were taken on)
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "mg_step_kernel<false,16,false,true,true>"
fetch_kb, nf = mean_counter(sys.argv[1], "FETCH_SIZE")
write_kb, nw = mean_counter(sys.argv[2], "WRITE_SIZE")
out = {"kernel": KERNEL, "kernel_names_seen": sorted(names), "source_sha256": kernel_source_hash(), "launches_sampled": [nf, nw], "FETCH_SIZE_KB_raw": fetch_kb,
       "WRITE_SIZE_KB_raw": write_kb,
       "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 if fetch_kb is not None and write_kb is not None else None,
       "correction": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE as reported (4-B/lane stores: width uncalibrated)",
       "note": "counted at the L2 fabric boundary: Infinity-Cache hits are included (MI355X_MICROARCH.md §HBM)"}
print(json.dumps(out))
