"""Writes the PMC traffic figure of one kernel into profiles/traffic.json, stamped with the kernel sources it was
measured on (bench.py quotes `roofline.traffic` only while that stamp matches the sources in the tree).

    python tools/update_traffic.py CONFIG_GPUS KERNEL_SUBSTRING FETCH_CSV WRITE_CSV PROFILE_TAG [KEY TIMES]
e.g. python tools/update_traffic.py c4_gpus1 "sdia_jacobi2c_finest<12, 2>" profiles/r02_c4_pmc_fetch.csv profiles/r02_c4_pmc_write.csv r02

KEY / TIMES: the name bench.py gives the roofline's unit of work and how many launches of the kernel it consists of, e.g.
    python tools/update_traffic.py c5_gpus1 "lat_march<3>" fetch.csv write.csv r02b "lat_march<MODE_GS> x 9 colours" 9
(a Gauss-Seidel sweep = nine colour launches: the figure is nine times the mean launch).

The CSVs are summarize.py's per-kernel tables (kernel, grid_size_threads, counter, launches, mean, max; KB).
FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 bytes: MI355X_MICROARCH.md, HBM section; checked in
profiles/README.md against a known byte count), WRITE_SIZE is exact.
"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha

cfg, kern, fetch_csv, write_csv, tag = sys.argv[1:6]
key = sys.argv[6] if len(sys.argv) > 6 else kern
times = float(sys.argv[7]) if len(sys.argv) > 7 else 1.0


def pick(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if kern in r["kernel"] and r["counter"] == counter]
    if not rows:
        raise SystemExit(f"{kern!r} / {counter} not found in {path}")
    r = max(rows, key=lambda r: int(r["grid_size_threads"]))
    return float(r["mean"]) * 1024.0, int(r["grid_size_threads"]), int(r["launches"])


rd, grid, n1 = pick(fetch_csv, "FETCH_SIZE")
wr, _, n2 = pick(write_csv, "WRITE_SIZE")
path = os.path.join(ROOT, "profiles", "traffic.json")
doc = json.load(open(path)) if os.path.exists(path) else {}
doc.setdefault("entries", {})[f"{cfg}:{key}"] = {
    "hbm_bytes_per_launch": times * (2.0 * rd + wr), "read_bytes": times * 2.0 * rd, "write_bytes": times * wr,
    "kernel": kern, "launches_per_unit": times, "grid_size_threads": grid, "launches_averaged": [n1, n2],
    "kernel_source_sha": kernel_source_sha(), "profile": f"profiles/{os.path.basename(fetch_csv)} + {os.path.basename(write_csv)} ({tag})",
    "note": "L2 <-> fabric boundary (requests served by the Infinity Cache included); FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024",
}
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(doc["entries"][f"{cfg}:{key}"], indent=1))
