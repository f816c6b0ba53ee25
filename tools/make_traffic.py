"""profiles/traffic.json entry for the dominant kernel from a round profile (tools/profile_round.sh):
HBM-side bytes per launch = 32 B x (TCC_EA0_RDREQ_DRAM_32B + TCC_EA0_WRREQ_WRITE_DRAM_32B), the size-aware
request counters calibrated by tools/calib_fetch.hip (exact on a known streaming read; 128 B per miss on
the traversal's scattered 64-byte reads).  Also records FETCH_SIZE / WRITE_SIZE and the request split.
    python tools/make_traffic.py gpurun_out/<tag>_pmc.txt "<kernel substring>" "<traffic key>" [calib file] """
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yart_amd import api

pmc, kernel, key = sys.argv[1], sys.argv[2], sys.argv[3]
calib = sys.argv[4] if len(sys.argv) > 4 else None
vals, cur = {}, None
for line in open(pmc):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    if cur and kernel in cur:
        m = re.match(r"\s+(\S+)\s+dispatches=(\d+)\s+avg=(\S+)", line)
        if m:
            vals.setdefault(m.group(1), float(m.group(3)))
need = ["TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"]
missing = [n for n in need if n not in vals]
if missing:
    raise SystemExit("missing counters %s for kernel %r in %s" % (missing, kernel, pmc))
rd = 32.0 * vals["TCC_EA0_RDREQ_DRAM_32B_sum"]
wr = 32.0 * vals["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"]
entry = {
    "kernel_fingerprint": api.kernel_fingerprint(),     # bench.py ignores the entry once the kernel sources change
    "hbm_bytes_per_launch": int(rd + wr),
    "read_bytes": int(rd), "write_bytes": int(wr),
    "source": "%s, %s: 32 B x (TCC_EA0_RDREQ_DRAM_32B + TCC_EA0_WRREQ_WRITE_DRAM_32B), separate --pmc passes; "
              "calibration: %s" % (os.path.basename(pmc).replace("gpurun_out/", "profiles/"), kernel,
                                   os.path.basename(calib) if calib else "profiles/r02_calib_fetch.txt"),
    "counters": {k: v for k, v in sorted(vals.items()) if k.startswith(("TCC_", "FETCH", "WRITE", "duration"))},
}
if "TCC_EA0_RDREQ_128B_sum" in vals and "TCC_EA0_RDREQ_sum" in vals:
    entry["read_requests_128B_fraction"] = round(vals["TCC_EA0_RDREQ_128B_sum"] / max(1.0, vals["TCC_EA0_RDREQ_sum"]), 4)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
tr = json.load(open(path)) if os.path.exists(path) else {}
tr[key] = entry
json.dump(tr, open(path, "w"), indent=1)
print(key, json.dumps(entry)[:400])
