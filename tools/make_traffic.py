#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py's roofline block quotes) from the committed summaries of one profiling round:
    python tools/make_traffic.py r4c      (reads profiles/r4c_summary.md, profiles/r4c_pergenome_summary.md)"""
import json, re, sys
tag = sys.argv[1]
def counters(path):
    out = {}
    for m in re.finditer(r"^([A-Z][A-Za-z0-9_]+)\s+([0-9.e+]+)\s+\d+\s*$", open(path).read(), re.M):
        out[m.group(1)] = float(m.group(2))
    return out
def launch_ms(path):
    m = re.search(r"void pg::k_probe<[^\n]*?\s(\d+)\s+(\d+)\s+([0-9.e+]+)\s", open(path).read())
    return float(m.group(3)) / 1e6
c, g = counters(f"profiles/{tag}_summary.md"), counters(f"profiles/{tag}_pergenome_summary.md")
pos = 799999200
hbm = lambda d: 2.0 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
t = launch_ms(f"profiles/{tag}_summary.md")
out = {
 "round": tag, "kernel": "k_probe",
 "kernel_sources_sha16": __import__("hashlib").sha256(b"".join(open("panagram_amd/csrc/" + f, "rb").read() for f in ("pg_anchor.hip", "pg_device.h", "pg_kernels.h", "pg_kernels.hip", "pg_api.hip"))).hexdigest()[:16],
 "kernel_sources_note": "sha256 of pg_anchor.hip + pg_device.h + pg_kernels.h + pg_kernels.hip + pg_api.hip (bench.KERNEL_SOURCES) when this file was made (right after the --pmc passes): bench.py reports whether the kernels it runs are still those",
 "workload": "bench.py default (8 x 100 Mb, k=21; the table as the library builds it, no re-hash), ONE co-scheduled launch over all 8 anchor genomes",
 "positions_per_launch": pos,
 "FETCH_SIZE_kb": c["FETCH_SIZE"], "WRITE_SIZE_kb": c["WRITE_SIZE"], "TCC_EA0_RDREQ": c["TCC_EA0_RDREQ_sum"], "TCC_HIT": c["TCC_HIT_sum"], "TCC_MISS": c["TCC_MISS_sum"],
 "fetch_correction": 2.0,
 "correction_note": "k_probe fetches whole 128-B table lines (8 lanes x 16 B, one request per line). On gfx950 a 128-B request is tallied as 64 B by FETCH_SIZE (MI355X_MICROARCH.md, HBM section); calibrated with tools/gather_bench: its 128-B-per-probe kernel reads 63.7 B/probe in FETCH_SIZE, 1 RDREQ/probe (profiles/r1_fetch_size_calibration.csv); cross-check here: %.1f M RDREQ x 128 B = %.2f GB ~ 2 x FETCH_SIZE (%.2f GB). WRITE_SIZE taken as is." % (c["TCC_EA0_RDREQ_sum"] / 1e6, c["TCC_EA0_RDREQ_sum"] * 128 / 1e9, 2 * c["FETCH_SIZE"] * 1024 / 1e9),
 "hbm_bytes_per_launch": hbm(c),
 "per_genome_launches": {"positions_per_launch": pos // 8, "FETCH_SIZE_kb": g["FETCH_SIZE"], "WRITE_SIZE_kb": g["WRITE_SIZE"],
                         "TCC_EA0_RDREQ": g.get("TCC_EA0_RDREQ_sum"), "hbm_bytes_per_launch": hbm(g), "profile": f"profiles/{tag}_pergenome_summary.md",
                         "SQ_INSTS_VALU": g["SQ_INSTS_VALU"]},
 "profile": f"profiles/{tag}_summary.md",
 "counter_note": "FETCH_SIZE / WRITE_SIZE sit on the L2's fabric side (MI355X_MICROARCH.md): Infinity Cache hits are included, so these are upper bounds for the bytes that reach HBM.",
 "SQ_INSTS_VALU": c["SQ_INSTS_VALU"],
 "SQ_INSTS_VALU_note": "VALU wave-instructions per launch of k_probe (pmc_sq pass): %.2f per position (round 4 before the batches were cut: 2.23, round 3: 2.38, round 2: 2.75, round 1: 3.79)" % (c["SQ_INSTS_VALU"] / pos),
 "SQ_INSTS_SALU": c["SQ_INSTS_SALU"], "SQ_INSTS_LDS": c["SQ_INSTS_LDS"], "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"],
 "clock_note": "GRBM_GUI_ACTIVE / 8 XCDs / launch time (%.2f ms in the kernel trace) = %.2f GHz under the profiler (max clock 2.4 GHz)" % (t, c["GRBM_GUI_ACTIVE"] / 8 / (t * 1e-3) / 1e9),
 "SQ_WAVE_CYCLES": c["SQ_WAVE_CYCLES"], "SQ_WAIT_ANY": c["SQ_WAIT_ANY"], "SQ_WAIT_INST_ANY": c["SQ_WAIT_INST_ANY"], "SQ_ACTIVE_INST_ANY": c["SQ_ACTIVE_INST_ANY"],
}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "SQ_INSTS_VALU", "clock_note")}, indent=1), out["per_genome_launches"]["hbm_bytes_per_launch"])
