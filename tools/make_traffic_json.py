#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_profile.sh -> summary.txt): HBM bytes per cut and VALU instructions per frame of the
bench kernel, stamped with the sha256 of the kernel sources they were measured on (bench.py refuses the numbers once those files change).

    python tools/make_traffic_json.py <summary.txt> <cuts per dispatch> [label of the summary file in profiles/] [power probe json line file]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

SOURCES = ["lhotse_amd/csrc/kernel_fft512c.hpp", "lhotse_amd/csrc/mel4_schedule.hpp", "lhotse_amd/csrc/fft_common.hpp", "lhotse_amd/csrc/fft512_common.hpp"]


def main():
    path, cuts = sys.argv[1], int(sys.argv[2])
    label = sys.argv[3] if len(sys.argv) > 3 else path
    vals, kernel = {}, None
    for line in open(path):
        m = re.match(r"== .*(fft512c_kernel<[^>]*>)", line)
        if m:
            kernel = m.group(1)
            vals = {}
            continue
        m = re.match(r"\s+(\w+)\s+per-dispatch median ([0-9.e+-]+)", line)
        if m and kernel and kernel not in vals.get("_done", ""):
            vals[m.group(1)] = float(m.group(2))
    assert kernel and "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and "SQ_INSTS_VALU" in vals, (kernel, sorted(vals))
    fetch = vals["FETCH_SIZE"] * 1024 * 2 / cuts  # gfx950 under-counts 16 B/lane coalesced reads by 2 (MI355X_MICROARCH.md, HBM section)
    write = vals["WRITE_SIZE"] * 1024 / cuts
    out = {
        "kernel": "fft512c_kernel<13>",
        "hbm_bytes_per_cut": round(fetch + write, 1),
        "fetch_bytes_per_cut": round(fetch, 1),
        "write_bytes_per_cut": round(write, 1),
        "valu_instr_per_frame": round(vals["SQ_INSTS_VALU"] / (cuts * 1000.0), 1),
        "valu_clk_per_instr": 2.7,
        "source": f"{label} (device symbol {kernel}) (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU, separate passes, one dispatch = {cuts} cuts x 10 s)",
        "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 under-count of 16 B/lane coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE KiB x 1024; "
                       "valu_instr_per_frame = wave-level VALU instructions / frames; valu_clk_per_instr = issue clocks of the measured instruction mix "
                       "(tools/ubench/valu_rate.hip: packed f32 / 3-source fma ~3, other VALU ~2)",
        "algorithmic_bytes_per_cut": 960000,
        "source_files": SOURCES,
        "source_sha256_16": bench.kernel_source_hash(SOURCES),
    }
    if len(sys.argv) > 4:  # shader clock the chip held while the bench kernel ran for seconds (rocm-smi samples, tools/r4_collect.sh)
        probe = json.loads(open(sys.argv[4]).readline())
        out["sclk_MHz_under_load"] = probe["sclk_MHz_median"]
        out["package_power_W_under_load"] = probe["package_power_W_median"]
        out["power_probe"] = sys.argv[4]
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
