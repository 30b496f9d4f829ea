#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_profile.sh -> summary.txt): HBM bytes per cut and VALU instructions per frame of the
bench kernel, stamped with the sha256 of the kernel sources they were measured on (bench.py refuses the numbers once those files change).

    python tools/make_traffic_json.py <summary.txt> <cuts per dispatch> [label of the summary file in profiles/] [power probe json line file]
    python tools/make_traffic_json.py --config mfcc40_libri|onthefly <dir of tools/collect.sh traffic>   (adds / replaces traffic.json["configs"][<name>])"""
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


CONFIG_SOURCES = {
    "mfcc40_libri": SOURCES,
    "onthefly": SOURCES + ["lhotse_amd/csrc/kernel_minibatch.hpp", "lhotse_amd/csrc/kernel_resample.hpp"],
}


def config_entry(name: str, root: str):
    """One of the other BASELINE configs: counters summed over ALL dispatches of the run per kernel (the mini-batches of `onthefly` all
    differ), divided by the passes over the workload the run made (= feature-kernel dispatches / feature launches per step), against the
    algorithmic bytes bench.py printed for the same workload."""
    import csv
    import glob
    from collections import defaultdict

    tot = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        per = defaultdict(lambda: [0.0, set()])
        for path in glob.glob(os.path.join(root, f"pass_{ctr}", "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = row.get("Kernel_Name", "")
                    if "hipfeat" not in k or row["Counter_Name"] != ctr:
                        continue
                    k = k.split("(")[0].replace("void ", "").replace("hipfeat::", "")
                    per[k][0] += float(row["Counter_Value"])
                    per[k][1].add(int(row["Dispatch_Id"]))
        tot[ctr] = {k: (v[0], len(v[1])) for k, v in per.items()}
    line = None
    for ln in open(os.path.join(root, "pass_FETCH_SIZE.log")):
        if ln.startswith("{"):
            line = json.loads(ln)
    assert line is not None, "no bench line in pass_FETCH_SIZE.log"
    roof = line["roofline"]
    algo = roof["algorithmic_bytes_per_launch"]
    parts = roof.get("algorithmic_bytes_parts") or {"feature_launches_per_step": 1}
    feat = [k for k in tot["FETCH_SIZE"] if k.startswith("fft512c_kernel")]
    assert len(feat) == 1, sorted(tot["FETCH_SIZE"])
    steps = tot["FETCH_SIZE"][feat[0]][1] / parts["feature_launches_per_step"]
    per_kernel, hbm = {}, 0.0
    for k in sorted(tot["FETCH_SIZE"]):
        fetch = tot["FETCH_SIZE"][k][0] * 1024 * 2 / steps  # KiB x 1024 x 2: the guide's gfx950 correction for wide coalesced reads
        write = tot["WRITE_SIZE"].get(k, (0.0, 0))[0] * 1024 / steps
        hbm += fetch + write
        per_kernel[k] = {"fetch_bytes_per_step": round(fetch), "write_bytes_per_step": round(write), "dispatches_per_step": round(tot["FETCH_SIZE"][k][1] / steps, 2)}
    for k, v in per_kernel.items():  # against the kernel's own algorithmic bytes
        if k.startswith("fft512c_kernel"):
            v["algorithmic_read"], v["algorithmic_write"] = parts.get("feature_read", None), parts.get("feature_write", None)
        elif "minibatch_prep" in k:
            v["algorithmic_read"], v["algorithmic_write"] = parts.get("resampler_read"), parts.get("resampler_write")
    entry = {
        "kernel": feat[0].split(",")[0] + ">" if "," in feat[0] else feat[0],
        "device_symbols": sorted(tot["FETCH_SIZE"]),
        "hbm_bytes_per_algorithmic_byte": round(hbm / algo, 4),
        "hbm_bytes_per_step": round(hbm),
        "algorithmic_bytes_per_step": algo,
        "steps_profiled": steps,
        "per_kernel": per_kernel,
        "source": f"tools/collect.sh traffic ({root}): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --config {name} --steps 2 --warmup 1`, "
                  "counters summed over all dispatches per kernel / passes over the workload",
        "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 under-count of 16 B/lane coalesced reads, MI355X_MICROARCH.md HBM section; the resampler's reads are "
                       "4 B/lane strided gathers for which the factor is uncalibrated -- its fetch figure is an upper bound); WRITE_SIZE KiB x 1024",
        "workload": line["config"]["workload"],
        "source_files": CONFIG_SOURCES[name],
        "source_sha256_16": bench.kernel_source_hash(CONFIG_SOURCES[name]),
    }
    if name == "onthefly":
        entry["infinity_cache_note"] = ("FETCH_SIZE / WRITE_SIZE count the L2's memory-side requests (TCC_EA0_RDREQ / WRREQ): traffic that the 256 MiB Infinity Cache "
                                        "absorbs behind the L2 is still counted, so these counters show that the perturbed tail LEAVES THE L2 (the feature launch fetches "
                                        "its whole input from the memory side), not whether it reaches HBM; the mini-batch's 25 MB tail fits the Infinity Cache, and no "
                                        "counter of this rocprofv3 separates MALL hits")
    return entry


def sq_counters(summary_path: str, kernel_prefix: str):
    """The SQ counters of tools/pmc_any.sh's summary for the feature kernel (per-dispatch medians), VERDICT r5 task 4."""
    out, take = {}, False
    for ln in open(summary_path):
        if ln.startswith("== "):
            take = kernel_prefix in ln
            if take:
                out["device_symbol"] = ln[3:].strip()
            continue
        if not take:
            continue
        m = re.match(r"\s+(\w+)\s+per-dispatch median ([0-9.e+-]+)", ln)
        if m:
            out[m.group(1)] = float(m.group(2))
        m = re.search(r"dispatch duration under profiling: median ([0-9.]+) us", ln)
        if m:
            out["duration_us_under_profiling"] = float(m.group(1))
    return out


def main():
    if sys.argv[1] == "--config":
        name, root = sys.argv[2], sys.argv[3]
        entry = config_entry(name, root)
        if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):  # tools/pmc_any.sh summary of the same config (separate --pmc passes)
            sym = {"mfcc40_libri": "fft512c_kernel<13, 12, 2, true>", "onthefly": "fft512c_kernel<13, 12, 0, true>"}[name]
            entry["sq_counters_per_dispatch"] = sq_counters(sys.argv[4], sym)
            if name == "onthefly":
                entry["sq_counters_per_dispatch_prep_launch"] = sq_counters(sys.argv[4], "minibatch_prep_inline_kernel")
            entry["sq_counters_source"] = f"tools/collect.sh pmc ({sys.argv[4]}): rocprofv3 --pmc SQ_* in two passes of `bench.py --config {name} --steps 2 --warmup 1`; per-frame table: profiles/r06_pmc_baseline_kernels.txt"
        path = os.path.join(ROOT, "profiles", "traffic.json")
        with open(path) as f:
            t = json.load(f)
        t.setdefault("configs", {})[name] = entry
        with open(path, "w") as f:
            json.dump(t, f, indent=1)
        print(json.dumps(entry, indent=1))
        return
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
    if len(sys.argv) > 4:  # shader clock the chip held while the bench kernel ran for seconds (rocm-smi samples, tools/collect.sh power)
        probe = json.loads(open(sys.argv[4]).readline())
        out["sclk_MHz_under_load"] = probe["sclk_MHz_median"]
        out["package_power_W_under_load"] = probe["package_power_W_median"]
        out["power_probe"] = sys.argv[4]
    try:  # the entries of the other configs survive a refresh of the headline entry
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            keep = json.load(f).get("configs")
        if keep:
            out["configs"] = keep
    except Exception:
        pass
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
