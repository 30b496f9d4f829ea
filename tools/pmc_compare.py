#!/usr/bin/env python3
"""Side-by-side SQ counters of the three BASELINE kernels, per frame (tools/collect.sh pmc -> pmc_<config>/summary.txt).
    python tools/pmc_compare.py gpurun_out/<run> > profiles/rNN_pmc_baseline_kernels.txt"""
import importlib.util
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
sys.argv = [sys.argv[0]] + sys.argv[1:]
spec.loader.exec_module(bench)


def block(path, kernel):
    for bl in open(path).read().split("== ")[1:]:
        if bl.startswith(kernel):
            d = {}
            for ln in bl.splitlines():
                m = re.match(r"\s+(\w+)\s+per-dispatch median ([0-9.e+-]+)", ln)
                if m:
                    d[m.group(1)] = float(m.group(2))
                m = re.search(r"dispatch duration under profiling: median ([0-9.]+) us", ln)
                if m:
                    d["duration_us"] = float(m.group(1))
            return d
    raise SystemExit(f"{kernel} not in {path}")


def main():
    run = sys.argv[1]
    lens = bench.libri_like_lengths(8000, 1000)
    frames = {"fbank16k": 1e7, "mfcc40_libri": float(((lens + 80) // 160).sum())}
    fb = block(f"{run}/pmc_fbank16k/summary.txt", "void hipfeat::fft512c_kernel<13, 12, 0, false>")
    mf = block(f"{run}/pmc_mfcc40_libri/summary.txt", "void hipfeat::fft512c_kernel<13, 12, 2, true>")
    of = block(f"{run}/pmc_onthefly/summary.txt", "void hipfeat::fft512c_kernel<13, 12, 0, true>")
    pr = block(f"{run}/pmc_onthefly/summary.txt", "hipfeat::minibatch_prep_inline_kernel")
    # a 600 s mini-batch, two thirds of its cuts perturbed by 0.9 / 1.1: frames of the collated feature launch from its matrix-core count
    # (8 blocks per full frame quad in MODE 0): the launch covers ~ SQ_INSTS_MFMA / 8 frames
    frames["onthefly"] = of["SQ_INSTS_MFMA"] / 8.0
    print("# SQ counters of the three BASELINE kernels per FRAME (wave-level instruction / cycle counts divided by the frames of a dispatch);")
    print("# rocprofv3 --pmc in separate passes of `bench.py --config <c> --steps 2 --warmup 1` (tools/collect.sh pmc); MI355X, this round's box.")
    print(f"# frames per dispatch: fbank16k {frames['fbank16k']:.0f} (10 000 x 10 s), mfcc40_libri {frames['mfcc40_libri']:.0f} (8 000 LibriSpeech-like cuts), "
          f"onthefly ~{frames['onthefly']:.0f} (one 600 s mini-batch after speed perturbation)")
    hdr = ("counter", "fft512c<13,12,0,false> fbank-80", "fft512c<13,12,2,true> MFCC 40x40", "MFCC / fbank", "fft512c<13,12,0,true> on-the-fly")
    print("%-26s %30s %32s %13s %32s" % hdr)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES",
              "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
        a, b, c = fb[k] / frames["fbank16k"], mf[k] / frames["mfcc40_libri"], of[k] / frames["onthefly"]
        print("%-26s %30.2f %32.2f %13.3f %32.2f" % (k, a, b, b / a, c))
    a, b = fb["duration_us"] / frames["fbank16k"] * 1e3, mf["duration_us"] / frames["mfcc40_libri"] * 1e3
    print("%-26s %30.4f %32.4f %13.3f %32.4f" % ("ns per frame (profiled)", a, b, b / a, of["duration_us"] / frames["onthefly"] * 1e3))
    print()
    print("# minibatch_prep_inline_kernel (launch 1 of the on-the-fly pair: mixed-factor resampling + padding rows + tables), per dispatch:")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAVES"):
        print("%-26s %14.0f" % (k, pr[k]))
    print("%-26s %14.1f" % ("duration_us (profiled)", pr["duration_us"]))


if __name__ == "__main__":
    main()
