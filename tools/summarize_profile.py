#!/usr/bin/env python3
"""Condense gpurun_out/prof_TAG (tools/profile.sh) into profiles/TAG_*.{csv,md,json}.

Per kernel: launches, average duration (rocprofv3 --kernel-trace), and per-launch PMC averages.
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are reported in KiB by
rocprofv3 (unit correction x1024) and on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced streams (gfx950 correction x2 on the read side; our row kernels read 4 B per lane per
instruction, 256 B per wave instruction, so the x2 figure is given beside the raw one and the
raw..x2 interval is what DESIGN.md quotes).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if "col_walk_prefetch_kernel" in name:           # the walking first forward pass (round 3): the forward column kernel
        return "col_kernel_fwd"
    for k in ("col16_kernel", "col_kernel", "row_mul_kernel", "row_fwd_kernel", "row_inv_kernel", "small_kernel",
              "permute_kernel", "crt_reduce_kernel", "crt_merge_kernel"):
        if k in name:
            tag = ""
            if k in ("col_kernel", "col16_kernel"):
                # template args <MM, LAZY, LOGN1, INVERSE, HALF>
                m = re.search(r"col(?:16)?_kernel<[^,]+, *(\w+), *(\d+), *(\w+), *(\w+)[,>]", name)
                tag = "_inv" if m and m.group(3) == "true" else "_fwd"
            return k + tag
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # kernel trace
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    # copy the stats csv verbatim (it is small)
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w").write(open(f).read())
    # pmc
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                pmc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    summary = {}
    for k in sorted(dur):
        d = dur[k]
        s = {"launches": len(d), "avg_us": sum(d) / len(d), "min_us": min(d), "max_us": max(d),
             "total_ms": sum(d) / 1e3}
        for c, vals in pmc.get(k, {}).items():
            s[c] = sum(vals) / len(vals)
        if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
            s["hbm_read_bytes_raw"] = s["FETCH_SIZE"] * 1024
            s["hbm_read_bytes_x2"] = s["FETCH_SIZE"] * 2048
            s["hbm_write_bytes"] = s["WRITE_SIZE"] * 1024
        summary[k] = s
    json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as out:
        out.write(f"# rocprofv3 summary {tag}\n\n`rocprofv3 --kernel-trace --stats -- python3 bench.py "
                  "--no-cpu-baseline --steps 5 --warmup 2` (durations) and separate `--pmc` passes "
                  "(`--steps 1 --warmup 1`), see tools/profile.sh.  Per-launch averages.\n\n")
        out.write("| kernel | launches | avg us | total ms | read MiB raw / x2 | write MiB | VALU insts/wave | "
                  "LDS bank-conflict cyc / LDS active | TCC hit rate | VALU instr per SIMD-cycle | wait_inst_any / wave_cycles | "
                  "VMEM instr/wave rd + wr |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, s in summary.items():
            rd = f"{s.get('hbm_read_bytes_raw', 0) / 2**20:.2f} / {s.get('hbm_read_bytes_x2', 0) / 2**20:.2f}"
            wr = f"{s.get('hbm_write_bytes', 0) / 2**20:.2f}"
            valu = s.get("SQ_INSTS_VALU", 0) / max(s.get("SQ_WAVES", 1), 1)
            lds = f"{s.get('SQ_LDS_BANK_CONFLICT', 0):.0f} / {s.get('SQ_ACTIVE_INST_LDS', 0):.0f}"
            hit = s.get("TCC_HIT_sum", 0) / max(s.get("TCC_HIT_sum", 0) + s.get("TCC_MISS_sum", 0), 1)
            # SQ_BUSY_CYCLES is summed over the 32 shader engines, each 8 CUs x 4 SIMDs; SQ_ACTIVE_INST_VALU reads
            # identical to SQ_INSTS_VALU on this stack (an instruction count, not cycles), so the utilisation
            # figure is instructions per SIMD and busy cycle (the butterfly mix peaks at 1 / 3.8 = 0.263)
            per_simd = s.get("SQ_INSTS_VALU", 0) / max(s.get("SQ_BUSY_CYCLES", 1), 1) / 32
            s["valu_insts_per_simd_cycle"] = per_simd
            wait = s.get("SQ_WAIT_INST_ANY", 0) / max(s.get("SQ_WAVE_CYCLES", 1), 1)
            vm = (f"{s.get('SQ_INSTS_VMEM_RD', 0) / max(s.get('SQ_WAVES', 1), 1):.0f} + "
                  f"{s.get('SQ_INSTS_VMEM_WR', 0) / max(s.get('SQ_WAVES', 1), 1):.0f}")
            out.write(f"| {k} | {s['launches']} | {s['avg_us']:.1f} | {s['total_ms']:.2f} | {rd} | {wr} | "
                      f"{valu:.0f} | {lds} | {hit:.2f} | {per_simd:.3f} | {wait:.2f} | {vm} |\n")
        out.write("\nVALU instr per SIMD-cycle = SQ_INSTS_VALU / SQ_BUSY_CYCLES / 32 (SQ_BUSY_CYCLES is summed over the 32 shader "
                  "engines of 8 CUs x 4 SIMDs; SQ_ACTIVE_INST_VALU reads identical to SQ_INSTS_VALU on this stack, so it is not a "
                  "cycle count).  The butterfly mix peaks at 1 / 3.8 = 0.263 (profiles/ubench_mix_r02.txt); the counter passes run "
                  "with the profiler serialising the kernels, i.e. every kernel alone on the chip.  A wave of col_kernel_fwd walks the "
                  "batch (16 workgroups per column tile): its per-wave figures cover 53.9 / 16 = 3.4 products.\n")
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())
    # profiles/traffic.json feeds bench.py's roofline.traffic: PMC bytes per product and kernel of THIS build
    # (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md "HBM"), stamped with the sha256 of the kernel sources
    sys.path.insert(0, ROOT)
    import bench
    units = 1024 / 19.0                                  # bench.py's default batch in 19 launch groups of up to 56 products
    names = {"col_kernel_fwd": "col_fwd", "col16_kernel_fwd": "col_fwd", "row_mul_kernel": "row_mul",
             "col16_kernel_inv": "col_inv", "col_kernel_inv": "col_inv"}
    traffic = {"_build": bench.kernel_source_stamp(), "_profile": tag}
    for k, s in summary.items():
        if k in names and "hbm_read_bytes_x2" in s:
            traffic[names[k]] = {"bytes_per_unit": (s["hbm_read_bytes_x2"] + s["hbm_write_bytes"]) / units,
                                 "source": f"profiles/{tag}_summary.json (FETCH_SIZE x2 + WRITE_SIZE, KiB->B), 1024 products "
                                           "in 19 launches (53.9 per launch)"}
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
