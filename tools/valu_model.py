#!/usr/bin/env python3
"""tools/valu_model.py [summary.json] -- the integer-VALU issue model behind bench.py's `roofline.valu`.

Runs on the CPU box (hipcc -S, no GPU).  For the three kernels of the headline configuration
(n = 2^17, p < 2^30, Montgomery: col_kernel<6, fwd, HALF>, row_mul_kernel, col16_kernel<64 points, inv>)
it takes

* the instruction MIX from the gfx950 listing of the current sources: every VALU opcode is priced at its
  measured issue cost per wave64 instruction on a SIMD-32 -- 2 cycles for v_add / v_sub / logic / v_mov,
  4 cycles for v_mul_lo/hi_u32, v_mad_u64_u32, v_mad_i64_i32, v_min/max, shifts and the other VOP3
  integer ops (profiles/ubench_intmul_r01.txt, profiles/ubench_valu2_r01.txt);
* the dynamic COUNT per wave from the rocprofv3 counters of a profile summary (SQ_INSTS_VALU / SQ_WAVES,
  profiles/<tag>_summary.json) when one is given -- the listing holds both sides of the wave-uniform
  branches (negacyclic twist, canonical store), so its static count is an upper bound and only its mix
  is used then; without a summary the static count is used and marked as such;

and writes profiles/valu_model.json: issue cycles and lane-ops per product and kernel, stamped with the
sha256 of the kernel sources (bench.py reports whether the stamp matches the build it runs).
"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ntt_amd", "csrc")

FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32",
             "v_ashrrev_i32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32",
             "v_cndmask_b32", "v_not_b32", "v_add_f32", "v_fma_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
             "v_nop"}

UNITS_PER_LAUNCH = 1024 / 19.0          # bench.py's default batch in 19 launches of up to 56 products (tools/profile.sh)

# (listing, substring of the mangled kernel name, SQ key in the profile summary, waves per product and pass)
KERNELS = {
    # round 3: the first forward pass is the software-pipelined walking kernel (its listing holds the walk's loop body
    # once plus the prologue; the dynamic count comes from the counters), the row kernel has a WIDE variant for p < 2^29
    "col_fwd": ("kernels_col.hip", "col_walk_prefetch_kernelINS_6Mont32ELb1ELi6ELb0E", "col_kernel_fwd", 2 * 4096 / 64),
    "row_mul": ("kernels_row.hip", "row_mul_kernelINS_6Mont32ELb1ELb0E", "row_mul_kernel", 64 * 256 / 64),
    "col_inv": ("kernels_col16.hip", "col16_kernelINS_6Mont32ELb1ELi2ELb1ELb0E", "col16_kernel_inv", 4096 * 4 / 64),
}


def source_stamp():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        # the device code: kernels and the headers they include (engine.hpp, rccl_dl.hpp are host-only: no kernel sees them)
        if f.endswith((".hip", ".hpp")) and f not in ("engine.hpp", "rccl_dl.hpp"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def listing(src, tmp):
    out = os.path.join(tmp, src + ".s")
    if not os.path.exists(out):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                        "-S", os.path.join(CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def mix(txt, key):
    start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l) and key in l.split(":")[0])
    ops = collections.Counter()
    for l in txt[start + 1:]:
        if "s_endpgm" in l:
            break
        if l.startswith("\t") and not l.strip().startswith((".", ";")):
            ops[l.split()[0]] += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    n = sum(valu.values())
    cyc = sum(v * (2 if re.sub(r"_e(32|64)$", "", k) in FULL_RATE else 4) for k, v in valu.items())
    return n, cyc, valu


def main():
    summary = json.load(open(sys.argv[1])) if len(sys.argv) > 1 else None
    out = {"_build": source_stamp(), "_count_source": sys.argv[1] if summary else "static listing (upper bound)",
           "kernels": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (src, key, sqkey, waves) in KERNELS.items():
            n_static, cyc_static, valu = mix(listing(src, tmp), key)
            # wave instructions per product: from the counters, SQ_INSTS_VALU of a launch / the products it holds
            # (a wave of the batch-walking forward column pass serves several products, so per-wave figures do
            # not convert); without counters, the static count of one pass through the kernel x its waves
            per_product = n_static * waves
            if summary and sqkey in summary and summary[sqkey].get("SQ_INSTS_VALU"):
                per_product = summary[sqkey]["SQ_INSTS_VALU"] / UNITS_PER_LAUNCH
            out["kernels"][name] = {
                "valu_insts_per_wave_static": n_static, "valu_insts_per_product": per_product,
                "issue_cycles_per_valu_inst": cyc_static / n_static,
                "issue_cycles_per_product": cyc_static / n_static * per_product,
                "lane_ops_per_product": per_product * 64,
                "top_ops": dict(sorted(valu.items(), key=lambda kv: -kv[1])[:8]),
            }
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_model.json"), "w"), indent=1)
    tot = sum(k["issue_cycles_per_product"] for k in out["kernels"].values())
    print(json.dumps({k: round(v["issue_cycles_per_product"]) for k, v in out["kernels"].items()}),
          "floor us/product:", 1e6 * tot / (256 * 4 * 2.4e9))


if __name__ == "__main__":
    main()
