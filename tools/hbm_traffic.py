#!/usr/bin/env python3
"""HBM bytes per launch from rocprofv3 PMC passes (tools/profile.sh): FETCH_SIZE and WRITE_SIZE counter CSVs ->
profiles/hbm_traffic_by_label.json, keyed by the bench.py kernel labels whose kernel template instantiation is
unique in the BENCH workload.  Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section):
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so
the bytes of wide streaming reads are twice the counter.
usage: hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys

# bench label -> regex on the demangled / mangled kernel name (unique instantiations only)
LABELS = {
    "sed_conv3x3_wgrad_fused:bwd b0c2 32->32 H6001 W64": r"conv_wgrad3_kernelILi64ELi1ELi1ELi1ELi1E|conv_wgrad3_kernel<64, 1, 1, 1, 1>",
    "sed_conv3x3_wgrad_fused:bwd b1c2 64->64 H3000 W32": r"conv_wgrad3_kernelILi32ELi2ELi2ELi1ELi1E|conv_wgrad3_kernel<32, 2, 2, 1, 1>",
    "sed_conv3x3_wgrad_fused:bwd b2c2 128->128 H1500 W16": r"conv_wgrad3_kernelILi16ELi2ELi2ELi1ELi1E|conv_wgrad3_kernel<16, 2, 2, 1, 1>",
    "sed_conv3x3_wgrad_fused:bwd b1c1 32->64 H3000 W32": r"conv_wgrad3_kernelILi32ELi1ELi2ELi2ELi0E|conv_wgrad3_kernel<32, 1, 2, 2, 0>",
    "sed_conv3x3_fwd:fwd b0c2 32->32 H6001 W64": r"conv_pc_kernelILi64ELi32ELi1ELi1E|conv_pc_kernel<64, 32, 1, 1(, false)?>",
    "sed_conv3x3_fwd:bwd b0c2 32->32 H6001 W64": r"conv_pc_kernelILi64ELi32ELi0ELi2E|conv_pc_kernel<64, 32, 0, 2(, false)?>",
    "sed_conv3x3_fwd:fwd b1c2 64->64 H3000 W32": r"conv_pc_kernelILi32ELi64ELi1ELi1E|conv_pc_kernel<32, 64, 1, 1(, false)?>",
    "sed_conv3x3_fwd:bwd b1c2 64->64 H3000 W32": r"conv_pc_kernelILi32ELi64ELi0ELi2E|conv_pc_kernel<32, 64, 0, 2(, false)?>",
    "sed_conv3x3_c1_fwd:fwd b0c1 1->32 H6001 W64": r"conv_c1_fwd_kernel",
    "sed_conv3x3_wgrad_fused_c1:bwd b0c2 32->32 H6001 W64": r"conv_wgrad3_kernelILi64ELi1ELi1ELi1ELi2E|conv_wgrad3_kernel<64, 1, 1, 1, 2>",
    "sed_conv3x3_fwd_c1:fwd b0c2 32->32 H6001 W64": r"conv_pc_kernelILi64ELi32ELi2ELi1E|conv_pc_kernel<64, 32, 2, 1(, false)?>",
    "sed_conv3x3_dgrad_c1:bwd b0c2 32->32 H6001 W64": r"conv_pc_kernelILi64ELi32ELi0ELi3E|conv_pc_kernel<64, 32, 0, 3(, false)?>",
    "sed_conv3x3_c1_wgrad:bwd b0c1 1->32 H6001 W64": r"conv_c1_wgrad_kernelIDF16bLb0E|conv_c1_wgrad_kernel<__bf16, false>",
    "sed_conv3x3_c1_wgrad_fused:bwd b0c1 1->32 H6001 W64": r"conv_c1_wgrad_kernel",
    "sed_conv3x3_dgrad_c1_stats:bwd b0c2 32->32 H6001 W64": r"dgrad_c1a_kernelILi8E|dgrad_c1a_kernel<8>",
    "sed_logmel_fwd": r"frontend1024b?_kernel",
}


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        ids[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: tot[k] / max(1, len(ids[k])) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for label, rx in LABELS.items():
        f = [v for k, v in fetch.items() if re.search(rx, k)]
        w = [v for k, v in write.items() if re.search(rx, k)]
        if len(f) != 1 or len(w) != 1:
            print("skip (not unique / missing):", label, len(f), len(w))
            continue
        rd = f[0] * 1024.0 * 2.0      # KB -> bytes, gfx950: 128-B read requests tallied at 64 B
        wr = w[0] * 1024.0
        out[label] = {"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr,
                      "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 on gfx950 per "
                                "MI355X_MICROARCH.md, B=32 T=6001"}
    # whole-step traffic: every dispatch of the profiled run / steps (the bench ran `--steps S --warmup W`: argv[4] = S + W)
    if len(sys.argv) > 4:
        nsteps = float(sys.argv[4])
        tot_r = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE") * 1024.0 * 2.0
        tot_w = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[2])) if r["Counter_Name"] == "WRITE_SIZE") * 1024.0
        out["__step_total__"] = {"hbm_bytes_per_step": (tot_r + tot_w) / nsteps, "read": tot_r / nsteps, "write": tot_w / nsteps,
                                 "note": f"all dispatches of the profiled run (incl. set-up: generator, first-step allocations) / {nsteps:g} steps"}
        print(f"step total: {(tot_r + tot_w) / nsteps / 1e9:.2f} GB/step (read {tot_r / nsteps / 1e9:.2f}, write {tot_w / nsteps / 1e9:.2f})")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if k.startswith('__'):
            continue
        print(f"{v['hbm_bytes_per_launch'] / 1e9:8.3f} GB  rd {v['read'] / 1e9:6.3f}  wr {v['write'] / 1e9:6.3f}  {k}")


if __name__ == "__main__":
    main()
