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

# bench label -> regex on the demangled kernel name (unique instantiations of the BENCH workload only; later template
# parameters -- loader-wave count etc. -- may follow)
def _pc(w, bn, pro, epi):
    return rf"conv_pc_kernel<{w}, {bn}, {pro}, {epi}[,>]"


def _wg(w, ci, co, dz, pro):
    return rf"conv_wgrad3_kernel<{w}, {ci}, {co}, {dz}, {pro}[,>]"


LABELS = {
    "sed_conv3x3_wgrad_fused_c1:bwd b0c2 32->32 H6001 W64": _wg(64, 1, 1, 1, 2),
    "sed_conv3x3_fwd_c1:fwd b0c2 32->32 H6001 W64": _pc(64, 32, 2, 1),
    "sed_conv3x3_dgrad_c1_stats:bwd b0c2 32->32 H6001 W64": r"dgrad_c1a_kernel<8>",
    "sed_conv3x3_c1_gram:fwd b0c1 1->32 H6001 W64": r"conv_c1_gram_kernel",
    # round 3: weight + data gradient of a layer in one launch (dz only in LDS)
    "sed_conv3x3_bwd_fused_c1:bwd b0c2 32->32 H6001 W64": r"conv_bwd_fused_c1_kernel",
    "sed_conv3x3_bwd_fused:bwd b1c1 32->64 H3000 W32": r"conv_bwd_fused_kernel<32, 1, 2, 2, 0, 4[,>]",
    "sed_conv3x3_bwd_fused:bwd b1c2 64->64 H3000 W32": r"conv_bwd_fused_kernel<32, 2, 2, 1, 1, 2[,>]",
    "sed_logmel_fwd": r"frontend1024[bc]?_kernel",
}
for _b, _w, _cin, _c in ((1, 32, 32, 64), (2, 16, 64, 128), (3, 8, 128, 128)):
    _h = {1: 3000, 2: 1500, 3: 750}[_b]
    def _slice(ci, co):      # output-channel slice of the producer/consumer kernel: 128 in WR mode (>= 96 in, multiple of 128 out)
        return 128 if (ci >= 96 and co % 128 == 0) else (64 if co % 64 == 0 else 32)
    _t1, _t2 = f"b{_b}c1 {_cin}->{_c} H{_h} W{_w}", f"b{_b}c2 {_c}->{_c} H{_h} W{_w}"
    LABELS["sed_conv3x3_fwd:fwd " + _t1] = _pc(_w, _slice(_cin, _c), 0, 1)
    LABELS["sed_conv3x3_fwd:fwd " + _t2] = _pc(_w, _slice(_c, _c), 1, 1)
    LABELS["sed_conv3x3_fwd:bwd " + _t2] = _pc(_w, _slice(_c, _c), 0, 2)
    LABELS["sed_conv3x3_dgrad_poolstats:bwd " + _t1] = _pc(_w, _slice(_c, _cin), 0, 4)
    # round 5: layers with a multiple of 128 input channels run conv_wgrad_wide_kernel<W, 4, DZ, PRO> (csrc/sed_wgrad_wide.hip)
    def _wgx(w, cin, cout, dz, pro):
        if cin % 128 == 0 and cout % 64 == 0:
            return rf"conv_wgrad_wide_kernel<{w}, 4, {dz}, {pro}[,>]"
        return _wg(w, min(2, cin // 32), min(2, cout // 32), dz, pro)
    LABELS["sed_conv3x3_wgrad_fused:bwd " + _t2] = _wgx(_w, _c, _c, 1, 1)
    LABELS["sed_conv3x3_wgrad_fused:bwd " + _t1] = _wgx(_w, _cin, _c, 2, 0)


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
    # which kernels these counters belong to: bench.py compares this with the tree it runs from (traffic_stale)
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from csrc_sha import csrc_sha256
    out["__csrc_sha256__"] = csrc_sha256()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if k.startswith('__'):
            continue
        print(f"{v['hbm_bytes_per_launch'] / 1e9:8.3f} GB  rd {v['read'] / 1e9:6.3f}  wr {v['write'] / 1e9:6.3f}  {k}")


if __name__ == "__main__":
    main()
