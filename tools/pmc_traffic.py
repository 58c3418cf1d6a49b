#!/usr/bin/env python
"""Fold two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; CSV output,
--kernel-trace only) into per-kernel HBM traffic per launch.

Units / corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads -> doubled.  The doubling is calibrated in the same run
on maxpool_kernel, whose algorithmic bytes are known (reads its whole input once, writes a quarter).

usage: pmc_traffic.py <dir_fetch> <dir_write> <out.json> [provenance text] [batch_per_gpu | -] [forward_passes]

Besides the per-kernel table the output has "families": the kernel families of bench.py's roofline table (both Winograd
generations under "wino_conv_kernel", both filter-gradient generations under "conv_wgrad_kernel"), launch-weighted."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"wino4t_conv_kernel<(\d)", name)
    if m:   # one row per ITEM FORM (round 6; round 5 folded every form but <1> into "<2>", which mixed the 15 flattened transform-net launches <4> of a
        #     batch-32 step into the "per VGG launch" figure): <1> 16 tiles x 64 channels (transform net at small batches, small VGG16 grids), <2> 32 tiles
        #     x 64 channels and <3> 16 tiles x 128 channels (the VGG16 launches of the training batches), <4> flattened 16-tile lists (transform net at
        #     batch 32).  The family "wino4t_conv_kernel (VGG16 big items)" below = <2> + <3>.
        return "wino4t_conv_kernel<%s>" % m.group(1)
    m = re.search(r"(conv_igemm_kernel<[^>]*>|conv_wgrad_kernel<[^>]*>|wgrad2_kernel<[^>]*>|conv_stream_kernel<[^>]*>|gram_stream_kernel<[^>]*>|"
                  r"gram_bwd_kernel<[^>]*>|conv_bf16_\w+<[^>]*>|conv_bstream_kernel<[^>]*>|[a-z_0-9]+_kernel)", name)
    s = m.group(1) if m else name[:60]
    s = s.replace(" ", "")
    return s if s.startswith("wgrad2") else s.replace(",false", "").replace(",true", ",flat")


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    dfetch, dwrite, out = sys.argv[1:4]
    prov = sys.argv[4] if len(sys.argv) > 4 else ""
    fe, wr = collect(dfetch, "FETCH_SIZE"), collect(dwrite, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) & set(wr), key=lambda k: -(2 * sum(fe[k]) + sum(wr[k]))):
        f = sum(fe[k]) / len(fe[k])
        w = sum(wr[k]) / len(wr[k])
        kernels[k] = {"launches_sampled": len(fe[k]), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                      "traffic_bytes_per_launch": int((2 * f + w) * 1024)}
    fams = {"wino4t_conv_kernel (VGG16 big items)": r"^wino4t_conv_kernel<[23]>$", "wino4t_conv_kernel (transform net)": r"^wino4t_conv_kernel<[14]>$",
            "wino6 (VGG16, split-bf16 pipeline)": r"^wino6_",
            "wino_conv_kernel": r"^wino2?_conv_kernel$", "conv_wgrad_kernel": r"^(conv_wgrad_kernel<(?!1,4)|wgrad2_kernel<)",
            "conv_wgrad_kernel (Gram forward)": r"^(conv_wgrad_kernel<1,4>|gram_stream_kernel<)",
            "conv_igemm_kernel (Gram backward)": r"^gram_bwd_kernel<", "conv_igemm_kernel<32,2,1>": r"^conv_stream_kernel<"}
    families = {}
    for fam, pat in fams.items():
        ks = [k for k in kernels if re.search(pat, k)]
        n = sum(kernels[k]["launches_sampled"] for k in ks)
        if n:
            families[fam] = {"kernels": ks, "launches_sampled": n,
                             "fetch_bytes_per_launch": int(sum(2 * 1024 * kernels[k]["FETCH_SIZE_KiB"] * kernels[k]["launches_sampled"] for k in ks) / n),
                             "write_bytes_per_launch": int(sum(1024 * kernels[k]["WRITE_SIZE_KiB"] * kernels[k]["launches_sampled"] for k in ks) / n),
                             "traffic_bytes_per_launch": int(sum(kernels[k]["traffic_bytes_per_launch"] * kernels[k]["launches_sampled"] for k in ks) / n)}
    doc = {"_provenance": prov, "kernels": kernels, "families": families}
    try:     # the build these counters were taken from (bench.py quotes them only for the same sources)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from faststyle_amd import build as fsbuild
        doc["csrc_sha16"] = fsbuild.source_digest()
    except Exception as ex:
        doc["csrc_sha16"] = None
        print("pmc_traffic: no source digest (%s)" % ex, file=sys.stderr)
    if len(sys.argv) > 5 and sys.argv[5] != "-":
        doc["batch_per_gpu"] = int(sys.argv[5])
    if len(sys.argv) > 6:
        doc["forward_passes"] = int(sys.argv[6])     # launches_sampled / forward_passes = launches per batch
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:12]:
        print("%-40s %4d launches  %10.1f MB/launch" % (k, v["launches_sampled"], v["traffic_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
