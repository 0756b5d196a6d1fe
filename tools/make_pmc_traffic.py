#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 (gfx950: FETCH_SIZE
reports half of a wide coalesced read; calibrated here on ds_conf_kernel, which reads exactly 4*L*S*B bytes) + WRITE_SIZE[KB] * 1024.

    tools/make_pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv profiles/pmc_traffic.json [round-tag]
"""
import csv
import json
import sys
from collections import defaultdict

BENCH_NAME = [("fine_quad_kernel<1", "qta_fine_level[lists<=64]"), ("fine_quad_kernel<2", "qta_fine_level[lists>64]"),
              ("fine_level_dma_kernel<1", "qta_fine_level[lists<=64]"), ("fine_level_dma_kernel<2", "qta_fine_level[lists>64]"),
              ("fine_level_vreg_kernel", "qta_fine_level[lists<=64]"),
              ("quad_attn_kernel<8, 64, 0>", "qta_fine_level[lists<=64]"), ("quad_attn_kernel<8, 128, 0>", "qta_fine_level[lists>64]"),
              ("nchw_to_quads_kernel", "layout"),
              ("quad_attn_kernel<4, 128, 1>", "cascade_attn"), ("cascade_attn_dma_kernel", "cascade_attn"),
              ("cascade_quad_kernel", "cascade_attn"),
              ("coarse_fused_kernel", "qta_coarsest_level"), ("coarse_tile_kernel", "qta_coarsest_level"), ("window_match", "window_match"),
              ("ds_gemm16_kernel", "dual_softmax_gemm"), ("ds_gemm_kernel<", "dual_softmax_gemm[exact fp32; in split mode: the guarded fallback launch]"),
              ("ds_flagged_kernel", "dual_softmax_pass2"), ("ds_flagscan_kernel", "dual_softmax_pass2[flag scan]"),
              ("ds_sparse_kernel", "dual_softmax_pass2"), ("ds_conf_kernel", "dual_softmax_pass2[dense]"),
              ("ds_split_kernel", "dual_softmax_split_prepass"), ("ds_rownorm_kernel", "dual_softmax_split_prepass"), ("ds_fix_kernel", "dual_softmax_fix"),
              ("nchw_to_tokens_kernel", "layout[token-major]"), ("coarse_row_kernel", "coarse_row_kernel"),
              ("coarse_logits_kernel", "coarse_logits_kernel"), ("coarse_av_kernel", "coarse_av_kernel"),
              ("linear_nt_kernel", "linear_nt"), ("linear16_kernel", "linear_nt"), ("linear16s_kernel", "linear_nt"), ("linear16p_kernel", "linear_nt"), ("token_pool_kernel", "token_pool"), ("quad_pool_kernel", "token_pool")]


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    global SYMBOL
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for pat, name in BENCH_NAME:
                if pat in row["Kernel_Name"]:
                    SYMBOL.setdefault(name, row["Kernel_Name"].split("(")[0])
                    tot[name] += float(row["Counter_Value"])
                    cnt[name] += 1
                    break
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


SYMBOL = {}
fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    out[k] = {"hbm_bytes_per_launch": round(2 * f * 1024 + w * 1024), "fetch_KB_raw": round(f, 1), "write_KB_raw": round(w, 1),
              "launches_sampled": max(nf, nw), "symbol": SYMBOL.get(k)}
out["_round"] = sys.argv[4] if len(sys.argv) > 4 else "unlabelled"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
