#!/usr/bin/env python3
"""bench.py -- image pairs/sec through CasMTR-4c's cascaded-matching hot path at 832x832 on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot-path chain (casmtr_amd/pipeline.py) over a batch of synthetic pairs that is already
resident in HBM: 12 QTAttB + CoarseMatching + 4 CascadeQTAttB + CascadeMatching (+NMS/selection) per pair, followed
by the gather of the match lists to rank 0.  Image pairs are independent, so ranks shard them with no data-path
collective (weak scaling: --batch pairs per GPU); RCCL carries the start-up weight broadcast and the match gather.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     : the dominant kernel's achieved rate vs the gfx950 peak, from HIP events recorded on the launch stream
  cpu_baseline : the CPU oracle (oracle/, C + OpenMP "port" of the reference algorithm) on the host cores, rank 0, N=1
  kernels      : per-kernel ms/step breakdown (same events)
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from casmtr_amd import _lib, dist as cdist  # noqa: E402
from casmtr_amd.pipeline import HotPath, HotPathConfig, algorithmic_work, make_synthetic_inputs  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# kernels with an algorithmic-work formula (candidates for the dominant-kernel roofline)
ROOFLINE_KERNELS = ("ds_gemm_kernel", "quad_attn_kernel<fine>", "quad_attn_kernel<cascade>", "window_match_kernel",
                    "ds_conf_kernel", "linear_nt_kernel")
PEAK_HBM_GBPS = 8000.0          # same guide, HBM3E peak (6.3 TB/s achievable)


def cpu_baseline(cfg, inp, budget_s=60.0):
    """The oracle's chain for ONE pair of the same synthetic workload, every call of the chain, all host cores."""
    import numpy as np

    import oracle

    n = lambda t: t[:1].detach().cpu().numpy()
    t0 = time.perf_counter()
    w = inp["weight"].cpu().numpy()
    for layer in range(cfg.coarse_layers):
        pairs = ((0, 0), (1, 1)) if layer % 2 == 0 else ((0, 1), (1, 0))
        for a, b in pairs:
            oracle.qtattb_forward([n(x) for x in inp[f"cq{a}"]], [n(x) for x in inp[f"ck{b}"]],
                                  [n(x) for x in inp[f"cv{b}"]], w, cfg.coarse_heads, cfg.coarse_topks)
    d8 = oracle.dual_softmax(n(inp["feat_8c0"]), n(inp["feat_8c1"]), cfg.hw8, cfg.hw8, cfg.coarse_temperature,
                             cfg.coarse_thr, cfg.coarse_border_rm, recip=True)
    tp01 = oracle.window_warp_idx(d8["next_idx_c01"], *cfg.hw8, cfg.window_size)
    tp10 = oracle.window_warp_idx(d8["next_idx_c10"], *cfg.hw8, cfg.window_size)
    tok = lambda x: np.ascontiguousarray(n(x).transpose(0, 2, 3, 1).reshape(1, -1, cfg.cascade_dim))
    for _ in range(cfg.cascade_cross_layers):
        _, i01 = oracle.cascade_attn(tok(inp["fq0"]), tok(inp["fk1"]), tok(inp["fv1"]), tp01, cfg.hw4, cfg.hw4, cfg.cascade_heads)
        _, i10 = oracle.cascade_attn(tok(inp["fq1"]), tok(inp["fk0"]), tok(inp["fv0"]), tp10, cfg.hw4, cfg.hw4, cfg.cascade_heads)
    m01 = oracle.window_match(n(inp["feat_4c0"]), n(inp["feat_4c1"]), i01, cfg.cascade_temperature, recip=True)
    m10 = oracle.window_match(n(inp["feat_4c1"]), n(inp["feat_4c0"]), i10, cfg.cascade_temperature, recip=True, want_conf=False)
    sel = oracle.nms_select(m01["next_conf"], m01["next_idx"], m10["next_idx"], cfg.hw4, cfg.hw4, cfg.nms_window,
                            cfg.cascade_test_thr, [(d8["next_conf_c01"], cfg.hw8, cfg.cascade_pre_thr)],
                            cfg.cascade_border_rm)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 5), "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"1 pair, full chain (12 QTAttB + dual-softmax + 4 CascadeQTAttB + window match x2 + NMS), {dt:.1f} s, "
                      f"OpenMP on all host cores; {len(sel['b_ids'])} matches"}, sel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="image pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--materialize-conf", action="store_true", help="also write the [B,L,S] conf_matrix (drop-in default)")
    ap.add_argument("--channels-last", action="store_true",
                    help="q/k/v pyramids arrive channels_last in memory (zero-copy token view); NOT the headline configuration")
    ap.add_argument("--with-callers", action="store_true",
                    help="SURVEY.md section 8 f.1 workload: enter through QuadtreeAttention / CascadeQuadtreeAttention on [B,N,C] "
                         "tokens (q/k/v + output projections and the pyramid inside the step); NOT the headline configuration")
    args = ap.parse_args()

    rank, world, local = cdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    _lib.lib()  # fail loudly here if the HIP library is missing

    cfg = HotPathConfig(materialize_conf=args.materialize_conf, callers=args.with_callers)
    B = args.batch
    model = HotPath(cfg).to(device)
    inp = make_synthetic_inputs(cfg, B, device, seed=1234 + rank, channels_last=args.channels_last)
    with torch.no_grad():
        model.qta.weight.copy_(inp["weight"])
    cdist.broadcast_parameters(model)          # RCCL broadcast of the (tiny) parameter buffer from rank 0

    def step():
        out = model(inp)
        return cdist.gather_matches(out, pairs_per_rank=B)  # counts all-gather + gather of [M,5] / [M] to rank 0

    # Warm-up.  Its last (up to) two steps run with every kernel timed by HIP events: that survey names the dominant kernel
    # and fills the per-kernel table.  In the timed region only the dominant kernel keeps its two event records per launch
    # (each record serialises ~3 us on the stream; timing all ~75 launches/step costs ~2 % of the step).
    n_survey = min(2, args.warmup)
    for _ in range(args.warmup - n_survey):
        step()
    survey = {}
    if n_survey:
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(n_survey):
            step()
        torch.cuda.synchronize()
        survey = _lib.prof_read()
        _lib.prof_enable(False)
    # the interpreter's cyclic GC otherwise fires a ~15 ms full collection every few steps (measured: tools/step_jitter.py);
    # freezing the start-up object graph is the usual serving-process setting
    gc.collect()
    gc.freeze()
    cdist.barrier()
    torch.cuda.synchronize()
    if survey:
        dominant = max((k for k in survey if k in ROOFLINE_KERNELS), key=lambda k: survey[k][0])
        _lib.prof_enable_only(dominant)
    else:
        _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    cdist.barrier()
    dt = time.perf_counter() - t0
    prof = _lib.prof_read()
    _lib.prof_enable(False)
    dt = cdist.max_over_ranks(dt)
    table, table_steps = (survey, n_survey) if survey else (prof, args.steps)

    if rank != 0:
        cdist.finalize()
        return
    ms_step = dt / args.steps * 1e3
    pairs_s = B * world * args.steps / dt
    work = algorithmic_work(cfg)
    kernels = {k: {"ms_per_step": round(v[0] / table_steps, 4), "launches_per_step": v[1] // table_steps} for k, v in table.items()}
    # ---- roofline of the dominant kernel (largest share of the step), from the HIP events recorded on the launch stream
    N0 = cfg.hw8[0] * cfg.hw8[1]
    fine_bytes = 4 * cfg.coarse_dim * (3 * (N0 + N0 // 4) + N0 + N0 // 4 + N0 // 16)  # q,k,v of both fine levels + acc in/out
    N4 = cfg.hw4[0] * cfg.hw4[1]
    proj_flops = (2 * cfg.coarse_layers * 4 * 2 * N0 * cfg.coarse_dim ** 2
                  + 2 * cfg.cascade_cross_layers * 4 * 2 * N4 * cfg.cascade_dim ** 2)   # per pair, callers mode only
    per_launch = {  # kernel -> (bound, algorithmic work per launch for B pairs, unit, formula)
        "ds_gemm_kernel": ("mfma", work["coarse_flops"] * B, "flop", "2*L*S*C*B"),
        "quad_attn_kernel<fine>": ("hbm", fine_bytes * B / 2, "B", "4*C*(3*(N0+N1)+N0+N1+N2)*B / 2 launches (levels 1 and 0 averaged)"),
        "quad_attn_kernel<cascade>": ("hbm", work["cascade_bytes"] * B, "B", "(4*C*4N + 8*(N/4)*25*2 + 8*N*K)*B"),
        "window_match_kernel": ("hbm", work["match_bytes"] * B / 2, "B", "(8*N*C + 12*N*K + 12*N)*B per direction"),
        "ds_conf_kernel": ("hbm", 4.0 * N0 * N0 * B, "B", "4*L*S*B (one read of the similarity matrix)"),
        "nchw_to_tokens_kernel": ("hbm", None, "B", "8*B*C*HW per tensor"),
        "linear_nt_kernel": ("mfma", proj_flops * B / 32 if cfg.callers else None, "flop",
                             "sum over the step's 32 launches of nprob*2*M*N*K, / 32 (12 x (3+1) coarse C=256, 4 x (3+1) cascade C=128)"),
    }
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from rocprofv3 --pmc passes
    pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
    cand = {k: v for k, v in prof.items() if k in per_launch and per_launch[k][1]}
    dname, (dms, dn) = max(cand.items(), key=lambda kv: kv[1][0])
    avg_ms = dms / dn
    bound, wk, unit, formula = per_launch[dname]
    if bound == "mfma":
        ach, peak, u = wk / (avg_ms * 1e-3) / 1e12, PEAK_FP32_MFMA_TFLOPS, "TFLOP/s"
    else:
        ach, peak, u = wk / (avg_ms * 1e-3) / 1e9, PEAK_HBM_GBPS, "GB/s"
    # what the gather kernels really pull on: 128-byte key / value rows out of L2 / Infinity Fabric.  The chip's ceiling for
    # that pattern depends on the working set per XCD (tools/probes/gather_bw.hip: 29 TB/s L2-resident, 12 at 11 MB, 9.3 at 22 MB)
    gather = None
    if dname == "quad_attn_kernel<fine>":
        tk = cfg.coarse_topks
        gbytes = 2 * 128 * cfg.coarse_heads * ((N0 // 16) * 4 * tk[0] + (N0 // 4) * 4 * tk[1]) * B / 2
        gather = {"gathered_row_bytes_per_launch": gbytes, "achieved_TBps": round(gbytes / (avg_ms * 1e-3) / 1e12, 2),
                  "ceiling_TBps_at_this_working_set": 9.3, "working_set_per_xcd_MB": round(2 * N0 * cfg.coarse_dim * 4 / 1e6, 1),
                  "source": "tools/probes/gather_bw.hip (DESIGN.md section 8)"}
    roof = {"kernel": dname, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": u, "frac": round(ach / peak, 4),
            "traffic": pmc.get(dname, {}).get("hbm_bytes_per_launch"), "avg_launch_ms": round(avg_ms, 4),
            "launches_per_step": dn // args.steps, "share_of_step": round(dms / args.steps / ms_step, 3),
            "work_per_launch": f"{wk:.4e} {unit} = {formula}"}
    if gather:
        roof["l2_gather"] = gather
    # every hot kernel against its own roof, for the record
    roofs = {}
    for k, (ms, n) in table.items():
        if k in per_launch and per_launch[k][1]:
            bd, wk2, _, _ = per_launch[k]
            a2 = wk2 / (ms / n * 1e-3) / (1e12 if bd == "mfma" else 1e9)
            roofs[k] = {"bound": bd, "achieved": round(a2, 1), "frac": round(a2 / (PEAK_FP32_MFMA_TFLOPS if bd == "mfma" else PEAK_HBM_GBPS), 4)}
    # chain-level rates for the record (all kernels, not just the dominant one)
    chain = {"algorithmic_GB_per_pair": round(work["total_bytes"] / 1e9, 4), "algorithmic_GFLOP_per_pair": round(work["total_flops"] / 1e9, 2),
             "chain_GBps_per_gpu": round(work["total_bytes"] * B * args.steps / dt / 1e9, 1),
             "chain_TFLOPs_per_gpu": round(work["total_flops"] * B * args.steps / dt / 1e12, 2)}
    line = {
        "metric": "image pairs/sec (832x832, CasMTR-4c) cascaded-matching hot path", "value": round(pairs_s, 3),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg.name}, random-init, batch of {B} synthetic 832x832 pairs per GPU "
                               f"(BASELINE.json configs[1]); 12 QTAttB + dual-softmax + 4 CascadeQTAttB + cascade matching + NMS per pair",
                   "pairs_per_gpu": B, "conf_matrix_materialized": bool(cfg.materialize_conf),
                   "qkv_memory_format": "channels_last" if args.channels_last else "contiguous (NCHW)",
                   "entry": ("QuadtreeAttention / CascadeQuadtreeAttention on tokens (+ q/k/v/out projections, pyramid; "
                             "section 8 f.1 workload, 90.7 GFLOP/pair of projections added)") if args.with_callers
                            else "QTAttB / CascadeQTAttB on given q/k/v (section 8 d)",
                   "matches_last_step": int(res["n_total"]) if res is not None else None, "parallelism": f"pairs sharded over {world} GPU(s)"},
        "roofline": roof, "rooflines_all": roofs, "chain": chain, "kernels": kernels,
        "kernels_measured_over": (f"last {n_survey} warm-up step(s), every kernel timed; the timed region times only the "
                                  f"roofline kernel") if survey else "the timed region",
    }
    if world == 1 and not args.no_cpu_baseline and not args.with_callers:
        cb, _ = cpu_baseline(cfg, inp)
        line["cpu_baseline"] = cb
        line["gpu_over_cpu"] = round(pairs_s / cb["value"], 1)
    print(json.dumps(line))
    cdist.finalize()


if __name__ == "__main__":
    main()
