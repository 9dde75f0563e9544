#!/usr/bin/env python3
"""bench.py -- depth-maps/s of the DMVSNet hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one depth map: the full ``MVSNet.forward`` (FeatureNet + 3 stages x (main + refine) pass) on
synthetic inputs of BASELINE config 2 (DTU eval 1600x1184, 5 views, 64/32/8 hypotheses), inputs resident in
HBM before the timed region.  N > 1 (default ``--mode auto``): the timed region runs the N ranks as independent replicas --
every rank processes its own reference views (the depth maps of a scan are independent units): no data-path
collective, weak scaling, this is `value` -- and THE SAME LINE then carries `latency_mode`: north_star's partition, the source
views of ONE depth map sharded over a view group with RCCL on the data path (reduce_scatter along H + halo send / recv +
all-gather, H-slab regularisation; view group = the largest divisor of N that is <= V - 1, so 5 views on 8 GPUs run as 2 groups
x 4), with the bytes every collective moved per rank and `depth_rel_vs_unsharded` measured in the run.

Rank 0 prints ONE JSON line.  Besides the driver's keys it carries
  roofline      dominant kernel family (by time), HIP-event timed inside the timed region
  roofline_all  the same for every kernel family
  ms_per_stage  HIP-event spans of features / stage1 / stage2 / stage3
  cpu_baseline  the oracle (CPU restatement, "port") timed on this host on a bounded sample
  parity        the HIP path vs that oracle run, same inputs: depth rel-L1 / max-abs per stage, flipped selections
  aten_gpu_baseline  context only: the oracle's ATen op sequence with its tensors on this GPU (stock MIOpen kernels)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {"dtu": "the reference's DTU eval recipe (scripts/dtu_test.sh: 48/32/8, ratios 4/2/1, --inverse_depth; not a BASELINE line)",
             "tnt": "the reference's Tanks&Temples recipe (scripts/tank_test.sh + tank_test_config.py max_h 1080 / max_w 2048; not a BASELINE line)",
             "c1": "BASELINE configs[0]", "c2": "BASELINE configs[1]", "c3": "BASELINE configs[2] (on one GPU)",
             "c4": "BASELINE configs[3] (on one GPU)",
             "c5": "BASELINE configs[4] as a declared EXTENSION (4-stage pyramid on the three FPN levels; the reference "
                   "cannot express it; on one GPU)"}
DATASETS = {"dtu": "DTU (reference recipe)", "tnt": "Tanks&Temples (reference recipe)", "c1": "DTU scan1", "c2": "DTU", "c3": "DTU", "c4": "Tanks&Temples", "c5": "BlendedMVS",
            "c3_small": "DTU (quarter-size test shape)", "c2_small": "DTU (quarter-size test shape)"}
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (guide: 6.29 TB/s measured with a float4 copy)
FP32_PEAK_TF = 157.3      # fp32 vector == fp32-input MFMA peak (MI355X_MICROARCH.md)
VALU_PK_PEAK_TF = 113.7   # v_pk_fma_f32 with every SIMD busy, measured (scripts/dev/ub/mfma4.hip, profiles/r04_r_ub_mfma4.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--mode", default="auto", choices=["auto", "replicas", "view-shard", "view-shard-rows"],
                    help="N > 1: auto (default, what the driver's scaling runs time) = `value` from N replicas (one depth map per rank, "
                         "no data-path collective, weak scaling) AND, in the same line, `latency_mode` from the hybrid view shard "
                         "(view-shard-rows with --view-group = the largest divisor of N that is <= V - 1: N = 2 -> 2, 4 -> 4, 8 -> 2 "
                         "groups x 4 at 5 views) with the bytes per collective per rank and depth_rel_vs_unsharded; replicas = only "
                         "the replicas; view-shard-rows (v2) = ONE depth map over a view group: source views sharded, reduce_scatter "
                         "along H + halo exchange, H-slab regularisation -- with --view-group G < N the job is N / G such groups "
                         "(hybrid); view-shard (v1) = all-reduce of the similarity volume with the regularisation replicated: kept for "
                         "parity tests only -- it moves 383 MB per depth map at config 2 to parallelise K1, 11 %% of the step, and "
                         "cannot pay on any link speed")
    ap.add_argument("--view-group", type=int, default=0,
                    help="ranks per view group in the view-shard modes (0 = all N ranks).  N / G groups work on different reference "
                         "views (replicas of groups), the G ranks of a group share one depth map: 5 views on 8 GPUs = 2 groups x 4 "
                         "(SURVEY.md 8e: config 2 at 8 GPUs), so no rank is left without a source view")
    ap.add_argument("--full-outputs", action="store_true", help="(the default at N = 1 since r05; kept for older command lines)")
    ap.add_argument("--no-full-outputs", action="store_true",
                    help="skip the extra pass (10 depth maps, outside the timed region) that times the forward with prob_volume and "
                         "depth_values materialised -- everything the reference's forward returns -- and reports it as "
                         "value_full_outputs; `value` stays the eval setting (config.outputs)")
    ap.add_argument("--conv-backend", default="auto", choices=["auto", "direct", "mfma"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the newest committed profiles/*pmc_fetch_write.txt instead of measuring it in "
                         "this run (two ~20 s child runs under rocprofv3 --pmc, N = 1 only)")
    ap.add_argument("--no-aten-gpu-baseline", action="store_true",
                    help="skip the context number `aten_gpu_baseline` (the oracle's ATen ops on this GPU, ~10-60 s)")
    ap.add_argument("--no-coarse", action="store_true", help="A/B: K3w / K3 for conv4 / conv6 instead of the register-stationary K3r (ops.use_coarse = False)")
    ap.add_argument("--no-zmarch", action="store_true", help="A/B: K3w for conv2 instead of the z-marching K3z (ops.use_zmarch = False)")
    ap.add_argument("--no-prob-fused", action="store_true", help="A/B: `prob` and K4 as two kernels in every pass (ops.use_prob_fused = False) instead of the fused head + dmvs_depth_select where D is 4 or 8")
    ap.add_argument("--no-wino", action="store_true", help="A/B: direct-form K3 for the stride-1 3x3 layers too (ops.use_wino = False)")
    ap.add_argument("--no-c8", action="store_true", help="A/B: FeatureNet conv0.0 / conv0.1 on the direct-form K3 kernel (ops.use_c8 = False)")
    ap.add_argument("--no-c8-fused", action="store_true", help="A/B: FeatureNet conv0.0 and conv0.1 as two K3s launches (ops.use_c8_fused = False)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--split-probe", type=int, default=0, choices=[0, 3, 6],
                    help="SECONDARY line only (VERDICT r05 item 3): after the timed region, run the same steps once more with every "
                         "conv1 of the regularisation nets on the bf16-split probe kernel (3 or 6 term products, fp32 accumulation; "
                         "csrc/conv3d_split.hip) and report it as `value_split` with the depth rel-L1 against the fp32 product.  "
                         "`value` and `dtype` are never touched by it")
    ap.add_argument("--budget-s", type=float, default=600.0,
                    help="wall-clock budget of the whole command (the timed region is a fraction of a second; the rest are context "
                         "legs outside it).  Once more than HALF of it is used, the optional legs still ahead are dropped in this "
                         "order: aten_gpu_baseline, the K1 SQ-counter pass, the PMC traffic passes, k1_coherent, the single-stream "
                         "pass; `legs_s` in the line records the wall time of every leg, `legs_dropped` what was skipped")
    ap.add_argument("--latency-budget-s", type=float, default=150.0,
                    help="N > 1, --mode auto: wall-clock limit of the `latency_mode` leg (the hybrid view shard behind the timed "
                         "region).  The replicas result is complete before that leg starts; if the leg raises on any rank or is still "
                         "running when the limit expires (a collective that never returns), rank 0 prints the line with `value` from "
                         "the replicas and `latency_mode: {\"error\": ...}` and every rank leaves with exit code 0 -- the driver's "
                         "scaling record never depends on the secondary leg")
    ap.add_argument("--inject-latency-fault", default="", help=argparse.SUPPRESS)   # tests: "raise:RANK" / "hang:RANK" inside the latency leg
    ap.add_argument("--maps-in-flight", type=int, default=1,
                    help="depth maps issued concurrently on alternating HIP streams (throughput mode of a scan: its "
                         "reference views are independent); every step is still one full depth map")
    ap.add_argument("--launch-log", default=None,
                    help="write the family of every K1..K4 launch of this process, in host order, to this JSON file "
                         "(joined with rocprofv3's dispatch order by scripts/pmc_summary.py)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo only to exercise the multi-rank code on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0 (1-GPU box testing, with gloo)")
    ap.add_argument("--no-async-topdown", action="store_true",
                    help="keep FeatureNet's level-2/3 outputs on the main stream instead of a third stream under stage 1 "
                         "(MVSNet.feature_async_topdown, default on: 75.3 vs 74.2 depth-maps/s)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the captured HIP graph of a forward (MVSNet.use_graph: the same ~170 kernels on the same streams, "
                         "enqueued as ONE graph launch) instead of launching every kernel from the host.  Same-box A/B r04, 4 "
                         "alternating runs each: 90.03 (graph) vs 90.07 (eager) depth-maps/s -- the step is GPU-bound, the graph only "
                         "frees the host thread (r02: 73.3-74.9 vs 75.3); opt-in")
    ap.add_argument("--no-graph", action="store_true", help="(the default; kept for the r04 A/B command lines)")
    ap.add_argument("--feature-dtype", default="f32", choices=["f32", "f16"],
                    help="f16: FeatureNet outputs stored as fp16, K1 accumulates in fp32 (the BASELINE configs[4] extension; "
                         "never the headline: the line says dtype 'f32 (fp16 features)')")
    ap.add_argument("--tune", action="append", default=[],
                    help="name=value for dmvs_tune (repeatable): A/B knobs of the kernels, e.g. k3_deconv_prefetch=0")
    ap.add_argument("--single-stream", action="store_true",
                    help="run the two regularisation branches back to back (clean per-kernel durations for profiles)")
    return ap.parse_args()


def cpu_baseline(cfg):
    """Oracle (CPU restatement, kind "port") on the host cores, bounded sample.  First the same workload with both
    image axes divided by 4 (1/16 of the pixels; every view / stage / pass kept); if that finishes fast enough
    that the full-size depth map fits the ~30 s budget, the full workload is timed instead (1 depth map).
    Thread count is capped at 32: on the 256-thread GPU-box host the ATen CPU ops of this size get slower beyond.
    Returns (json entry, the oracle's output dict of the timed sample, (H, W) of the sample): the outputs are what
    `parity` compares the HIP path with -- the checker's result is no longer thrown away."""
    from dmvsnet_amd import MVSNet, synth
    from oracle import dmvs_oracle

    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
    sd = synth.synth_state_dict(net.state_dict(), 0)

    def run(H, W):
        imgs, proj, dv = synth.synth_inputs(H, W, cfg["V"], 0)
        t0 = time.time()
        out = dmvs_oracle.mvsnet_forward(sd, cfg["ndepths"], cfg["ratios"], imgs, proj, dv,
                                         inverse_depth=cfg.get("inverse", False))
        return time.time() - t0, out

    run(64, 64)  # warm the thread pool
    # probe at 1/16 of the pixels, then time the largest of {full size, 1/4 of the pixels} predicted to stay within
    # ~45 s: the reported sample is 10-45 s of CPU work on every host seen so far (29 s full size on the fast boxes,
    # ~10 s at quarter size on the slow ones)
    H, W = cfg["H"] // 4 // 32 * 32, cfg["W"] // 4 // 32 * 32
    dt, _ = run(H, W)
    frac = (H * W) / float(cfg["H"] * cfg["W"])
    if dt / frac < 45.0:
        H, W, frac = cfg["H"], cfg["W"], 1.0
    else:
        H, W = cfg["H"] // 2 // 32 * 32, cfg["W"] // 2 // 32 * 32
        frac = (H * W) / float(cfg["H"] * cfg["W"])
    dt, out = run(H, W)
    entry = {"value": frac / dt, "unit": "depth-maps/s", "cores": cores, "kind": "port",
             "sample": f"1 depth map of the workload at {W}x{H} ({frac:.4f} of the pixels, all views/stages/passes), "
                       f"{dt:.1f} s wall on {cores} threads" + ("" if frac == 1.0 else ", scaled by the pixel ratio")}
    return entry, out, (H, W)


def parity_block(gpu_out, ref_out, nstage, size):
    """The HIP path against the oracle ON THE SAME INPUTS (seed 0, the size the CPU baseline was timed at; the full
    workload on every host fast enough for it): per stage the relative L1 and the max-abs error of the depth map (mm),
    the mean abs error of the confidence, and the share of pixels whose dual-depth SELECTION flipped -- which of the two
    regressed depths of a (small | huge) pair is the smaller one decides what DepthNet.forward / .refine pick for a
    checkerboard cell (mvsnet.py:25-56, 80-91), so an order flip between the two implementations is counted for the main
    pass's four estimates and for the refine pass's."""
    rel, mx, conf, flip = [], [], [], []
    for s in range(nstage):
        g, r = gpu_out[f"stage{s + 1}"], ref_out[f"stage{s + 1}"]
        d, dr = g["depth"].double().cpu(), r["depth"].double()
        rel.append(float((d - dr).abs().mean() / dr.abs().mean()))
        mx.append(float((d - dr).abs().max()))
        conf.append(float((g["photometric_confidence"].cpu() - r["photometric_confidence"]).abs().mean()))
        fl = torch.zeros(d.shape[-2:], dtype=torch.bool)
        for key in ("depth_sub_plus", "depth_sub_plus_refine"):
            a, b = g[key][0].cpu(), r[key][0]
            for c in (0, 2):
                fl |= (a[c] < a[c + 1]) != (b[c] < b[c + 1])
        flip.append(100.0 * float(fl.float().mean()))
    return {"against": "oracle/dmvs_oracle.py (CPU restatement pinned by reference-generated fixtures), same seed-0 inputs",
            "size": f"{size[1]}x{size[0]}", "depth_rel_l1": rel, "max_abs_mm": mx, "confidence_mean_abs": conf,
            "flipped_selection_pct": flip, "bound": "north_star: depth rel-L1 <= 1e-3"}


def aten_gpu_baseline(cfg, dev, budget_s=100.0):
    """CONTEXT number, never the headline and not the product: the reference's op sequence -- the oracle, i.e. stock ATen /
    MIOpen kernels (grid_sample, conv3d, conv_transpose3d, batch_norm, softmax ...) -- with its tensors on THIS MI355X,
    outside the timed region.  First call = MIOpen's kernel search for ~60 conv shapes (untimed, its
    wall time reported); then up to 3 depth maps within the budget."""
    from dmvsnet_amd import MVSNet, synth
    from oracle import dmvs_oracle
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    net = MVSNet(cfg["ndepths"], cfg["ratios"], verbose=False)
    sd = {k: v.to(dev) for k, v in synth.synth_state_dict(net.state_dict(), 0).items()}
    imgs, proj, dv = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], 0)
    imgs, dv, proj = imgs.to(dev), dv.to(dev), {k: v.to(dev) for k, v in proj.items()}

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = dmvs_oracle.mvsnet_forward(sd, cfg["ndepths"], cfg["ratios"], imgs, proj, dv,
                                         inverse_depth=cfg.get("inverse", False))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    try:
        t_first, out = run()
        times = []
        while len(times) < 3 and (not times or sum(times) + t_first + times[-1] < budget_s):
            times.append(run()[0])
        best = min(times)
        entry = {"value": 1.0 / best, "unit": "depth-maps/s", "ms_per_map": 1e3 * best, "maps_timed": len(times),
                 "first_call_s": t_first,
                 "what": "the oracle's ATen op sequence (= the reference's) on this GPU through stock PyTorch-ROCm / MIOpen "
                         "kernels, fp32; context only -- not the product path, not in the timed region"}
        return entry, out
    except Exception as e:   # noqa: BLE001 -- a context number must not take the bench line down (e.g. out of memory)
        return {"error": repr(e)[:200]}, None
    finally:
        torch.cuda.empty_cache()


def k1_coherent(cfg, dev, reps=5):
    """K1 (warp + correlation) ALONE on this configuration's shapes with SPATIALLY COHERENT hypotheses -- the planes of a smooth
    depth map, as a trained network produces them -- next to the in-pipeline figure, whose hypotheses come from the random-weight
    benchmark network and are spatially incoherent (neighbouring pixels sample unrelated depths: wide source windows, more
    channel slabs, the per-tap global path).  Random features (the kernel's cost does not depend on feature values); six
    stage-passes, median of ``reps``; algorithmic bytes as for the pipeline figure."""
    from dmvsnet_amd import ops, synth
    if len(cfg["ndepths"]) != 3 or cfg.get("inverse", False):
        return None
    H, W, V = cfg["H"], cfg["W"], cfg["V"]
    cams = synth.synth_cameras(H, W, V)
    dv = synth.synth_depth_values().to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    tot_ms = tot_b = 0.0
    last, per_pass = None, {}
    for s in range(3):
        sc = 2 ** (2 - s)
        h, w, C, D = H // sc, W // sc, (32, 16, 8)[s], cfg["ndepths"][s]
        feats = [ops.hwc_to_q4(torch.randn(h, w, C, generator=g).to(dev)) for _ in range(V)]
        p12 = ops.relative_proj(cams[f"stage{s + 1}"][0].to(dev).contiguous())
        if s == 0:
            hyp, _ = ops.hypotheses_first(dv, D, h, w, False, True)
        else:
            hyp, _ = ops.hypotheses_next(last, dv, float(cfg["ratios"][s]), D, False, True)
        yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        last = (650.0 + 100.0 * torch.sin(xx / w * 6.0) + 50.0 * torch.cos(yy / h * 4.0)).float().contiguous()
        hyp_c = (last[None] + (torch.arange(4, device=dev).view(4, 1, 1) - 1.5) * (8.0, 4.0, 2.0)[s]).contiguous()
        for name, hy in (("main", hyp), ("refine", hyp_c)):
            Dp = hy.shape[0]
            ops.warp_corr(feats[0], feats[1:], p12, hy, layout="q4")
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops.warp_corr(feats[0], feats[1:], p12, hy, layout="q4")
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ms = sorted(ts)[len(ts) // 2]
            tot_ms += ms
            tot_b += 4.0 * (V * C * h * w + 2 * Dp * h * w + (h * w if isinstance(hy, ops.AffinePlanes) else Dp * h * w))
            per_pass[f"s{s + 1}.{name}"] = ms
        del feats
    torch.cuda.empty_cache()
    gbs = tot_b / tot_ms / 1e6
    return {"ms_per_map": tot_ms, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "ms_per_pass": per_pass,
            "what": "K1 alone, random features, hypotheses of a smooth depth map (coherent windows); the pipeline figure beside it "
                    "runs on the random-weight network's incoherent hypotheses"}


def live_k1_issue_side(config, cfg, timeout_s=120):
    """The issue-side view of K1 MEASURED IN THIS RUN: one short child run of this script under
    `rocprofv3 --pmc` with five SQ counters (counters only, no tracing), summed over the warp_corr_q4 dispatches of 2 depth
    maps.  Returns per-depth-map instruction totals and the LDS bank-conflict factor, or None when rocprofv3 is missing / fails."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    ctrs = ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES"]
    steps, warm = 2, 1
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        cmd = ["rocprofv3", "--pmc", *ctrs, "-d", os.path.join(tmp, "sq"), "-o", "p", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--config", config, "--steps", str(steps), "--warmup", str(warm),
               "--no-cpu-baseline", "--no-aten-gpu-baseline", "--no-kernel-timing", "--no-live-traffic", "--no-full-outputs", "--single-stream"]
        try:
            subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), capture_output=True, timeout=timeout_s, check=True)
            csvs = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(tmp, "sq")) for f in fs if f.endswith("counter_collection.csv")]
            if not csvs:
                return None
            tot, disp = {c: 0.0 for c in ctrs}, set()
            with open(csvs[0]) as f:
                for row in csv.DictReader(f):
                    if "warp_corr_q4" in row["Kernel_Name"] and row["Counter_Name"] in tot:
                        tot[row["Counter_Name"]] += float(row["Counter_Value"])
                        disp.add(row["Dispatch_Id"])
        except Exception:   # noqa: BLE001 -- a missing profiler must not take the bench line down
            return None
    nstage = len(cfg["ndepths"])
    if not disp or len(disp) % (2 * nstage):
        return None
    maps = len(disp) / (2.0 * nstage)   # 2 K1 launches (main + refine) per stage and depth map, settle steps included
    chans = (32, 16, 8) if nstage == 3 else (32, 32, 16, 8)
    useful = 0.0   # lane-FMAs the algorithm needs: samples x (4 C + 8)
    for si in range(nstage):
        sc = 2 ** (2 - (si if nstage == 3 else max(0, si - 1)))
        h, w = cfg["H"] // sc, cfg["W"] // sc
        useful += (cfg["V"] - 1) * (cfg["ndepths"][si] + 4) * h * w * (4 * chans[si] + 8)
    act, conf = tot["SQ_LDS_IDX_ACTIVE"], tot["SQ_LDS_BANK_CONFLICT"]
    return {"source": "measured in this run (rocprofv3 --pmc " + " ".join(ctrs) + f", {maps:.0f} depth maps, single stream)",
            "insts_valu_per_map": tot["SQ_INSTS_VALU"] / maps, "insts_lds_per_map": tot["SQ_INSTS_LDS"] / maps,
            "lds_conflict_factor": act / max(act - conf, 1.0),
            "valu_useful_frac": useful / 64.0 / max(tot["SQ_INSTS_VALU"] / maps, 1.0)}


FAMILY_PATTERNS = {"conv3d_mfma": ("mfma_kernel", "wino_kernel", "conv2d_c8_kernel", "conv0_fused_kernel", "coarse_kernel", "zmarch_kernel",
                                   "conv1_split_kernel"),
                   "warp_corr": ("warp_corr",), "prob_head": ("conv_cout2",),
                   "conv3d_direct": ("conv_direct", "deconv_direct"), "depth_regress": ("depth_regress",)}


def live_pmc_traffic(config, timeout_s=90):
    """HBM-side bytes per launch MEASURED IN THIS RUN (the committed-file figure could not be vouched for by the
    driver's line): two short child runs of this script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
    passes, counters only -- never combined with tracing), 2 depth maps each, single stream, with the launch log that
    attributes every dispatch to its kernel family (scripts/pmc_summary.py).  Returns the summary text (the format of
    profiles/*pmc_fetch_write.txt) or None when rocprofv3 is missing / fails; outside the timed region, ~20 s per pass."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    text = []
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            log = os.path.join(tmp, f"launch_{counter}.json")
            cmd = ["rocprofv3", "--pmc", counter, "-d", os.path.join(tmp, counter), "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--config", config, "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-aten-gpu-baseline", "--no-kernel-timing", "--no-live-traffic", "--no-full-outputs", "--single-stream", "--launch-log", log]
            try:
                subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), capture_output=True, timeout=timeout_s, check=True)
                csvs = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(tmp, counter)) for f in fs if f.endswith("counter_collection.csv")]
                if not csvs or not os.path.exists(log):
                    return None
                r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), "--launch-log", log, csvs[0]],
                                   capture_output=True, text=True, timeout=60, check=True)
                text.append(r.stdout)
            except Exception:   # noqa: BLE001 -- a missing profiler must not take the bench line down
                return None
    return "\n".join(text)


def pmc_traffic(live_text=None):
    """HBM-side bytes per launch for each kernel family: from this run's own PMC passes (``live_text``, live_pmc_traffic) or,
    without them, from the newest committed rocprofv3 PMC summary (profiles/*pmc_fetch_write.txt).  FETCH_SIZE and WRITE_SIZE
    are collected in separate passes, unit KiB; FETCH_SIZE doubled -- on gfx950 it reports half of a coalesced read,
    MI355X_MICROARCH.md, and the NCHW->HWC transposer in the same profile reads exactly 2x its FETCH_SIZE.
    Returns {family: (bytes_per_launch, source)}."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_fetch_write.txt")))
    if live_text is None and not files:
        return {}
    out = {}
    lines = (live_text if live_text is not None else open(files[-1]).read()).splitlines()
    source = "measured in this run" if live_text is not None else os.path.basename(files[-1])
    fam_lines = [l for l in lines if l.startswith("FAMILY ")]
    if fam_lines:   # "FAMILY <counter> family=<name> n=<launches> total=<KiB>" (scripts/pmc_summary.py --launch-log)
        tot = {}
        for l in fam_lines:
            m = re.match(r"FAMILY (FETCH_SIZE|WRITE_SIZE) family=(\S+) n=\s*(\d+) total=\s*([\d.]+)", l)
            if not m:
                continue
            # the two passes may run a different number of depth maps: normalise each by its own launch count
            t = tot.setdefault(m.group(2), [0.0, 0.0, 1])
            if m.group(1) == "FETCH_SIZE":
                t[0] += 2.0 * float(m.group(4)) * 1024 / int(m.group(3))
            else:
                t[1] += float(m.group(4)) * 1024 / int(m.group(3))
    elif live_text is not None:
        # a live run whose launch log did not match the dispatch count: kernel NAMES cannot tell a FeatureNet launch from a
        # regularisation launch, so no per-family figure is claimed -- traffic stays null
        return {}
    else:
        tot = {f: [0.0, 0.0, 0] for f in FAMILY_PATTERNS}
        for line in lines:
            m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)\s+total=\s*([\d.]+)\s+(.*)", line)
            if not m:
                continue
            for fam, pats in FAMILY_PATTERNS.items():
                if any(p in m.group(5) for p in pats):
                    if m.group(1) == "FETCH_SIZE":
                        tot[fam][0] += 2.0 * float(m.group(4)) * 1024
                        tot[fam][2] += int(m.group(2))
                    else:
                        tot[fam][1] += float(m.group(4)) * 1024
    for fam, (rd, wr, n) in tot.items():
        if n:
            out[fam] = ((rd + wr) / n, source)
    return out


def default_view_group(world, views):
    """Ranks per view group of the hybrid latency mode: the largest divisor of ``world`` that is <= the number of source views
    (every rank of a group then owns at least one source view): 5 views -> N = 2: 2, 4: 4, 8: 4 (two groups)."""
    return max(g for g in range(1, world + 1) if world % g == 0 and g <= max(1, views - 1))


def main():
    T0 = time.time()
    args = parse()
    legs, dropped = {}, []

    class leg:   # wall time of one leg of the command -> legs_s[name]
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            self.t = time.time()

        def __exit__(self, *a):
            legs[self.name] = legs.get(self.name, 0.0) + time.time() - self.t
            return False

    def affordable(name):
        """Optional legs are dropped once more than half of --budget-s is gone."""
        if time.time() - T0 > 0.5 * args.budget_s:
            dropped.append(name)
            return False
        return True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, the same launcher the
        # driver uses) and pass their single JSON line through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    import torch.distributed as dist

    from dmvsnet_amd import MVSNet, ops, synth

    if args.launch_log:
        ops.launch_log = []
    for kv in args.tune:
        from dmvsnet_amd import _lib
        k, v = kv.split("=")
        _lib.check(_lib.load().dmvs_tune(k.encode(), int(v)), f"dmvs_tune({k})")
    ops.use_wino = not args.no_wino
    ops.use_coarse = not args.no_coarse
    ops.use_zmarch = not args.no_zmarch
    ops.use_prob_fused = not args.no_prob_fused
    ops.use_c8 = not args.no_c8
    ops.use_c8_fused = not args.no_c8_fused
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL
        else:
            dist.init_process_group("gloo")

    cfg = synth.CONFIGS[args.config]
    net = MVSNet(cfg["ndepths"], cfg["ratios"], inverse_depth=cfg.get("inverse", False), verbose=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net = net.to(dev)
    net.return_prob_volume = False          # eval never reads it; parity tests ask for it
    net.return_depth_values = False         # nor the [1,D,H,W] hypothesis volumes (formed inside K1 / K4, row N2)
    net.conv_backend = args.conv_backend
    net.feature_dtype = args.feature_dtype
    net.two_streams = not args.single_stream
    use_graph = args.graph and not args.no_graph and args.maps_in_flight == 1 and not (world > 1 and args.mode != "replicas")   # (auto: no graph either)
    net.use_graph = use_graph
    net.feature_async_topdown = not args.no_async_topdown and not args.single_stream
    # auto (the default): the timed region = replicas; the hybrid view shard is timed afterwards into `latency_mode`
    auto = args.mode == "auto"
    mode = "replicas" if auto else args.mode
    vg, shard = world, None   # shard = (process group, rank in group, ranks per group) of the view-shard pass
    if world > 1 and (auto or mode in ("view-shard", "view-shard-rows")):
        vg = args.view_group or (default_view_group(world, cfg["V"]) if auto else world)
        if world % vg:
            raise SystemExit(f"--view-group {vg} does not divide the {world} ranks")

        def make_shard():
            if vg == world:
                return (dist.group.WORLD, rank, world)
            # hybrid: world / vg groups of vg ranks; new_group is collective over ALL ranks, for every group
            groups = [dist.new_group(ranks=list(range(g * vg, (g + 1) * vg))) for g in range(world // vg)]
            return (groups[rank // vg], rank % vg, vg)
        if not auto:   # (auto: the groups are made inside the guarded latency leg, behind the timed region)
            shard = make_shard()
            net.set_view_shard(*shard, shard_rows=mode == "view-shard-rows")
    n_groups = world // vg if mode != "replicas" else world   # depth maps in flight per step

    def inputs(seed):
        i, p, d = synth.synth_inputs(cfg["H"], cfg["W"], cfg["V"], seed)
        return i.to(dev), {k: v.to(dev) for k, v in p.items()}, d.to(dev)

    # every rank (replicas) / every view group (hybrid) gets its own reference view (different seed); identical inputs inside a group
    imgs, proj, dv = inputs(rank if (world > 1 and mode == "replicas") else (rank // vg if world > 1 else 0))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lanes = [torch.cuda.Stream(device=dev) for _ in range(args.maps_in_flight)] if args.maps_in_flight > 1 else None

    # The host may run at most `ahead` steps in front of the GPU: an unbounded run-ahead keeps every step's
    # temporaries (GBs, some tied to the side stream by record_stream) alive until the GPU catches up, the caching
    # allocator then has to hipMalloc new segments INSIDE the timed region (measured: 30 queued steps run at 19-23
    # maps/s instead of 73; 10 queued steps gave sporadic 56-64).  Two steps in flight keep the GPU fed (the host
    # needs ~3 ms to enqueue a 13.6 ms step) and are what a caller that consumes each result would do.
    ahead = max(2, args.maps_in_flight)

    def run_steps(k):
        out = None
        done = []
        for i in range(k):
            if i >= ahead:
                done[i - ahead].synchronize()
            if lanes is None:
                out = net(imgs, proj, dv)
            else:
                st = lanes[i % len(lanes)]
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    out = net(imgs, proj, dv)
            ev = torch.cuda.Event()
            ev.record(lanes[i % len(lanes)] if lanes is not None else torch.cuda.current_stream())
            done.append(ev)
        if lanes is not None:
            for st in lanes:
                torch.cuda.current_stream().wait_stream(st)
        return out

    # Settle first (untimed, before the W warmup steps of the contract): a fresh box sometimes runs its first steps 10-30 % slow
    # (allocator growth, clocks leaving idle; one r06 run of this command reported 81.7 where ten others on the same box gave 90.1-90.9:
    # two single steps had agreed with each other while both were still slow).  Chunks of 4 pipelined steps -- the way the timed region
    # runs them -- are repeated until two in a row agree to 2 % AND the last one is within 2 % of the fastest chunk seen (at most 10
    # chunks = 40 steps, ~0.5 s); every rank takes the same decisions (the chunk time is max-reduced), so the collectives of the
    # view-shard modes stay matched.
    prev = best = None
    for _ in range(3 if args.share_gpu else 10):   # (--share-gpu: N test ranks on ONE device never settle; keep their runs short)
        fence()
        ts = time.perf_counter()
        run_steps(4)
        fence()
        cur = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(cur, op=dist.ReduceOp.MAX)
        cur = float(cur.item())
        if prev is not None and abs(cur - prev) <= 0.02 * prev and cur <= 1.02 * best:
            break
        best = cur if best is None else min(best, cur)
        prev = cur
    legs["setup+settle"] = time.time() - T0
    with leg("timed_region"):
        run_steps(args.warmup)
        fence()
        # the timed region: EXACTLY `steps` depth maps, no instrumentation inside
        t0 = time.perf_counter()
        out = run_steps(args.steps)
        fence()
        dt = time.perf_counter() - t0
    # latency modes: the same ranks also run as independent replicas (throughput mode) so that one line carries both
    dt_rep = None
    if world > 1 and mode != "replicas":
        with leg("replicas_pass"):
            group, shard_rows, grank, gworld = net.view_group, net.shard_rows, net.view_rank, net.view_world
            net.set_view_shard(None, 0, 1)
            run_steps(max(1, args.warmup))
            fence()
            t2 = time.perf_counter()
            run_steps(args.steps)
            fence()
            dt_rep = time.perf_counter() - t2
            net.set_view_shard(group, grank, gworld, shard_rows=shard_rows)
    # north_star's partition -- the hybrid view shard (source views over a view group, RCCL reduce_scatter + halo exchange +
    # all-gather on the data path), timed like the main region (barrier + synchronize both sides, max over ranks).  `auto`: it runs
    # LAST, guarded (below), so that the replicas result never depends on it; `--mode view-shard-rows`: right here, unguarded.
    dt_lat, comm, rel_unsharded = None, None, None

    def latency_leg():
        nonlocal imgs, proj, dv
        dt_l, out_lat = None, out
        rep_inputs = (imgs, proj, dv)
        if auto:
            fault = args.inject_latency_fault.split(":") if args.inject_latency_fault else None
            if fault and int(fault[1]) == rank:
                if fault[0] == "raise":
                    raise RuntimeError("injected fault in the latency leg (test)")
                time.sleep(1e6)
            shard_ = make_shard()
            imgs, proj, dv = inputs(rank // vg)        # identical inputs inside a view group
            net.set_view_shard(*shard_, shard_rows=True)
            run_steps(max(1, args.warmup))
            fence()
            t4 = time.perf_counter()
            out_lat = run_steps(args.steps)
            fence()
            dt_l = time.perf_counter() - t4
        # what every collective moved, per rank, in ONE depth map (bytes handed to / received from the collective)
        net.comm_log = []
        run_steps(1)
        fence()
        log, net.comm_log = net.comm_log, None
        comm_ = {}
        for kind, sent, recv in log:
            c = comm_.setdefault(kind, {"calls_per_map": 0, "bytes_sent_per_rank": 0, "bytes_received_per_rank": 0, "largest_call_bytes": 0})
            c["calls_per_map"] += 1
            c["bytes_sent_per_rank"] += sent
            c["bytes_received_per_rank"] += recv
            c["largest_call_bytes"] = max(c["largest_call_bytes"], sent, recv)
        # ... and that the sharded forward computes the unsharded one: every rank runs its group's depth map alone
        d_shard = out_lat["depth"].clone()
        keep_shard = (net.view_group, net.view_rank, net.view_world, net.shard_rows)
        net.set_view_shard(None, 0, 1)
        d_one = run_steps(1)["depth"]
        fence()
        stat = torch.stack([((d_shard - d_one).abs().mean() / d_one.abs().mean()).double(),
                            torch.tensor(dt_l or 0.0, dtype=torch.float64, device=dev)])
        dist.all_reduce(stat, op=dist.ReduceOp.MAX)      # (rel-L1 against the unsharded forward, the leg's time): max over ranks
        if auto:
            imgs, proj, dv = rep_inputs
        else:
            net.set_view_shard(keep_shard[0], keep_shard[1], keep_shard[2], shard_rows=keep_shard[3])
        return (float(stat[1].item()) if dt_l is not None else None), comm_, float(stat[0].item())

    if world > 1 and mode == "view-shard-rows":
        with leg("latency_mode"):
            _, comm, rel_unsharded = latency_leg()
    split = None
    if args.split_probe and world == 1:
        with leg("split_probe"):
            d_fp32 = [out[f"stage{s_ + 1}"]["depth"].clone() for s_ in range(len(cfg["ndepths"]))]
            ops.split_probe = args.split_probe
            run_steps(max(2, args.warmup))
            fence()
            t5 = time.perf_counter()
            o_s = run_steps(args.steps)
            fence()
            dt_s = time.perf_counter() - t5
            ops.split_probe = 0
            split = {"value": args.steps / dt_s, "unit": "depth-maps/s", "ms_per_step": 1e3 * dt_s / args.steps, "terms": args.split_probe,
                     "depth_rel_l1_vs_fp32_product": [float((o_s[f"stage{s_ + 1}"]["depth"] - d).abs().mean() / d.abs().mean())
                                                      for s_, d in enumerate(d_fp32)],
                     "what": "SECONDARY, not the headline and not fp32 arithmetic in the strict sense: the 12 conv1 launches of a depth map "
                             "(8 -> 16, stride 2; module.py:363, 405) on the bf16-split probe kernel -- operands split into exact bf16 terms, "
                             f"{args.split_probe} term products on v_mfma_f32_16x16x32_bf16, fp32 accumulation; everything else unchanged"}
    dt_full = None
    # everything the reference's forward returns (prob_volume [1,4,D,H,W] + depth_values [1,D,H,W] per stage); the H-slab path of the
    # view shard never forms prob_volume: the pass then runs unsharded, one depth map per rank
    if (args.full_outputs or world == 1) and not args.no_full_outputs:
        with leg("full_outputs"):
            keep_shard = (net.view_group, net.view_rank, net.view_world, net.shard_rows)
            if net.shard_rows:
                net.set_view_shard(None, 0, 1)
            net.return_prob_volume = net.return_depth_values = True
            keep_graph, net.use_graph = net.use_graph, False
            n_full = min(args.steps, 10)
            run_steps(2)
            fence()
            t3 = time.perf_counter()
            run_steps(n_full)
            fence()
            dt_full = (time.perf_counter() - t3) / n_full
            net.return_prob_volume = net.return_depth_values = False
            net.use_graph = keep_graph
            full_groups = world if keep_shard[3] else n_groups
            if keep_shard[3]:
                net.set_view_shard(keep_shard[0], keep_shard[1], keep_shard[2], shard_rows=True)
    # second pass of the same `steps` maps with HIP events around every kernel launch (roofline numbers); the
    # ~700 events per map cost ~4 % wall time, which is why this pass is not the one `value` comes from
    timer, dt_instr = None, None
    if not args.no_kernel_timing:
        with leg("kernel_timing"):
            net.use_graph = False              # per-kernel HIP events need the individual launches
            net.feature_async_topdown = False  # and per-family busy times need FeatureNet off the stage-1 kernels' back
            ops.timer = ops.KernelTimer()
            ops.timer.reserve(700 * args.steps)
            t1 = time.perf_counter()
            run_steps(args.steps)
            fence()
            dt_instr = time.perf_counter() - t1
            timer, ops.timer = ops.timer, None
    # K3's fraction with the two regularisation branches back to back on ONE stream (no overlap between kernels):
    # reported beside the two-stream number, which leans on that overlap
    ss_frac = None
    by_label_ss = None
    if timer is not None and world == 1 and not args.single_stream and affordable("single_stream_pass"):
      with leg("single_stream_pass"):
        net.two_streams = False
        n_ss = min(args.steps, 10)
        run_steps(2)
        fence()
        ops.timer = ops.KernelTimer()
        ops.timer.reserve(700 * n_ss)
        run_steps(n_ss)
        fence()
        d = ops.timer.summary().get("conv3d_mfma")
        by_label_ss = (ops.timer.by_label(), n_ss)   # clean per-layer durations: nothing else runs beside a kernel
        ops.timer = None
        net.two_streams = True
        if d:
            ss_frac = {"achieved": d["flops"] / (d["ms"] * 1e-3) / 1e12, "ms_per_map": d["ms"] / n_ss,
                       "executed": d["exec_flops"] / (d["ms"] * 1e-3) / 1e12}
            ss_frac["frac"] = ss_frac["achieved"] / FP32_PEAK_TF
    assert torch.isfinite(out["depth"]).all()

    tmax = torch.tensor([dt, dt_rep or 0.0, dt_full or 0.0], dtype=torch.float64, device=dev)
    n_ranks, rccl_version = 1, None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)   # the collective itself counts the ranks it reached
        n_ranks = int(round(float(ones.item())))
        if args.dist_backend == "nccl":
            try:
                rccl_version = ".".join(map(str, torch.cuda.nccl.version()))
            except Exception:
                rccl_version = "unknown"
    dt, dt_rep = float(tmax[0].item()), float(tmax[1].item())
    dt_full = float(tmax[2].item()) if dt_full is not None else None   # max over ranks, like dt
    maps = args.steps * n_groups

    # the HIP-event readouts of the instrumented pass, taken HERE on the main thread: emit() may run on the watchdog thread while the
    # main thread sits in a device synchronisation that never returns, and must then not touch the HIP runtime at all
    timer_fams = timer.summary() if timer is not None else None
    timer_labels = timer.by_label() if (timer is not None and by_label_ss is None) else None
    timer_spans = timer.spans() if timer is not None else None

    def emit(lat_err):
        """Rank 0: build and print THE line (lat_err: why `latency_mode` carries no measurement, or None)."""
        nonlocal out
        res = {
            "metric": f"depth-maps/sec, {DATASETS.get(args.config, args.config)} {cfg['W']}x{cfg['H']} {cfg['V']}-view "
                      f"{len(cfg['ndepths'])}-stage ({'/'.join(map(str, cfg['ndepths']))} hyp)",
            "value": maps / dt, "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak" if (mode == "replicas" or vg != world) else "strong", "vs_baseline": None,
            "dtype": "f32" if args.feature_dtype == "f32" else "f32 (fp16 features)",
            "data": "synthetic (NumPy PCG64 images/cameras, random-init weights with randomised BatchNorm statistics)",
            "config": {"workload": f"{WORKLOADS.get(args.config, args.config)}: {cfg['W']}x{cfg['H']}, {cfg['V']} views, "
                                   f"{'/'.join(map(str, cfg['ndepths']))} hypotheses, {len(cfg['ndepths'])} stage(s) x (main + 4-plane refine)"
                                   + (", inverse-depth sampling" if cfg.get("inverse") else ""),
                       "parallelism": ("1 GPU" if world == 1 else
                                       (f"{world} replicas over reference views, no collective" + (" (the timed region; `latency_mode` = the view shard)" if auto else "") if mode == "replicas"
                                        else ((f"{world // vg} view groups (one reference view each) x " if vg != world else "") +
                                              (f"source views sharded over {vg} GPUs, all-reduce of the similarity volume per stage-pass"
                                               if mode == "view-shard" else
                                               f"source views sharded over {vg} GPUs, reduce_scatter along H + halo send/recv "
                                               "per stage-pass, H-slab regularisation, all-gather of the regression outputs")))),
                       "outputs": "depth + confidences of every stage (prob_volume / depth_values not materialised: the eval "
                                  "driver never reads them, SURVEY.md 8b; the full-size parity tests run the same setting)",
                       "k1": "warp_corr_q4 (quad-planar features, one launch configuration per shape)",
                       "k3": ("fp32 MFMA, direct implicit GEMM for every layer (--no-wino)" if args.no_wino else
                              "fp32 MFMA: Winograd F(2x2,3x3) for the stride-1 3x3 layers (conv0/2/4/6, FeatureNet conv1.x/2.x/out2/out3), "
                              "direct implicit GEMM for the stride-2 / transposed / 5x5 / 1x1 layers; FeatureNet conv0.0 + conv0.1: one register-only row sweep on the 4x4x1 MFMA (K3s)"),
                       "conv_backend": args.conv_backend,
                       "streams": 1 if args.single_stream else 2, "maps_in_flight": args.maps_in_flight,
                       "hip_graph": bool(use_graph)},
        }
        if world > 1:
            res["n_ranks"] = n_ranks
            res["dist_backend"] = args.dist_backend + (f" (RCCL {rccl_version})" if rccl_version else "")
        if world > 1 and mode != "replicas":
            res["view_group"] = vg
            res["latency_mode"] = {"value": n_groups * args.steps / dt, "unit": "depth-maps/s", "ms_per_map": 1e3 * dt / args.steps,
                                   "what": f"ONE depth map at a time per view group of {vg} ranks (" + res["config"]["parallelism"] + ")"}
            res["throughput_mode"] = {"value": world * args.steps / dt_rep, "unit": "depth-maps/s",
                                      "ms_per_step": 1e3 * dt_rep / args.steps,
                                      "what": f"{world} independent replicas on the same ranks, no collective"}
        if world > 1 and auto:
            # the default line of a multi-GPU run: `value` is the replicas rate (throughput_mode repeats it), latency_mode is
            # north_star's partition measured right behind it (mvsnet.py:131-146 summed over view shards)
            res["view_group"] = vg
            res["throughput_mode"] = {"value": maps / dt, "unit": "depth-maps/s", "ms_per_step": 1e3 * dt / args.steps,
                                      "what": f"{world} independent replicas, no data-path collective (= `value`)"}
            res["latency_mode"] = {"mode": "view-shard-rows", "view_group": vg, "view_groups": world // vg,
                                   "what": (f"{world // vg} view group(s) x {vg} ranks: ONE depth map per group at a time, its {cfg['V'] - 1} source views "
                                            f"sharded over the group ((v - 1) mod {vg}), reduce_scatter of the partial similarity volumes along H + halo "
                                            "send / recv per stage-pass, H-slab regularisation, all-gather of the regression outputs")}
            if lat_err is None and dt_lat:
                res["latency_mode"].update(value=(world // vg) * args.steps / dt_lat, unit="depth-maps/s", ms_per_map=1e3 * dt_lat / args.steps)
            else:   # the guarded leg did not finish: the replicas result above stands on its own
                res["latency_mode"].update(value=None, error=lat_err or "the leg returned no time",
                                           budget_s=args.latency_budget_s)
        if comm is not None and lat_err is None:
            res["latency_mode"]["collectives_per_map_per_rank"] = comm
            res["latency_mode"]["depth_rel_vs_unsharded"] = rel_unsharded
            res["latency_mode"]["depth_rel_vs_unsharded_bound"] = 2e-6
        if split is not None:
            res["value_split"] = split
        if dt_full is not None:
            res["value_full_outputs"] = {"value": full_groups / dt_full, "unit": "depth-maps/s", "ms_per_step": 1e3 * dt_full,
                                         "what": "the same forward with prob_volume [1,4,D,H,W] and depth_values [1,D,H,W] of every stage "
                                                 "materialised -- the full dict the reference's forward returns (mvsnet.py:254-258)"}
        if timer is not None:
            res["instrumented_ms_per_step"] = 1e3 * dt_instr / args.steps
            fams = timer_fams
            allr = {}
            for fam, d in fams.items():
                ms = d["ms"] / args.steps
                # ms = busy time (interval union: the two regularisation branches overlap on two streams)
                entry = {"launches_per_map": d["launches"] // args.steps, "ms_per_map": ms,
                         "avg_launch_us": 1e3 * d["ms"] / d["launches"],
                         "avg_launch_us_overlapped": 1e3 * d["sum_ms"] / d["launches"]}
                if fam in ("conv3d_mfma", "feature_mfma"):
                    a = d["flops"] / (d["ms"] * 1e-3) / 1e12
                    x = d["exec_flops"] / (d["ms"] * 1e-3) / 1e12
                    # achieved = ALGORITHMIC (direct-form) FLOPs / time; executed = the FLOPs the MFMAs really issue (the
                    # stride-1 3x3 layers run in Winograd F(2x2,3x3) form: 16 fp32 products per 2x2 patch instead of 36)
                    entry.update(bound="mfma", achieved=a, peak=FP32_PEAK_TF, unit="TFLOP/s", frac=a / FP32_PEAK_TF,
                                 traffic=None, algorithmic_gflop_per_map=d["flops"] / args.steps / 1e9,
                                 executed=x, executed_frac=x / FP32_PEAK_TF, executed_gflop_per_map=d["exec_flops"] / args.steps / 1e9)
                elif fam == "prob_head":
                    # K2: 432 MACs per voxel and branch on the VALUs (two output channels: no matrix shape pays, docs/kernels/K2): the
                    # roofline that bounds it is the packed-FMA issue rate, not HBM (VERDICT r05 Weak 4)
                    a = d["flops"] / (d["ms"] * 1e-3) / 1e12
                    entry.update(bound="valu", achieved=a, peak=VALU_PK_PEAK_TF, unit="TFLOP/s", frac=a / VALU_PK_PEAK_TF, traffic=None,
                                 peak_note="v_pk_fma_f32 rate all SIMDs sustain (profiles/r04_r_ub_mfma4.txt)",
                                 algorithmic_gflop_per_map=d["flops"] / args.steps / 1e9,
                                 hbm_gbs=d["bytes"] / (d["ms"] * 1e-3) / 1e9, algorithmic_mb_per_map=d["bytes"] / args.steps / 1e6)
                else:
                    a = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                    entry.update(bound="hbm", achieved=a, peak=HBM_PEAK_GBS, unit="GB/s", frac=a / HBM_PEAK_GBS,
                                 traffic=None, algorithmic_mb_per_map=d["bytes"] / args.steps / 1e6)
                allr[fam] = entry
            live = None
            if world == 1 and not args.no_live_traffic and affordable("live_pmc_traffic"):
                with leg("live_pmc_traffic"):
                    live = live_pmc_traffic(args.config)
            for fam, (b, src) in pmc_traffic(live).items():
                if fam in allr:
                    allr[fam]["traffic"] = b
                    allr[fam]["traffic_unit"] = "bytes/launch (rocprofv3 PMC FETCH_SIZE x 2 + WRITE_SIZE, single-stream pass, " + src + ")"
            dom = max(allr, key=lambda k: allr[k]["ms_per_map"])
            r = allr[dom]
            res["roofline"] = {"kernel": dom, "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"],
                               "unit": r["unit"], "frac": r["frac"], "traffic": r["traffic"],
                               "traffic_source": r.get("traffic_unit")}
            if "executed" in r:
                res["roofline"]["executed"] = r["executed"]
                res["roofline"]["executed_frac"] = r["executed_frac"]
                res["roofline"]["note"] = ("achieved = algorithmic direct-form FLOPs / busy time; executed = FLOPs the fp32 MFMAs "
                                           "issue (Winograd F(2x2,3x3) on the stride-1 3x3 layers)")
            # the largest single KERNEL of the step (by its summed launch durations per depth map), priced against the roofline that
            # bounds IT: the family figure above is an interval union over two streams (VERDICT r05 Weak 10)
            labs, nmaps, src = (by_label_ss[0], by_label_ss[1], "single-stream pass") if by_label_ss else (timer_labels, args.steps, "two-stream pass (durations include overlap)")
            if labs:
                # a layer of the small / huge branch of every stage-pass is ONE kernel configuration (conv11 = 12 launches of the same
                # deconv_mfma_kernel instantiation per depth map): group the launch labels by the layer name without stage / branch
                import re
                grp = {}
                for lab, d0 in labs.items():
                    key = re.sub(r"^(reg|ref)\d+\.((small|huge)\.)?", "", lab)
                    g_ = grp.setdefault(key, dict(family=d0["family"], launches=0, sum_ms=0.0, flops=0.0, exec_flops=0.0, bytes=0.0))
                    for f_ in ("launches", "sum_ms", "flops", "exec_flops", "bytes"):
                        g_[f_] += d0[f_]
                name, d = max(grp.items(), key=lambda kv: kv[1]["sum_ms"])
                ms = d["sum_ms"] / nmaps
                t_h, t_m = d["bytes"] / (HBM_PEAK_GBS * 1e9), d["exec_flops"] / (FP32_PEAK_TF * 1e12)
                if d["family"] == "prob_head":
                    lk = {"bound": "valu", "achieved": d["flops"] / (d["sum_ms"] * 1e-3) / 1e12, "peak": VALU_PK_PEAK_TF, "unit": "TFLOP/s"}
                elif t_h >= t_m or d["family"] in ("warp_corr", "depth_regress"):
                    lk = {"bound": "hbm", "achieved": d["bytes"] / (d["sum_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
                else:
                    lk = {"bound": "mfma", "achieved": d["exec_flops"] / (d["sum_ms"] * 1e-3) / 1e12, "peak": FP32_PEAK_TF, "unit": "TFLOP/s (executed)"}
                lk.update(name=name, family=d["family"], launches_per_map=d["launches"] // nmaps, ms_per_map=ms,
                          avg_launch_us=1e3 * d["sum_ms"] / d["launches"], frac=lk["achieved"] / lk["peak"], source=src)
                res["roofline"]["largest_kernel"] = lk
            if ss_frac is not None and "conv3d_mfma" in allr:
                allr["conv3d_mfma"]["single_stream"] = ss_frac
                if dom == "conv3d_mfma":
                    res["roofline"]["frac_single_stream"] = ss_frac["frac"]
            res["roofline_all"] = allr
            if "warp_corr" in allr:   # north_star names the warp kernel's achieved HBM-bandwidth fraction explicitly
                res["warp_hbm_frac"] = allr["warp_corr"]["frac"]
                # ... and the issue-side view: the kernel's own instruction streams against this run's time -- SQ
                # counters of a child rocprofv3 pass of THIS run; the newest committed summary
                # (profiles/*k1_sq_summary.json) only when that pass is switched off or unavailable, labelled as such
                import glob
                ms = allr["warp_corr"]["ms_per_map"]
                q = None
                if world == 1 and not args.no_live_traffic and affordable("k1_sq_pass"):
                    with leg("k1_sq_pass"):
                        q = live_k1_issue_side(args.config, cfg)
                if q is None and args.config == "c2" and args.feature_dtype == "f32":
                    sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "*k1_sq_summary.json")))
                    if sq:
                        q = json.load(open(sq[-1]))
                        q["source"] = "committed file " + os.path.basename(sq[-1]) + " (not measured in this run)"
                if q is not None:
                    valu_floor = q["insts_valu_per_map"] * 2.0 / 1024 / 2.1e9 * 1e3          # INSTS_VALU x 2 clk on 1024 SIMDs at 2.1 GHz
                    lds_floor = q["insts_lds_per_map"] * 4.0 * q["lds_conflict_factor"] / 256 / 2.1e9 * 1e3   # ds_read_b128 x 4 clk x conflicts on 256 CUs
                    allr["warp_corr"]["issue_side"] = {
                        "source": q["source"], "valu_useful_frac": q["valu_useful_frac"],    # needed FMAs / SQ_INSTS_VALU
                        "valu_issue_floor_frac": valu_floor / ms, "lds_floor_frac": lds_floor / ms,
                        "lds_conflict_factor": q["lds_conflict_factor"]}
                if world == 1 and affordable("k1_coherent"):
                    with leg("k1_coherent"):
                        coh = k1_coherent(cfg, dev)
                    if coh is not None:
                        allr["warp_corr"]["coherent_hypotheses"] = coh
                        res["warp_hbm_frac_coherent"] = coh["frac"]
            spans = timer_spans
            res["ms_per_stage"] = {k: v / args.steps for k, v in spans.items() if k != "end"}
        if world == 1 and not args.no_cpu_baseline:
            with leg("cpu_baseline"):
                res["cpu_baseline"], ref_out, (Hs, Ws) = cpu_baseline(cfg)
            # parity of THIS build on THIS box, in the line: the HIP path on the inputs the oracle just processed
            net.two_streams, net.feature_async_topdown = not args.single_stream, not args.no_async_topdown and not args.single_stream
            pi, pp, pd = synth.synth_inputs(Hs, Ws, cfg["V"], 0)
            gpu_out = net(pi.to(dev), {k: v.to(dev) for k, v in pp.items()}, pd.to(dev))
            torch.cuda.synchronize()
            res["parity"] = parity_block(gpu_out, ref_out, len(cfg["ndepths"]), (Hs, Ws))
            del gpu_out
        if world == 1 and not args.no_aten_gpu_baseline and affordable("aten_gpu_baseline"):
            del out
            torch.cuda.empty_cache()
            with leg("aten_gpu_baseline"):
                res["aten_gpu_baseline"], aten_out = aten_gpu_baseline(cfg, dev, budget_s=max(20.0, min(100.0, args.budget_s - (time.time() - T0) - 20.0)))
            if aten_out is not None and not args.no_cpu_baseline and (Hs, Ws) == (cfg["H"], cfg["W"]):
                # ATen's GPU kernels against ATen's CPU kernels on the same inputs: how far two stock implementations of
                # the reference's ops sit from each other (context for the product's own parity figures)
                d, r = aten_out["depth"].cpu(), ref_out["depth"]
                res["aten_gpu_baseline"]["depth_rel_l1_vs_cpu"] = float((d - r).abs().mean() / r.abs().mean())
        if args.launch_log:
            with open(args.launch_log, "w") as f:
                json.dump(ops.launch_log, f)
        legs["total"] = time.time() - T0
        res["legs_s"] = {k: round(v, 2) for k, v in legs.items()}
        res["legs_dropped"] = dropped
        res["budget_s"] = args.budget_s
        print(json.dumps(res))
        sys.stdout.flush()

    # ---- `auto` at N > 1: the latency leg, LAST and guarded.  Everything the replicas line needs is complete at this point.  If the
    # leg raises on this rank, or is still running after --latency-budget-s (a rank died, a collective never returns: nothing of this
    # path has run over RCCL / xGMI before the driver's first multi-GPU node), rank 0 prints the line with `latency_mode.error` and
    # every rank leaves with exit code 0 through os._exit: no further collective, no destroy_process_group that could hang as well.
    lat_error = None
    if world > 1 and auto:
        import threading
        emit_lock = threading.Lock()
        state = {"done": False}

        def bail(msg):
            with emit_lock:
                if state["done"]:
                    return
                state["done"] = True
                if rank == 0:
                    try:
                        emit(msg)
                    finally:
                        sys.stdout.flush()
                os._exit(0)

        wd = threading.Timer(args.latency_budget_s, bail, args=(f"no result after --latency-budget-s = {args.latency_budget_s:.0f} s "
                                                                 "(a rank left the leg or a collective did not return)",))
        wd.daemon = True
        wd.start()
        try:
            with leg("latency_mode"):
                dt_lat, comm, rel_unsharded = latency_leg()
        except Exception as e:   # noqa: BLE001 -- the secondary leg must not take the replicas line down
            lat_error = f"{type(e).__name__}: {e}"[:500]
        wd.cancel()
        if lat_error is not None:
            bail(lat_error)      # does not return
        with emit_lock:          # (a watchdog that fired between the leg's end and cancel() owns the line)
            state["done"] = True

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    emit(None)
    if world > 1:
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
