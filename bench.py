#!/usr/bin/env python3
"""Headline benchmark: faces/sec of embedding extraction, MobileNet-192, batch 256, fp32 --
BASELINE.json configs[1] -- on N MI355X GPUs of one node (one process per GPU).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (conv1 -> 13 x (depthwise + pointwise) -> GAP) over one
batch of 256 synthetic preprocessed images already resident in HBM.  Prints ONE JSON line on
rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel class by device time, priced against its bound
                (fp32-MFMA 157.3 TFLOP/s for the pointwise GEMM, HBM 8 TB/s for the others),
                durations from HIP events recorded on the forward's own stream during the timed
                steps; `kernels` lists every class, `roofline_depthwise` is the class the
                north-star target (>= 60 % of HBM roofline) is stated against.
  cpu_baseline  the reference's batch-1 extract loop (facerec_test.py:394) on the host cores:
                the same frozen graph executed op-by-op by torch-CPU/oneDNN (oracle/torch_cpu.py),
                kind "port" -- TensorFlow itself cannot be installed on this image.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = 157.3 TFLOP/s
MFMA_F16_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA ~2.5 PFLOP/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--no-op-events", action="store_true", help="skip the instrumented second pass (no roofline objects)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true", help="also print per-layer times to stderr")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from hse_facerec_tf_amd import lowering
    from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, TensorFlowInference

    # tuning/debug knobs (never needed for correct results), e.g. HSEFR_DEBUG="dw_variant=3,pw_tile=1"
    if os.environ.get("HSEFR_DEBUG"):
        from hse_facerec_tf_amd import _lib
        for kv in os.environ["HSEFR_DEBUG"].split(","):
            k, v = kv.split("=")
            _lib.check(_lib.lib().hsefr_debug_set(k.strip().encode(), int(v)), "hsefr_debug_set")

    B, S = args.batch, args.size
    tfi = TensorFlowInference(AGE_GENDER_PB, input_tensor="input_1:0", output_tensor="global_pooling/Mean:0",
                              convert2BGR=True, imageNetUtilsMean=True, input_size=(S, S), max_batch=B, device=local_rank)
    eng, plan = tfi.engine, tfi.plan
    # synthetic preprocessed batch: U(-128,128) fp32 NHWC, seed 123 (+rank)
    x_host = np.random.RandomState(123 + rank).uniform(-128, 128, (B, S, S, 3)).astype(np.float32)
    x = torch.from_numpy(x_host).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        return eng.forward(x)["features"]

    for _ in range(args.warmup):
        out = step()
    # ---- timed region: EXACTLY `steps` forwards, barrier + synchronize on both sides -----------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(out).all())

    # ---- the same `steps` forwards again with HIP events around every launch, recorded on the
    # forward's own stream into a ring (no sync inside the region).  Kept out of the region above
    # because the event packets cost ~4 % of a step; `ms_per_step_instrumented` reports that run.
    per_op = None
    instrumented_ms = None
    use_events = not args.no_op_events
    if use_events:
        eng.set_profiling(args.steps)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        instrumented_ms = (time.perf_counter() - t1) / args.steps * 1e3
        per_op = np.zeros(len(plan.layers))
        for s in range(args.steps):
            per_op += np.asarray(eng.op_times_ms(s))
        per_op /= args.steps
        eng.set_profiling(0)
        if args.layers and rank == 0:
            for i, L in enumerate(plan.layers):
                nb = 4 * B * (int(np.prod(L.in_shape)) + int(np.prod(L.out_shape)))
                oh, ow, co = L.out_shape
                fl = plan.flops_per_image([L.kind]) * B if L.kind in (lowering.OP_STEM_F16S, lowering.OP_STEM2_F16S, lowering.OP_DWPW_F32, lowering.OP_DWPW_F16S) else \
                    2 * oh * ow * co * L.kh * L.kw * (L.in_shape[2] if L.kind != lowering.OP_DWCONV3X3 else 1) * B
                print("layer %2d kind %d %-34s in %-16s out %-16s s%d  %8.2f us  %7.1f GB/s  %6.1f TF" %
                      (i, L.kind, L.name[:34], L.in_shape, L.out_shape, L.stride, per_op[i] * 1e3,
                       nb / (per_op[i] * 1e-3) / 1e9 if per_op[i] > 0 else 0, fl / (per_op[i] * 1e-3) / 1e12 if per_op[i] > 0 else 0),
                      file=sys.stderr)

    # the exchange of config 5 (one all-gather of the embeddings), outside the timed region
    allgather_ms = None
    if world > 1:
        full = torch.empty((world * B, out.shape[1]), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(full, out)
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            dist.all_gather_into_tensor(full, out)
        barrier()
        allgather_ms = (time.perf_counter() - t1) / 10 * 1e3
        assert torch.equal(full[rank * B:(rank + 1) * B], out)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    classes = {
        "stem_conv1_dw_pw_dw_fused": lambda L: L.kind == lowering.OP_STEM2_F16S,
        "stem_conv1_dw_pw_fused": lambda L: L.kind == lowering.OP_STEM_F16S,
        "conv1_3x3x3_s2": lambda L: L.kind == lowering.OP_CONV_C3,
        "depthwise3x3": lambda L: L.kind == lowering.OP_DWCONV3X3,
        "fused_dw3x3_pw1x1": lambda L: L.kind == lowering.OP_DWPW_F32,
        "fused_dw3x3_pw1x1_f16split": lambda L: L.kind == lowering.OP_DWPW_F16S,
        "pointwise1x1_f32mfma": lambda L: L.kind == lowering.OP_PWCONV_F32 and L.a_log2 == 0,
        "pointwise1x1_f16split": lambda L: L.kind == lowering.OP_PWCONV_F32 and L.a_log2 > 0,
        "gap": lambda L: L.kind == lowering.OP_GAP,
    }
    # HBM traffic per launch measured with rocprofv3 PMC counters (separate --pmc FETCH_SIZE / WRITE_SIZE
    # passes, FETCH_SIZE doubled per the gfx950 correction) and committed under profiles/ -- counters cannot
    # be read from inside this process, so the latest committed profile is what is reported.
    traffic_by_class, mfma_by_class, traffic_src = {}, {}, None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        if cands:
            traffic_src = os.path.relpath(cands[-1], ROOT)
            prof = json.load(open(cands[-1]))["kernels"]
            prefixes = {"stem_conv1_dw_pw_dw_fused": "stem2_fused_kernel", "stem_conv1_dw_pw_fused": "stem_fused_kernel", "conv1_3x3x3_s2": "conv3x3_c3", "depthwise3x3": "dwconv3x3_kernel",
                        "pointwise1x1_f32mfma": "pwconv_f32_", "pointwise1x1_f16split": "pwconv_f16s_kernel", "gap": "hsefr::gap_kernel",
                        "fused_dw3x3_pw1x1": "dwpw_fused_kernel", "fused_dw3x3_pw1x1_f16split": ("dwpw3_f16s_kernel", "dwpw2_f16s_kernel", "dwpw_f16s_kernel")}
            for cls, pre in prefixes.items():
                rows = [v for k, v in prof.items() if k.startswith(pre)]      # str.startswith takes a tuple of prefixes too
                n = sum(r["launches"] for r in rows)
                if n:
                    traffic_by_class[cls] = sum(r["launches"] * r["hbm_bytes_per_launch"] for r in rows) / n
                    # share of the matrix pipes' cycles in use, time-weighted over the class (PMC: SQ_VALU_MFMA_BUSY_CYCLES /
                    # (1024 SIMDs x kernel cycles); tools/make_profile_summary.py)
                    tw = sum(r["launches"] * r["avg_us"] for r in rows if "mfma_util" in r)
                    if tw:
                        mfma_by_class[cls] = sum(r["launches"] * r["avg_us"] * r["mfma_util"] for r in rows if "mfma_util" in r) / tw
    except Exception:
        traffic_by_class, mfma_by_class = {}, {}
    kernels = []
    if per_op is not None:
        for name, pred in classes.items():
            idx = [i for i, L in enumerate(plan.layers) if pred(L)]
            if not idx:
                continue
            ms = float(per_op[idx].sum())
            nbytes = sum(4 * (int(np.prod(plan.layers[i].in_shape)) + int(np.prod(plan.layers[i].out_shape))) for i in idx) * B \
                + sum(4 * sum(a.size for a in (plan.layers[i].w, plan.layers[i].scale, plan.layers[i].shift, plan.layers[i].w2,
                                               plan.layers[i].shift2, plan.layers[i].w0, plan.layers[i].shift0, plan.layers[i].w3,
                                               plan.layers[i].scale3, plan.layers[i].shift3) if a is not None) for i in idx)
            flops = sum(plan.flops_per_image([plan.layers[i].kind]) for i in idx[:1]) * B
            # a fused block also saves writing + re-reading the depthwise result: report both byte counts
            unfused_extra = sum(2 * 4 * int(np.prod(plan.layers[i].out_shape[:2])) * plan.layers[i].in_shape[2] for i in idx
                                if plan.layers[i].kind in (lowering.OP_DWPW_F32, lowering.OP_DWPW_F16S)) * B \
                + sum(2 * 2 * 4 * int(np.prod(plan.layers[i].out_shape[:2])) * 32 for i in idx
                      if plan.layers[i].kind == lowering.OP_STEM_F16S) * B \
                + sum(2 * 4 * ((plan.layers[i].in_shape[0] + 1) // 2) * ((plan.layers[i].in_shape[1] + 1) // 2) * (32 + 32 + 64) for i in idx
                      if plan.layers[i].kind == lowering.OP_STEM2_F16S) * B
            launches = len(idx)
            note = None
            if name == "pointwise1x1_f32mfma":
                bound, achieved, peak, unit = "mfma", flops / (ms * 1e-3) / 1e12, MFMA_F32_PEAK_TF, "TFLOP/s"
            elif name == "pointwise1x1_f16split":
                # every fp32 product costs 3 f16 MFMA products, so the matrix roofline for ALGORITHMIC flops is the dense
                # f16 peak / 3; the class is priced against whichever of the two rooflines is the higher floor
                t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_F16_PEAK_TF / 3 * 1e12)
                if t_mfma > t_hbm:
                    bound, achieved, peak, unit = "mfma", flops / (ms * 1e-3) / 1e12, round(MFMA_F16_PEAK_TF / 3, 1), "TFLOP/s"
                else:
                    bound, achieved, peak, unit = "hbm", nbytes / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
                note = ("split-f16 products: floors for this class are %.0f us (HBM, algorithmic bytes at 8 TB/s) and %.0f us "
                        "(3 f16 MFMA products per fp32 product at the %.0f TFLOP/s dense f16 peak)" % (t_hbm * 1e6, t_mfma * 1e6, MFMA_F16_PEAK_TF))
            else:
                bound, achieved, peak, unit = "hbm", nbytes / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            kernels.append({"kernel": name, "launches_per_step": launches, "ms_per_step": round(ms, 4),
                            "avg_launch_us": round(ms / launches * 1e3, 2), "bound": bound,
                            "achieved": round(achieved, 2), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
                            "algorithmic_bytes_per_step": int(nbytes), "flops_per_step": int(flops),
                            "hbm_gbs_algorithmic": round(nbytes / (ms * 1e-3) / 1e9, 1),
                            "algorithmic_bytes_per_launch": int(nbytes / launches),
                            "unfused_equivalent_gbs": round((nbytes + unfused_extra) / (ms * 1e-3) / 1e9, 1) if unfused_extra else None,
                            "traffic": None if name not in traffic_by_class else int(traffic_by_class[name]),
                            "traffic_unit": "HBM bytes per launch (class average)", "traffic_source": traffic_src,
                            "mfma_util_pmc": None if not mfma_by_class.get(name) else round(mfma_by_class[name], 4),
                            "note": note})
    dominant = max(kernels, key=lambda k: k["ms_per_step"]) if kernels else None
    dw = next((k for k in kernels if k["kernel"] == "depthwise3x3"), None)

    def roof(k):
        if k is None:
            return None
        return {"kernel": k["kernel"], "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"],
                "unit": k["unit"], "frac": k["frac"], "traffic": k["traffic"], "avg_launch_us": k["avg_launch_us"],
                "launches_per_step": k["launches_per_step"], "mfma_util_pmc": k.get("mfma_util_pmc")}

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle.torch_cpu import time_reference_loop
        # threads actually used: the cores this process may run on, capped at 32 (batch-1 convolutions
        # of this size stop scaling long before that; oversubscribing a cgroup-limited box is far slower)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, 32))
        fps, n_img = time_reference_loop(AGE_GENDER_PB, "global_pooling/Mean:0", x_host[:32], cores,
                                         budget_s=args.cpu_baseline_seconds, batch=1)
        cpu_baseline = {"value": round(fps, 2), "unit": "faces/s", "cores": cores, "kind": "port",
                        "sample": "%d images of the same synthetic %dx%dx3 batch through the reference's batch-1 loop "
                                  "(facerec_test.py:394): same frozen graph, op-by-op fp32 on torch-CPU/oneDNN; not TensorFlow"
                                  % (n_img, S, S)}

    line = {
        "metric": "faces/sec embedding-extract (MobileNet-192, bs=256)",
        "value": round(value, 1), "unit": "faces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "MobileNet-v1 192x192x3 embeddings (1024-D), batch %d per GPU, fp32 -- BASELINE configs[1]" % B,
                   "global_batch": B * world, "input": [S, S, 3],
                   "arithmetic": "fp32 activations, weights and accumulators; conv1 / depthwise in fp32 FMA; pointwise products "
                                 + ("as a two-term f16 split of both operands on the f16 MFMA (3 products per fp32 product, "
                                    "error <= 3*2^-22 per product: fp32-grade, 1e-6 end to end vs the fp64 oracle)"
                                    if any(L.a_log2 for L in plan.layers) else "on the fp32 MFMA"),
                   "weights": "trunk of age_gender_tf2_new-01-0.14-0.92_quantized.pb (the reference's only shipped graph)",
                   "parallelism": "%d independent replicas, gallery sharded by image, one all-gather of embeddings" % world,
                   "op_events": "second pass of the same %d steps, HIP events on the forward stream" % args.steps if use_events else None},
        "ms_per_step_instrumented": None if instrumented_ms is None else round(instrumented_ms, 4),
        "roofline": roof(dominant), "roofline_depthwise": roof(dw), "kernels": kernels,
        "cpu_baseline": cpu_baseline,
        "allgather_ms": None if allgather_ms is None else round(allgather_ms, 4),
        "device_bytes": eng.device_bytes,
    }
    print(json.dumps(line))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
