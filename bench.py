#!/usr/bin/env python3
"""Headline benchmark: faces/sec of embedding extraction, MobileNet-192, batch 256, fp32 --
BASELINE.json configs[1] -- on N MI355X GPUs of one node (one process per GPU).

    python bench.py [--gpus N --steps K --warmup W]          # N > 1: starts its own N ranks (before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or pre-launched

A "step" is one pass of the hot path (conv1 -> 13 x (depthwise + pointwise) -> GAP) over one
batch of 256 synthetic preprocessed images already resident in HBM.  Prints ONE JSON line on
rank 0 (contract in the task statement) with these extra objects:
  roofline       the dominant kernel class by device time, priced against its bound; durations from HIP
                 events recorded on the forward's own stream over the same K steps; `kernels` lists every
                 class; `roofline_depthwise` is the class the north-star target (>= 60 % of the HBM roofline)
                 is stated against, split into the standalone layers and the ones inside fused kernels.
  cpu_baseline   the reference's batch-1 extract loop (facerec_test.py:394) on the host cores: the same frozen
                 graph executed op-by-op by torch-CPU/oneDNN (oracle/torch_cpu.py), kind "port" -- TensorFlow
                 itself cannot be installed on this image; plus the CPU's best case (batch 256).
  config5        BASELINE configs[4] end to end: 9164 synthetic 250x250 photos of 1680 persons, sharded
                 S = ceil(N/P) per rank -> device preprocessing + extract -> ONE all-gather -> L2-normalise ->
                 stratified 50/50 split -> 1-NN (4582 x 4582 x 1024), wall time per phase
                 (facerec_test.py:377-432).
  sustained      the same forward for >= 1 s of warm-up and >= 2 s measured, per-100-step windows (min / max): the figure a
                 seconds-long job sees, beside the K-step `value`.
  latency_batch1 (N = 1) the reference's own published quantities (AgeGenderIdentityDemo.ipynb:109-125): per-call latency of
                 age_gender_fun(img) / extract_features(path), construct + first-call times, next to the notebook's numbers.
  pipeline       (N = 1) the callers' view: H2D-inclusive and file-inclusive faces/s, never `value`.
  other_configs  (N = 1) the other BASELINE configs measured in the same process: ResNet-50 batch 128 bf16,
                 age/gender MobileNet-224 batch 512 with three outputs, MobileNet-192 with strict-fp32 pointwise
                 products.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = 157.3 TFLOP/s
MFMA_F16_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA ~2.5 PFLOP/s


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--no-op-events", action="store_true", help="skip the instrumented second pass (no roofline objects)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=6.0,
                    help="budget of EACH of the four CPU legs (fused / op-by-op x batch 1 / batch 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the H2D-inclusive / file-inclusive measurements")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency / first-call leg")
    ap.add_argument("--no-sustained", action="store_true", help="skip the seconds-long sustained-rate leg")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="measured seconds of the sustained leg (after its warm-up)")
    ap.add_argument("--sustained-warmup-s", type=float, default=1.0)
    ap.add_argument("--pipeline-files", type=int, default=8192)
    ap.add_argument("--config5-images", type=int, default=9164)
    ap.add_argument("--config5-classes", type=int, default=1680)
    ap.add_argument("--layers", action="store_true", help="also print per-layer times to stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher + process-group plumbing only (no GPU, no measurement): every rank joins the group, one "
                         "all-gather of rank ids, rank 0 prints {\"dry_run\": true, ...}; used by the CPU test of the self-launch path")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo lets several ranks share one GPU (launcher/plumbing tests only)")
    ap.add_argument("--force-group", action="store_true",
                    help="create the process group and run every collective of the N > 1 path even at N = 1 (a world-1 RCCL "
                         "communicator: ncclAllGather over one rank), so the code the 8-GPU run executes runs on a 1-GPU box")
    return ap.parse_args(argv)


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` typed as is: start N ranks with torch.distributed.run as a CHILD process.  Nothing in
    this (parent) process has touched the GPU -- not even torch.cuda.is_available() -- and it never exec()s."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    env["HSEFR_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    # the ranks inherit stdout / stderr: a rank that dies prints its own traceback there, torchrun adds its failure table and
    # exits non-zero, and that code is this process's exit code (never 0 with a dead rank)
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        print("bench.py: the launched job failed (exit code %d); the failing rank's traceback is above" % rc, file=sys.stderr)
    return rc if 0 <= rc < 256 else 1


def csrc_hash() -> str:
    """sha256 over the kernel sources: lets the line say whether the committed PMC profile is of THIS code."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hse_facerec_tf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------------------------------------------
# synthetic LFW-shaped photos for config 5, generated on the device (data generation only: torch ops, not the path)
# ----------------------------------------------------------------------------------------------------------------
def synth_photos_u8(idx, labels, hw=250, grid=10, class_w=0.8, inst_w=0.2, noise_amp=16.0):
    """uint8 RGB [n, hw, hw, 3]: a smooth colour pattern per PERSON + a smooth pattern per PHOTO + pixel noise.
    Every pixel is a pure function of (global photo index, label), so any sharding generates the same gallery."""
    import torch
    import torch.nn.functional as F
    dev = idx.device
    n = idx.numel()
    A, C = 6364136223846793005, 1442695040888963407

    def smooth(keys, salt):
        cell = torch.arange(grid * grid * 3, device=dev, dtype=torch.int64)
        h = (keys[:, None] * 1000003 + cell[None, :] * 7919 + salt) * A + C
        h = (h ^ (h >> 29)) * A + C
        p = ((h >> 33) & 255).to(torch.float32).reshape(n, grid, grid, 3).permute(0, 3, 1, 2)
        return F.interpolate(p, size=(hw, hw), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)

    pix = torch.arange(hw * hw * 3, device=dev, dtype=torch.int64)
    hn = (idx[:, None] * 2654435761 + pix[None, :] * 40503 + 977) * A + C
    hn = (hn ^ (hn >> 31)) * A + C
    noise = (((hn >> 35) & 255).to(torch.float32) - 127.5) * (noise_amp / 127.5)
    img = class_w * smooth(labels, 12345) + inst_w * smooth(idx, 54321) + noise.reshape(n, hw, hw, 3)
    return img.clamp_(0, 255).round_().to(torch.uint8).contiguous()


def run_config5(args, tfi, dev, world, rank, backend, dist, grouped=False):
    """BASELINE configs[4]: shard -> extract -> ONE all-gather -> normalise -> 1-NN (facerec_test.py:377-432)."""
    import torch
    from hse_facerec_tf_amd import gallery, identification
    N, C = args.config5_images, args.config5_classes
    y = gallery.lfw_like_labels(N, C)                       # directory-walk order: identical on every rank
    lo, hi = gallery.shard_range(N, rank, world)
    y_dev = torch.from_numpy(y).to(dev)
    # this rank's photos, resident in HBM as decoded 250x250 RGB uint8 (LFW's size, facerec_test.py:82)
    photos = torch.empty((hi - lo, 250, 250, 3), dtype=torch.uint8, device=dev)
    for i in range(lo, hi, 128):
        j = min(i + 128, hi)
        ids = torch.arange(i, j, device=dev, dtype=torch.int64)
        photos[i - lo:j - lo] = synth_photos_u8(ids, y_dev[i:j])

    def extract(ids):                                       # ids: contiguous global indices of one batch
        return tfi.extract_images(photos[ids[0] - lo:ids[-1] + 1 - lo])

    B = args.batch
    warm = tfi.extract_images(photos[:min(B, hi - lo)])     # warm-up: tap tables, kernels ...
    if warm.shape[0] >= 4:                                  # ... and the identification stage's imports / first-call costs
        identification.one_nn_identification(warm[:warm.shape[0] // 2 * 2], np.arange(warm.shape[0] // 2 * 2) // 2)
    # (steady state: the [S, D] shard buffer and the gathered matrix come out of torch's caching allocator, not a fresh hipMalloc)
    warm_bufs = [torch.zeros((gallery.shard_size(N, world), tfi.feature_dim), device=dev), torch.empty((gallery.shard_size(N, world) * world, tfi.feature_dim), device=dev)]
    del warm_bufs
    timings = {}
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # the label-only host work (class filter + scikit-learn's stratified split, 6-9 ms) starts once this rank's batches are
    # enqueued and runs under the device's extraction; host_split_ms below is what is still left to wait for afterwards
    split_job = []
    X = gallery.extract_sharded(extract, list(range(N)), tfi.feature_dim, dev, batch=B, timings=timings,
                                on_issued=lambda: split_job.append(identification.start_split(y)))
    t_gathered = time.perf_counter()
    ident = {}
    res = identification.one_nn_identification(X, y, split=split_job[0], timings=ident)
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    # per-rank phase times -> rank 0 (host-side bookkeeping, after the timed pipeline)
    mine = torch.tensor([timings["extract_s"], timings["allgather_s"], ident["normalize_s"], ident["host_split_s"],
                         ident["select_s"], ident["nn1_s"], t_end - t0, float(hi - lo), ident["readback_s"],
                         (t_gathered - t0) - timings["extract_s"] - timings["allgather_s"]], dtype=torch.float64)
    if grouped:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if backend == "nccl":
            allr_d = [t.to(dev) for t in allr]
            dist.all_gather(allr_d, mine.to(dev))
            allr = [t.cpu() for t in allr_d]
        else:
            dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).numpy()
    else:
        per_rank = mine.numpy()[None]
    if rank != 0:
        return None
    # ---- verification, outside the timed pipeline: fp64 brute force + scikit-learn's own classifier (:422) on the
    # gathered embeddings; the shard of rank 0 inside the gathered matrix must be its local result bit for bit
    Xh = X.cpu().numpy().astype(np.float64)
    Xn = Xh / np.maximum(np.linalg.norm(Xh, axis=1, keepdims=True), 1e-300)
    tr, te = res["train"], res["test"]
    d2 = 2.0 - 2.0 * (Xn[res["indices"]][te] @ Xn[res["indices"]][tr].T)
    best = d2.min(axis=1)
    chosen = d2[np.arange(len(te)), res["nn_index"]]
    near_ties = int((chosen > best + 1e-6).sum())              # picks that are not a nearest row within fp32 round-off
    mismatch = int((res["nn_index"] != d2.argmin(axis=1)).sum())
    acc64 = float((res["y"][tr][d2.argmin(axis=1)] == res["y"][te]).mean())
    acc_sk = None
    try:
        from sklearn.neighbors import KNeighborsClassifier
        from sklearn.preprocessing import normalize
        Xs = normalize(X.cpu().numpy(), norm="l2")[res["indices"]]
        clf = KNeighborsClassifier(n_neighbors=1, p=2).fit(Xs[tr], res["y"][tr])
        acc_sk = float((clf.predict(Xs[te]) == res["y"][te]).mean())
    except ImportError:
        pass
    local_again = torch.cat([extract(list(range(i, min(i + B, hi)))) for i in range(lo, hi, B)])
    shard_ok = bool(torch.equal(local_again, X[lo:hi]))
    # N > 1: the gathered matrix against a SINGLE-RANK extraction of the whole gallery, row for row (rank 0 regenerates every
    # photo -- a pure function of its index -- and extracts them in its own batches: other batch boundaries than the shards',
    # pad rows of the last shard cut off by the [:N] slice); features do not depend on batch composition, so equality is bitwise
    rows_differ = None
    if world > 1:
        rows_differ = 0
        for i in range(0, N, B):
            j = min(i + B, N)
            ids = torch.arange(i, j, device=dev, dtype=torch.int64)
            f = tfi.extract_images(synth_photos_u8(ids, y_dev[i:j]))
            rows_differ += int((f != X[i:j]).any(dim=1).sum())
    ext = per_rank[:, 0]
    nq, ng, d = ident["nn1_shape"]
    return {
        "workload": "BASELINE configs[4]: %d synthetic 250x250 RGB photos of %d persons (LFW totals after the >1-photo filter, "
                    "long-tailed class sizes), device-resident uint8 -> PIL-bilinear resize + BGR + mean on the GPU -> MobileNet-192 "
                    "-> ONE all-gather -> L2-normalise -> StratifiedShuffleSplit(0.5, seed 0) -> 1-NN" % (N, C),
        "n_gpus": world, "shard_rows": gallery.shard_size(N, world), "pad_rows": gallery.shard_size(N, world) * world - N,
        "extract_ms_per_rank": [round(float(v) * 1e3, 3) for v in ext],
        "extract_faces_per_s_per_rank": [round(float(per_rank[r, 7] / ext[r]), 1) for r in range(world)],
        "extract_faces_per_s": round(float(N / ext.max()), 1),
        "allgather_ms": round(float(per_rank[:, 1].max()) * 1e3, 4),
        "allgather_bytes_per_rank": timings.get("allgather_bytes_per_rank", 0),
        "normalize_ms": round(float(per_rank[0, 2]) * 1e3, 4),
        "host_split_ms": round(float(per_rank[0, 3]) * 1e3, 3),
        "select_ms": round(float(per_rank[0, 4]) * 1e3, 4),
        "nn1_ms": round(float(per_rank[0, 5]) * 1e3, 4), "nn1_shape": [nq, ng, d],
        "nn1_tflops": round(2.0 * nq * ng * d / float(per_rank[0, 5]) / 1e12, 2),
        "readback_ms": round(float(per_rank[0, 8]) * 1e3, 3),
        "shard_bookkeeping_ms": round(float(per_rank[0, 9]) * 1e3, 3),
        "total_ms": round(float(per_rank[:, 6].max()) * 1e3, 3),
        "unaccounted_ms": round(float(per_rank[0, 6] - per_rank[0, :6].sum() - per_rank[0, 8] - per_rank[0, 9]) * 1e3, 3),
        "phases_note": "total = extract + allgather + normalize + host_split (what is still left to WAIT for of the class filter + "
                       "scikit-learn StratifiedShuffleSplit: it runs in a host thread started when the last batch has been enqueued, "
                       "under the device's extraction -- labels need no features) + select (index gathers on the device) + nn1 + readback (indices/distances to the host, label "
                       "comparison) + shard_bookkeeping (shard buffer, Python) + unaccounted",
        "accuracy": res["accuracy"], "accuracy_fp64_bruteforce": acc64, "accuracy_sklearn": acc_sk,
        "nn_index_mismatches_vs_fp64": mismatch, "picks_not_nearest_within_1e-6": near_ties,
        "num_classes": res["num_classes"], "gathered_shard_equals_local": shard_ok,
        "gathered_rows_differing_from_single_rank_extraction": rows_differ, "gathered_rows": int(X.shape[0]),
    }


# ----------------------------------------------------------------------------------------------------------------
# the callers' view ("H2D-inclusive and file-inclusive numbers reported separately"), N = 1
# ----------------------------------------------------------------------------------------------------------------
def run_pipeline(args, tfi, dev):
    """h2d_inclusive: decoded uint8 250x250 photos in pinned HOST memory -> copy stream -> device preprocessing + forward,
    double-buffered.  file_inclusive: JPEG files on disk -> TensorFlowInference.extract_files (decoder processes + the same
    pipeline): the loop of facerec_test.py:394 as a user would run it.  Neither is `value`."""
    import shutil
    import tempfile
    import torch
    from PIL import Image
    B = args.batch
    out = {}
    rs = np.random.RandomState(123)
    # ---- H2D-inclusive
    host = [torch.from_numpy(rs.randint(0, 256, (B, 250, 250, 3), dtype=np.uint8)).pin_memory() for _ in range(2)]
    copy = torch.cuda.Stream(device=dev)
    compute = torch.cuda.current_stream(dev)
    steps = max(10, min(args.steps, 50))

    def one(i):
        with torch.cuda.stream(copy):
            d = host[i & 1].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
        compute.wait_event(ev)
        d.record_stream(compute)
        return tfi.extract_images(d)
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        f = one(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out["h2d_inclusive"] = {"value": round(B / dt, 1), "unit": "faces/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps,
                            "what": "uint8 250x250x3 photos in pinned host memory (%.1f MB per batch) -> H2D on a copy stream, double-buffered, "
                                    "-> PIL-bilinear resize + BGR + mean + MobileNet-192 on the compute stream" % (B * 250 * 250 * 3 / 1e6)}
    # ---- file-inclusive
    d = tempfile.mkdtemp(prefix="hsefr_bench_")
    try:
        distinct = 256
        for i in range(distinct):
            Image.fromarray(rs.randint(0, 256, (250, 250, 3), dtype=np.uint8)).save(os.path.join(d, "%04d.jpg" % i), quality=90)
        paths = [os.path.join(d, "%04d.jpg" % (i % distinct)) for i in range(args.pipeline_files)]
        from hse_facerec_tf_amd.decode_pool import DecodePool, cpu_quota, default_workers
        workers = default_workers()

        def decode_rates(nw, files, passes=2):     # decode only: the pool's workers writing into their staging slots, no GPU
            """WARM figures (the first version timed one pass right after the spawns, with half the workers
            never having opened a JPEG, and came out 2.7x BELOW the pipeline it was meant to bound): one untimed pass over the
            whole list -- every worker has imported its decoder, every staging page has been touched, every file is in the page
            cache -- then `passes` timed passes of the same chunks the pipeline uses."""
            pool = DecodePool(nw, slot_bytes=max(8 << 20, B * (256 << 10)), slots=3, pin=True)   # this job owns the host's quota
            try:
                chunks = [files[i:i + B] for i in range(0, len(files), B)]

                def one_pass():
                    t0 = time.perf_counter()
                    for ci in range(min(2, len(chunks))):
                        pool.submit(ci, chunks[ci], ci % 3)
                    for ci in range(len(chunks)):
                        pool.collect(ci)
                        if ci + 2 < len(chunks):
                            pool.submit(ci + 2, chunks[ci + 2], (ci + 2) % 3)
                    return len(files) / (time.perf_counter() - t0)
                one_pass()
                return [one_pass() for _ in range(passes)]
            finally:
                pool.close()
        dec_all = decode_rates(workers, paths)
        # worker-count sweep: where the decoders stop scaling on THIS host -- on the round-5 boxes at the container's
        # CPU quota (16 CPUs of the 256 listed: 32 workers decode what 16 do), not in the pool
        sweep = {}
        for nw in (1, 4, 8, 16, 32):
            if nw == workers:
                sweep[nw] = max(dec_all)
            elif nw < 2 * workers or nw <= 4:
                sweep[nw] = decode_rates(nw, paths[:min(len(paths), max(2 * B, B * nw))], passes=1)[0]
        dec_one = sweep.get(1) or decode_rates(1, paths[:max(B, len(paths) // 16)], passes=1)[0]
        tfi.extract_files(paths[:2 * B], batch=B)          # warm-up: starts the extractor's own decoder processes
        runs = []
        for _ in range(2):                                 # the host side is noisy (32 decoder processes beside this one): best of two, both reported
            st = {}
            X = tfi.extract_files(paths, batch=B, stats=st)
            assert X.shape == (len(paths), tfi.feature_dim) and bool(np.isfinite(X).all())
            runs.append(st)
        st = min(runs, key=lambda r: r["seconds"])
        out["file_inclusive"] = {"value": round(len(paths) / st["seconds"], 1), "unit": "faces/s", "files": len(paths), "workers": st["workers"],
                                 "pinned_staging": st.get("pinned_staging"), "runs_faces_per_s": [round(len(paths) / r["seconds"], 1) for r in runs],
                                 "host_decode_faces_per_s": round(max(dec_all), 1),
                                 "host_decode_runs_faces_per_s": [round(v, 1) for v in dec_all],
                                 "host_decode_faces_per_s_per_worker": round(max(dec_all) / workers, 1),
                                 "host_decode_faces_per_s_one_worker_alone": round(dec_one, 1),
                                 "host_decode_worker_sweep": {str(k): {"faces_per_s": round(v, 1), "per_worker": round(v / k, 1)} for k, v in sorted(sweep.items())},
                                 "host_cpus": {"affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                                               "cgroup_quota_cpus": cpu_quota()},
                                 "fraction_of_host_decode": round(len(paths) / st["seconds"] / max(dec_all), 3),
                                 "what": "%d JPEG files (250x250, quality 90; %d distinct) -> TensorFlowInference.extract_files: %d decoder "
                                         "PROCESSES (PIL) writing into shared page-locked staging, upload on a copy stream, device "
                                         "preprocessing + forward.  host_decode_*: the same pool, same chunks, WITHOUT the GPU side, warm (one untimed pass, then "
                                         "the passes listed) -- the ceiling the decoders set; fraction_of_host_decode = this pipeline's best pass over it"
                                         % (len(paths), distinct, st["workers"])}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


# ----------------------------------------------------------------------------------------------------------------
# the reference's OWN published quantities: batch-1 latency and first-call time, N = 1
# ----------------------------------------------------------------------------------------------------------------
def run_latency_batch1(args, dev):
    """What the reference's notebook prints (AgeGenderIdentityDemo.ipynb:109-125, unstated hardware, TF 1.x): 4.97 ms per
    face steady state and 763 ms first call for age_gender_fun(img); 2.62 s for the first MTCNN detection.  The same calls,
    end to end through the drop-in classes (host preprocessing / upload / forward / read-back included), one image per call."""
    import torch
    from hse_facerec_tf_amd import FacialImageProcessing, preprocess
    from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, TensorFlowInference
    photo = os.path.join(ROOT, "tests", "golden", "test_image.jpg")
    rgb = preprocess.imread_rgb(photo)
    face = np.ascontiguousarray(rgb[60:310, 250:500])                  # a 250x250 crop: what process_image hands to age_gender_fun
    reps = 100

    def med(fn):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return round(float(np.median(ts)) * 1e3, 4), round(float(np.percentile(ts, 95)) * 1e3, 4)
    out = {"reference_published": {"age_gender_fun_ms": 4.97, "age_gender_fun_first_call_ms": 763.0, "mtcnn_first_call_s": 2.62,
                                   "source": "AgeGenderIdentityDemo.ipynb:109-125 (TensorFlow 1.x, hardware unstated, another checkpoint "
                                             "of the same architecture)"},
           "note": "this process has already initialised the GPU and loaded libhsefr.so: construct / first-call times exclude HIP start-up"}
    # ---- age_gender_fun(img): facial_analysis.py:93-129
    t0 = time.perf_counter()
    fp = FacialImageProcessing(mtcnn_detector=False, max_batch=8, device=dev.index)
    t1 = time.perf_counter()
    fp.age_gender_fun(face)
    t2 = time.perf_counter()
    m, p95 = med(lambda: fp.age_gender_fun(face))
    out["age_gender_fun"] = {"median_ms": m, "p95_ms": p95, "construct_s": round(t1 - t0, 4), "first_call_ms": round((t2 - t1) * 1e3, 3),
                             "calls": reps, "what": "FacialImageProcessing.age_gender_fun(250x250 RGB uint8 crop) -> (age, gender, 1024 features) "
                                                    "on the host: upload + cv-resize + forward (3 outputs) + read-back + age decode",
                             "vs_reference_published": round(4.97 / m, 2),
                             "plan": "small-batch lowering (FacialImageProcessing's default for its per-face calls: latency_plan=True)"}
    fp.close()
    fpb = FacialImageProcessing(mtcnn_detector=False, max_batch=8, device=dev.index, latency_plan=False)
    fpb.age_gender_fun(face)
    mb_, _ = med(lambda: fpb.age_gender_fun(face))
    out["age_gender_fun"]["median_ms_bulk_plan"] = mb_        # the batch-256 plan on one face (what rounds 1-3 reported)
    fpb.close()
    # ---- TensorFlowInference.extract_features(path): facerec_test.py:114-122 (decode + PIL resize on the host, as the reference)
    t0 = time.perf_counter()
    tfi = TensorFlowInference(AGE_GENDER_PB, input_tensor="input_1:0", output_tensor="global_pooling/Mean:0", convert2BGR=True,
                              imageNetUtilsMean=True, input_size=(args.size, args.size), max_batch=8, device=dev.index)
    t1 = time.perf_counter()
    tfi.extract_features(photo)
    t2 = time.perf_counter()
    m, p95 = med(lambda: tfi.extract_features(photo))
    x1 = torch.from_numpy(np.ascontiguousarray(tfi.preprocess_image(photo, False)[None], dtype=np.float32)).to(dev)

    def fwd():
        tfi.engine.forward(x1)["features"].cpu()
    mf, _ = med(fwd)
    mh, _ = med(lambda: tfi.preprocess_image(photo, False))
    from hse_facerec_tf_amd import preprocess as _pre
    md, _ = med(lambda: _pre.imread_rgb(photo))
    out["extract_features"] = {"median_ms": m, "p95_ms": p95, "construct_s": round(t1 - t0, 4), "first_call_ms": round((t2 - t1) * 1e3, 3),
                               "host_decode_median_ms": md, "host_preprocess_image_median_ms": mh,
                               "upload_forward_readback_median_ms": mf, "calls": reps,
                               "plan": "the bulk plan (default: extract_files(paths) is [extract_features(p) for p in paths] bit for bit)",
                               "what": "TensorFlowInference.extract_features(%dx%d JPEG path): PIL decode on the host (host_decode), the decoded "
                                       "bytes uploaded, misc.imresize + float conversion + BGR + mean on the device (bit-exact / 2e-6), "
                                       "MobileNet-%d forward, read-back.  host_preprocess_image = the reference's whole preprocess_image on "
                                       "the host (decode + PIL resize + float64 BGR/mean), which this call no longer runs; "
                                       "upload_forward_readback = the fp32-input forward alone" % (rgb.shape[1], rgb.shape[0], args.size)}
    tfi.close_session()
    # the same call on an extractor built with latency_plan=True (small-batch lowering for the one-image calls; 3e-7 from the bulk plan)
    tfl = TensorFlowInference(AGE_GENDER_PB, input_tensor="input_1:0", output_tensor="global_pooling/Mean:0", convert2BGR=True,
                              imageNetUtilsMean=True, input_size=(args.size, args.size), max_batch=8, device=dev.index, latency_plan=True)
    tfl.extract_features(photo)
    ml, _ = med(lambda: tfl.extract_features(photo))

    def fwd_l():
        tfl.engine.forward(x1, latency=True)["features"].cpu()
    mfl, _ = med(fwd_l)
    out["extract_features"]["latency_plan"] = {"median_ms": ml, "upload_forward_readback_median_ms": mfl,
                                               "what": "TensorFlowInference(..., latency_plan=True): the small-batch lowering for one-image calls"}
    tfl.close_session()
    # ---- first MTCNN detection (ipynb:109)
    try:
        bgr = np.ascontiguousarray(rgb[..., ::-1])
        t0 = time.perf_counter()
        fp = FacialImageProcessing(mtcnn_detector=True, minsize=32, device=dev.index)
        t1 = time.perf_counter()
        r = fp.process_image(bgr)
        t2 = time.perf_counter()
        out["mtcnn_process_image"] = {"construct_s": round(t1 - t0, 4), "first_call_s": round(t2 - t1, 4), "faces": int(len(r[0])),
                                      "what": "FacialImageProcessing(mtcnn_detector=True) + the first process_image(784x588 frame)"}
        fp.close()
    except Exception as e:
        out["mtcnn_process_image"] = {"error": repr(e)}
    return out


# ----------------------------------------------------------------------------------------------------------------
# the other BASELINE configs, same process (N = 1)
# ----------------------------------------------------------------------------------------------------------------
def time_engine(eng, x, want, steps, warm):
    import torch
    for _ in range(warm):
        eng.forward(x, want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = eng.forward(x, want)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert all(bool(torch.isfinite(v).all()) for v in out.values())
    return dt


def profile_traffic_per_forward(tag, once_kernel=None):
    """HBM bytes per forward of a side config from its newest committed PMC profile (profiles/rNN_<tag>_traffic.json, written by
    tools/make_profile_summary.py from tools/gpu_pmc.sh over tools/bench_configs.py <config>: separate FETCH_SIZE / WRITE_SIZE passes,
    FETCH_SIZE doubled): sum over the kernels of launches x bytes per launch, divided by the number of forwards the profiled command
    ran (the profile's `forwards`, or the launch count of a once-per-forward kernel).  -> (bytes | None, source | None, stale | None)"""
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_traffic.json" % tag)))
        if not cands:
            return None, None, None
        pj = json.load(open(cands[-1]))
        ks = pj["kernels"]
        fwd = pj.get("forwards") or (max([v["launches"] for k, v in ks.items() if once_kernel and once_kernel in k] or [0]))
        if not fwd:
            return None, None, None
        return (int(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in ks.values()) / fwd), os.path.relpath(cands[-1], ROOT),
                pj.get("csrc_hash") != csrc_hash())
    except Exception:
        return None, None, None


_BF16_STORE = None
_SPLIT_F16 = ("pwconv_ps_kernel", "pwconv_f16s_kernel", "dwpw_f16s_kernel", "dwpw2_f16s_kernel", "dwpw3_f16s_kernel", "stem2_fused_kernel",
              "stem3_fused_kernel", "stem4_fused_kernel", "stem5_stream_kernel")          # fp32 products as three f16 MFMA products
_BF16_MATRIX = ("stem7s_stream_kernel",)          # bf16 MFMA kernels whose name does not say so
_F32_MATRIX = ("pwconv_f32_dma_kernel", "pwconv_f32_kernel", "conv_f32_mfma_kernel", "conv3x3_c3_mfma_kernel")


def side_kernel_table(plan, eng, x, want, B, steps, tag):
    """The per-KERNEL view of a side config, in the shape of the headline's `kernels` / `roofline`: one row per kernel
    instantiation as Plan.describe() names it (family<template arguments> -- the launchers' own routing), device time from the engine's
    HIP-event ring over `steps` forwards on the forward's stream, algorithmic bytes and flops of the layers it runs (a launch that covers
    several layers -- hsefr_op_flags -- counts them all; a tensor that never leaves the launch is not counted), the bound it is priced
    against (the higher of its HBM floor and its matrix floor: bf16 at the dense 2.5 PF, split-f16 at a third of it, fp32 MFMA at 157 TF),
    and the counter figures of the same kernel name in the newest committed profile of this config (profiles/rNN_<tag>_traffic.json).
    -> (rows sorted by time, the dominant row as a `roofline` object)"""
    import glob
    from hse_facerec_tf_amd import lowering
    eng.set_profiling(steps)
    for _ in range(steps):
        eng.forward(x, want)
    per = np.mean([eng.op_times_ms(s) for s in range(steps)], axis=0)
    eng.set_profiling(0)
    rows = plan.describe(B)
    prof, src, stale = {}, None, None
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_traffic.json" % tag)))
    if cands:
        pj = json.load(open(cands[-1]))
        prof, src, stale = pj.get("kernels", {}), os.path.relpath(cands[-1], ROOT), pj.get("csrc_hash") != csrc_hash()
    bf16_out = set(lowering._BF16_OUT)
    bf16_in = {lowering.OP_CONV_BF16, lowering.OP_MAXPOOL_BF16, lowering.OP_GAP_BF16}

    def lbytes(i, count_in=True):
        L = plan.layers[i]
        b = int(np.prod(L.out_shape)) * (2 if L.kind in bf16_out else 4) * B
        if count_in:
            b += int(np.prod(L.in_shape)) * (2 if L.kind in bf16_in else 4) * B
        if L.res >= 0:      # the residual, or the block input a projected shortcut reads (strided: every stride-th pixel of it)
            R = plan.layers[L.res]
            px = int(np.prod(L.out_shape[:2])) if (L.proj is not None or L.res_geom is not None) else int(np.prod(R.out_shape[:2]))
            b += px * (L.proj[0] if L.proj is not None else R.out_shape[2]) * 2 * B
        for a in (L.w, L.scale, L.shift, L.w2, L.shift2, L.w0, L.shift0, L.w3, L.scale3, L.shift3):
            if a is not None:
                b += a.size * (2 if a.dtype == np.uint16 else 4)
        return b
    table = {}
    for r in rows:
        if r["inside"] is not None:
            continue
        i = r["layer"]
        covered = [q["layer"] for q in rows if q["inside"] == i]
        key = " + ".join(r["kernels"])
        t = table.setdefault(key, {"kernel": key, "launches_per_step": 0, "ms": 0.0, "bytes": 0, "flops": 0, "layers": []})
        t["launches_per_step"] += 1
        t["ms"] += float(per[i])
        t["bytes"] += lbytes(i) + sum(lbytes(j, count_in=plan.layers[j].src not in [i] + covered) for j in covered)
        t["flops"] += sum(plan.layer_flops(plan.layers[j]) for j in [i] + covered) * B
        t["layers"].append(plan.layers[i].name)
    out = []
    for key, t in table.items():
        fam = key.split("<")[0]
        peak_tf = MFMA_F16_PEAK_TF / 3 if fam in _SPLIT_F16 else MFMA_F32_PEAK_TF if fam in _F32_MATRIX else MFMA_F16_PEAK_TF if "bf16" in fam or fam in _BF16_MATRIX else None
        ms = max(t["ms"], 1e-9)
        t_hbm = t["bytes"] / (HBM_PEAK_GBS * 1e9)
        t_mfma = t["flops"] / (peak_tf * 1e12) if peak_tf and t["flops"] else 0.0
        if t_mfma > t_hbm:
            bound, achieved, peak, unit = "mfma", t["flops"] / (ms * 1e-3) / 1e12, round(peak_tf, 1), "TFLOP/s"
        else:
            bound, achieved, peak, unit = "hbm", t["bytes"] / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        k0 = key.split(" + ")[0]
        pk = prof.get(k0) or prof.get("hsefr::" + k0) or {}      # (the profile keeps the namespace on names without template arguments)
        out.append({"kernel": key, "launches_per_step": t["launches_per_step"], "ms_per_step": round(t["ms"], 4),
                    "avg_launch_us": round(t["ms"] / t["launches_per_step"] * 1e3, 2), "bound": bound, "achieved": round(achieved, 2), "peak": peak,
                    "unit": unit, "frac": round(achieved / peak, 4), "algorithmic_bytes_per_launch": int(t["bytes"] / t["launches_per_step"]),
                    "flops_per_step": int(t["flops"]), "floors_us": {"hbm": round(t_hbm * 1e6, 1), "mfma": round(t_mfma * 1e6, 1)},
                    "traffic": None if "hbm_bytes_per_launch" not in pk else int(pk["hbm_bytes_per_launch"]),
                    "traffic_unit": "HBM bytes per launch (PMC counters, this kernel name in the profile)", "traffic_source": src, "traffic_stale": stale,
                    "mfma_util_pmc": pk.get("mfma_util"), "lds_conflict_share": pk.get("lds_conflict_share"),
                    "layers": t["layers"] if len(t["layers"]) <= 6 else t["layers"][:5] + ["... %d more" % (len(t["layers"]) - 5)]})
    out.sort(key=lambda k: -k["ms_per_step"])
    dom = out[0] if out else None
    roof = None if dom is None else {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_us",
                                                         "launches_per_step", "mfma_util_pmc")}
    return out, roof


def run_other_configs(args, dev):
    import torch
    from hse_facerec_tf_amd import lowering, resnet50
    from hse_facerec_tf_amd.engine import Engine
    from hse_facerec_tf_amd.graphdef import read_graph
    from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB
    out = []
    steps, warm = max(5, min(args.steps, 20)), 3
    rs = np.random.RandomState(123)

    def gen(n, s):
        g = torch.Generator(device=dev)
        g.manual_seed(123)
        return (torch.rand((n, s, s, 3), device=dev, generator=g) * 256.0 - 128.0).contiguous()

    # -- configs[2]: ResNet-50 (resnet50_ft topology, synthetic He weights: vgg2_resnet.pb is a missing blob), bf16 MFMA
    try:
        B = 128
        plan = resnet50.build_plan(resnet50.synthetic_weights(123), (224, 224), "caffe")
        eng = Engine(plan, max_batch=B, device=dev.index)
        xr = gen(B, 224)
        dt = time_engine(eng, xr, (0,), steps, warm)
        rn_kernels, rn_roof = side_kernel_table(plan, eng, xr, (0,), B, steps, "resnet50")
        del xr
        fl, by = resnet50.flops_per_image(plan), resnet50.activation_bytes_per_image(plan)
        wb = sum(int(np.asarray(L.w).size) * 2 for L in plan.layers if L.w is not None)
        t_hbm, t_mfma = (by * B + wb) / (HBM_PEAK_GBS * 1e9), fl * B / (MFMA_F16_PEAK_TF * 1e12)
        # measured HBM bytes per forward from the committed PMC profile of this config (tools/gpu_pmc.sh over
        # tools/bench_configs.py resnet50: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled): sum over its kernels
        # of launches x bytes per launch, divided by the number of forwards (= launches of the once-per-forward stem kernel)
        rn_traffic, rn_src, rn_stale = profile_traffic_per_forward("resnet50", "stem7")
        out.append({"config": "BASELINE configs[2]: ResNet-50 embeddings (2048-D), batch 128, 224x224x3, bf16 storage + bf16 MFMA, fp32 accumulate",
                    "value": round(B / dt, 1), "unit": "faces/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "dtype": "bf16",
                    "weights": "synthetic (seed 123)", "tflops": round(fl * B / dt / 1e12, 1),
                    "roofline": {"bound": "hbm", "achieved": round((by * B + wb) / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(t_hbm / dt, 4), "traffic": rn_traffic, "traffic_unit": "HBM bytes per forward (batch %d)" % B,
                                 "traffic_source": rn_src, "traffic_stale": rn_stale, "algorithmic_bytes_per_forward": int(by * B + wb),
                                 "note": "layer-wise HBM floor %.3f ms vs bf16 MFMA floor %.3f ms per batch: the net is HBM-bound unless layers are fused"
                                         % (t_hbm * 1e3, t_mfma * 1e3)},
                    "roofline_dominant_kernel": rn_roof, "kernels": rn_kernels})
        eng.close()
    except Exception as e:      # a failing side config must not take the headline line down
        out.append({"config": "BASELINE configs[2]: ResNet-50", "error": repr(e)})
    # -- configs[3]: age/gender multi-head MobileNet-224, batch 512, three outputs
    try:
        B = 512
        g = read_graph(AGE_GENDER_PB)
        fetch = {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}
        # (input_bound: what the drop-in declares -- FacialImageProcessing feeds pixels minus a mean, |x| < 256,
        # facial_analysis.py:95-107 -- and what the synthetic U(-128, 128) batch satisfies; without it the stem forms conv1's
        # products in exact fp32 and this config ran the older stem2 kernel: 0.91 instead of 0.78 ms of a 3.8 ms step)
        plan = lowering.lower_graph(g, "input_1:0", fetch, input_bound=256.0)
        eng = Engine(plan, max_batch=B, device=dev.index)
        xa = gen(B, 224)
        dt = time_engine(eng, xa, (0, 1, 2), steps, warm)
        ag_kernels, ag_roof = side_kernel_table(plan, eng, xa, (0, 1, 2), B, steps, "agegender")
        del xa
        by = 40.948e6 * B + 12.74e6          # unfused layer-wise bytes per face @224 + weights per batch
        ag_traffic, ag_src, ag_stale = profile_traffic_per_forward("agegender", "stem5_stream")
        out.append({"config": "BASELINE configs[3]: age_gender_tf2 MobileNet-224 multi-head (features + age softmax + gender sigmoid), batch 512, fp32",
                    "value": round(B / dt, 1), "unit": "faces/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "dtype": "f32",
                    "weights": "age_gender_tf2_new-01-0.14-0.92_quantized.pb",
                    "roofline": {"bound": "hbm", "achieved": round(by / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(by / dt / 1e9 / HBM_PEAK_GBS, 4), "traffic": ag_traffic,
                                 "traffic_unit": "HBM bytes per forward (batch %d), PMC counters" % B, "traffic_source": ag_src, "traffic_stale": ag_stale,
                                 "measured_hbm_gbs": None if not ag_traffic else round(ag_traffic / dt / 1e9, 1),
                                 "algorithmic_bytes_per_forward": int(by),
                                 "note": "achieved / frac: unfused layer-wise algorithmic bytes (40.948 MB/face, SURVEY 8d) / time; measured_hbm_gbs: "
                                         "counter bytes (the fused plan moves fewer) / time"},
                    "roofline_dominant_kernel": ag_roof, "kernels": ag_kernels})
        eng.close()
    except Exception as e:
        out.append({"config": "BASELINE configs[3]: age/gender MobileNet-224", "error": repr(e)})
    # -- configs[1] again with every pointwise product on the fp32 MFMA (no f16 split anywhere)
    try:
        B, S = args.batch, args.size
        g = read_graph(AGE_GENDER_PB)
        plan = lowering.lower_graph(g, "input_1:0", {0: "global_pooling/Mean:0"}, (S, S), pw_math="f32")
        eng = Engine(plan, max_batch=B, device=dev.index)
        dt = time_engine(eng, gen(B, S), (0,), steps, warm)
        fl = plan.flops_per_image([lowering.OP_PWCONV_F32])
        f32_traffic, f32_src, f32_stale = profile_traffic_per_forward("mobilenet_f32")
        out.append({"config": "BASELINE configs[1] with pw_math='f32': MobileNet-192 batch %d, every product on the fp32 pipes "
                              "(v_mfma_f32_32x32x2_f32 / fp32 FMA), no f16 split" % B,
                    "value": round(B / dt, 1), "unit": "faces/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "dtype": "f32",
                    "roofline": {"bound": "mfma", "achieved": round(plan.flops_per_image() * B / dt / 1e12, 1), "peak": MFMA_F32_PEAK_TF,
                                 "unit": "TFLOP/s", "frac": round(plan.flops_per_image() * B / dt / 1e12 / MFMA_F32_PEAK_TF, 4), "traffic": f32_traffic,
                                 "traffic_unit": "HBM bytes per forward (batch %d), PMC counters" % B, "traffic_source": f32_src, "traffic_stale": f32_stale,
                                 "note": "whole-net flops (%.0f %% pointwise) over the fp32-MFMA peak" % (100.0 * fl / plan.flops_per_image())}})
        eng.close()
    except Exception as e:
        out.append({"config": "BASELINE configs[1] pw_math=f32", "error": repr(e)})
    # -- configs[2] in the fp32-grade mode (exact-fp32 general kernels; the 1e-4 mode of a ResNet): a correctness mode, timed for the record
    try:
        B = 128
        plan = resnet50.build_plan(resnet50.synthetic_weights(123), (224, 224), "caffe", dtype="f32")
        eng = Engine(plan, max_batch=B, device=dev.index)
        dt = time_engine(eng, gen(B, 224), (0,), max(3, steps // 4), 1)
        fl = resnet50.flops_per_image(plan)
        rf_traffic, rf_src, rf_stale = profile_traffic_per_forward("resnet50_f32")
        if rf_traffic is not None and rf_src:      # (profiled at the batch tools/bench_configs.py resnet50f32 runs: scale the activations' share to this one)
            try:
                pb = json.load(open(os.path.join(ROOT, rf_src))).get("batch")
                rf_traffic = int(rf_traffic * B / pb) if pb else None
            except Exception:
                rf_traffic = None
        out.append({"config": "BASELINE configs[2] in the fp32-grade mode: ResNet-50 batch %d, every convolution an exact-fp32 implicit GEMM on the "
                              "fp32 matrix pipe (v_mfma_f32_32x32x2_f32; dtype='f32'; 1e-5 of the feature scale vs the fp64 oracle)" % B,
                    "value": round(B / dt, 1), "unit": "faces/s", "ms_per_step": round(dt * 1e3, 4), "steps": max(3, steps // 4), "dtype": "f32",
                    "roofline": {"bound": "mfma", "achieved": round(fl * B / dt / 1e12, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                 "frac": round(fl * B / dt / 1e12 / MFMA_F32_PEAK_TF, 4), "traffic": rf_traffic,
                                 "traffic_unit": "HBM bytes per forward (batch %d), PMC counters scaled from the profiled batch" % B,
                                 "traffic_source": rf_src, "traffic_stale": rf_stale,
                                 "note": "csrc/conv_f32_mfma.hip: the mode that meets the 1e-4 bar on a ResNet (the bf16 mode is at 5e-3)"}})
        eng.close()
    except Exception as e:
        out.append({"config": "BASELINE configs[2] fp32-grade mode", "error": repr(e)})
    # -- the detector in front of the hot loop: FacialImageProcessing.process_image on the reference's own test photo
    try:
        from hse_facerec_tf_amd import FacialImageProcessing, preprocess
        from hse_facerec_tf_amd.mtcnn import MTCNNDetector
        photo = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "test_image.jpg")
        bgr = np.ascontiguousarray(preprocess.imread_rgb(photo)[..., ::-1])
        rec = {"config": "MTCNN detection + alignment of one %dx%d photo (FacialImageProcessing.process_image, minsize 32), per call"
                         % (bgr.shape[1], bgr.shape[0]), "unit": "ms", "higher_is_better": False}
        for name, dres, dbox in (("device_cascade", True, True), ("device_pyramid_host_boxes", True, False), ("host_pyramid", False, False)):
            fp = FacialImageProcessing(mtcnn_detector=True, minsize=32, device=dev.index,
                                       detector=MTCNNDetector(minsize=32, device=dev.index, device_resize=dres, device_boxes=dbox))
            for _ in range(2):
                r = fp.process_image(bgr)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                r = fp.process_image(bgr)
            torch.cuda.synchronize(dev)
            rec[name + "_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
            rec["faces"] = int(len(r[0]))
            fp.close()
        rec["value"] = rec["device_cascade_ms"]
        rec["note"] = ("device_cascade (default): frame uploaded once as uint8; INTER_AREA pyramid levels and 24x24 / 48x48 crops "
                       "(csrc/area_resize.hip), candidate generation, NMS, box regression, squaring, crop windows and landmarks "
                       "(csrc/mtcnn_post.hip) all on the GPU, three box-count read-backs per frame; device_pyramid_host_boxes: the box "
                       "logic in NumPy (identical results); host_pyramid: round 1's NumPy resampling as well")
        out.append(rec)
    except Exception as e:
        out.append({"config": "MTCNN process_image", "error": repr(e)})
    del rs
    return out


# ----------------------------------------------------------------------------------------------------------------
def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv))
    # both launch paths (self-launched above, pre-launched by torchrun, or a plain single rank): this pool's host driver only
    # supports dmabuf IPC, RCCL fails with `hipIpcGetMemHandle: invalid argument` without it -- set before HIP initialises
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # this job owns the host: its decoder pools pin their workers (decode_pool.pin_offset: ranks take consecutive runs of cores through
    # LOCAL_RANK; a plain single rank starts at core 0) -- a DecodePool pins by default only when a launcher says where it is
    os.environ.setdefault("HSEFR_DECODE_CPU_OFFSET", "0")

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        from hse_facerec_tf_amd import gallery
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("HSEFR_BENCH_FAIL_RANK") == str(rank):       # test hook: a rank that dies must fail the whole job
            raise SystemExit("rank %d: injected failure (HSEFR_BENCH_FAIL_RANK)" % rank)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            got = gallery.all_gather_rows(torch.full((2, 3), float(rank))).tolist()
            dist.barrier()
            dist.destroy_process_group()
        else:
            got = [[0.0] * 3] * 2
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "gathered_first_column": [r[0] for r in got],
                              "self_launched": os.environ.get("HSEFR_BENCH_SELF_LAUNCHED") == "1"}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit("--gpus %d but only %d GPU(s) visible (RCCL needs one GPU per rank)" % (world, n_dev))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    grouped = world > 1 or args.force_group              # the collectives run (at N = 1 only when asked to)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:              # --force-group without a launcher: a free port for the world-1 store
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        if args.backend == "nccl":
            # RCCL prints a version banner on STDOUT when its communicator is created; this program's stdout is ONE JSON line, so the
            # file descriptor points at stderr while the group comes up (the first collective creates the communicator)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from hse_facerec_tf_amd import lowering
    from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, TensorFlowInference

    B, S = args.batch, args.size
    tfi = TensorFlowInference(AGE_GENDER_PB, input_tensor="input_1:0", output_tensor="global_pooling/Mean:0",
                              convert2BGR=True, imageNetUtilsMean=True, input_size=(S, S), max_batch=B, device=dev_index)
    eng, plan = tfi.engine, tfi.plan
    # synthetic preprocessed batch: U(-128,128) fp32 NHWC, seed 123 (+rank)
    # FOUR different batches fed round-robin (4 x 113 MB > the 256 MiB Infinity Cache): every step reads its input from HBM,
    # not from a cache-resident copy of the previous step's
    gen = torch.Generator(device=dev)
    gen.manual_seed(123 + rank)
    xs = [(torch.rand((B, S, S, 3), device=dev, generator=gen) * 256.0 - 128.0).contiguous() for _ in range(4)]
    x = xs[0]
    x_host = x.cpu().numpy()                 # the CPU baseline runs on the same data
    step_no = [0]

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        step_no[0] += 1
        return eng.forward(xs[step_no[0] & 3])["features"]

    for _ in range(args.warmup):
        out = step()
    # ---- timed region: EXACTLY `steps` forwards, barrier + synchronize on both sides -----------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    per_rank_fps = [B * args.steps / elapsed_local]
    if grouped:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        per_rank_fps = [B * args.steps / float(v.item()) for v in ts]
        elapsed = max(float(v.item()) for v in ts)             # MAX over ranks
    assert bool(torch.isfinite(out).all())

    # ---- sustained rate: the timed region above is ~25 ms on a part whose clock sags under dense MFMA
    # streams; the same forward for >= 1 s of warm-up and then >= 2 s, in windows of 100 steps bracketed by events on the
    # forward's stream.  Reported beside `value`, never instead of it.
    sustained = None
    if not args.no_sustained:
        tw, nw = time.perf_counter(), 0
        while time.perf_counter() - tw < args.sustained_warmup_s:          # warm-up by the clock (also yields the per-step estimate)
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            nw += 50
        per = max((time.perf_counter() - tw) / nw, 1e-5)
        n_win = max(2, int(args.sustained_s * 1.05 / (100 * per)) + 1)
        if grouped and world > 1:      # every rank runs the same number of windows (their count decides how often barrier() is called)
            nt = torch.tensor([n_win], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(nt, op=dist.ReduceOp.MAX)
            n_win = int(nt.item())
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
        # socket power and shader clock while the windows run (rocm-smi from a thread, rank 0 only): the MFMA-dense kernels of this
        # forward run under the chip's power management (DESIGN.md lesson 56), so a roofline fraction is read against these
        smi = {}

        def smi_sample():
            try:
                time.sleep(min(0.5, args.sustained_s / 2))
                o = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
                for ln in o.splitlines():
                    if "Socket Graphics Package Power" in ln or "Average Graphics Package Power" in ln:
                        smi["socket_power_w"] = float(ln.rsplit(":", 1)[1])
                    elif "sclk clock level" in ln and "Mhz" in ln:
                        smi["sclk_mhz"] = float(ln.rsplit("(", 1)[1].split("Mhz")[0])
            except Exception as e:      # (a box without rocm-smi: the fields stay out of the line)
                smi["error"] = repr(e)[:80]
        smi_thread = threading.Thread(target=smi_sample) if rank == 0 else None
        barrier()
        if smi_thread:
            smi_thread.start()
        t1 = time.perf_counter()
        evs[0].record()
        done, plan_n = 0, n_win
        while True:
            for _w in range(plan_n):
                for _ in range(100):
                    out = step()
                done += 1
                if done >= len(evs):
                    evs.append(torch.cuda.Event(enable_timing=True))
                evs[done].record()
            barrier()
            wall = time.perf_counter() - t1
            # (the window count came from the warm-up's estimate; a host that other jobs load makes that estimate too slow and the run too
            # short: more windows until the requested duration is there -- one extra barrier, a few microseconds in two seconds)
            # The decision is COLLECTIVE: every rank branches on the SLOWEST rank's clock -- with each rank reading its own,
            # a wall time near the threshold lets one rank leave while another calls barrier() again: mismatched collectives.
            wall_all = wall
            if grouped and world > 1:
                wt = torch.tensor([wall], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
                dist.all_reduce(wt, op=dist.ReduceOp.MAX)
                wall_all = float(wt.item())
            if wall_all >= args.sustained_s or done >= 2000:
                break
            plan_n = max(1, int((args.sustained_s - wall_all) / (wall_all / done)) + 1)
        n_win = done
        win_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n_win)]
        sustained = {"value": round(B * 100 * n_win / wall, 1), "unit": "faces/s (this rank)", "seconds": round(wall, 3), "steps": 100 * n_win,
                     "warmup_seconds": args.sustained_warmup_s, "ms_per_step": round(wall / (100 * n_win) * 1e3, 4),
                     "window_steps": 100, "window_faces_per_s_min": round(B * 100 / (max(win_ms) * 1e-3), 1),
                     "window_faces_per_s_max": round(B * 100 / (min(win_ms) * 1e-3), 1),
                     "window_faces_per_s_first_last": [round(B * 100 / (win_ms[0] * 1e-3), 1), round(B * 100 / (win_ms[-1] * 1e-3), 1)]}
        if smi_thread:
            smi_thread.join()
            sustained.update({"rocm_smi_" + k: v for k, v in smi.items()})

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
                fl = plan.layer_flops(L) * B
                print("layer %2d kind %d %-34s in %-16s out %-16s s%d  %8.2f us  %7.1f GB/s  %6.1f TF" %
                      (i, L.kind, L.name[:34], L.in_shape, L.out_shape, L.stride, per_op[i] * 1e3,
                       nb / (per_op[i] * 1e-3) / 1e9 if per_op[i] > 0 else 0, fl / (per_op[i] * 1e-3) / 1e12 if per_op[i] > 0 else 0),
                      file=sys.stderr)

    # the exchange of config 5 (one all-gather of one batch of embeddings per rank), outside the timed region
    allgather_ms = None
    if grouped:
        from hse_facerec_tf_amd import gallery
        full = gallery.all_gather_rows(out)
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            full = gallery.all_gather_rows(out)
        barrier()
        allgather_ms = (time.perf_counter() - t1) / 10 * 1e3
        assert torch.equal(full[rank * B:(rank + 1) * B], out)

    # who took part: the line certifies its own ranks -- the communicator's size as the backend reports it after the
    # barriers above, the collective library's version, and every rank's device (name, PCI bus id, the GPU index it bound to) gathered
    # through the group itself: an N > 1 line whose ranks shared a GPU, or that ran fewer ranks than it claims, shows it here
    ranks_info, rccl_ranks, rccl_version = None, None, None
    if grouped:
        rccl_ranks = int(dist.get_world_size())
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_name": props.name,
                "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
                "gcn_arch": getattr(props, "gcnArchName", None), "host": socket.gethostname(), "pid": os.getpid()}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        if args.backend == "nccl":
            try:
                v = torch.cuda.nccl.version()
                rccl_version = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
            except Exception as e:
                rccl_version = "unknown (%r)" % (e,)

    config5 = None
    if not args.no_config5:
        try:
            config5 = run_config5(args, tfi, dev, world, rank, args.backend, dist, grouped)
        except Exception as e:
            if grouped:
                raise
            config5 = {"error": repr(e)}

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    classes = {
        "stem_conv1_dw_pw_dw_fused": lambda L: L.kind in (lowering.OP_STEM2_F16S, lowering.OP_STEM3_F16S),
        "stem_conv1_dw_pw_fused": lambda L: L.kind == lowering.OP_STEM_F16S,
        "conv1_3x3x3_s2": lambda L: L.kind == lowering.OP_CONV_C3,
        "depthwise3x3": lambda L: L.kind == lowering.OP_DWCONV3X3,
        "fused_dw3x3_pw1x1": lambda L: L.kind == lowering.OP_DWPW_F32,
        "fused_dw3x3_pw1x1_f16split": lambda L: L.kind == lowering.OP_DWPW_F16S,
        "pointwise1x1_f32mfma": lambda L: L.kind == lowering.OP_PWCONV_F32 and L.a_log2 == 0,
        "pointwise1x1_f16split": lambda L: L.kind == lowering.OP_PWCONV_F32 and L.a_log2 > 0,
        "fused_pw1x1_dw3x3_f16split": lambda L: L.kind in (lowering.OP_PWDW_PS, lowering.OP_PWGAP_PS),
        "gap": lambda L: L.kind == lowering.OP_GAP,
    }
    # HBM traffic per launch measured with rocprofv3 PMC counters (separate --pmc FETCH_SIZE / WRITE_SIZE
    # passes, FETCH_SIZE doubled per the gfx950 correction) and committed under profiles/ -- counters cannot
    # be read from inside this process, so the latest committed profile is what is reported; `traffic_stale`
    # says whether the kernel sources changed since that profile was taken.
    traffic_by_class, mfma_by_class, traffic_src, traffic_stale = {}, {}, None, None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
        if cands:
            traffic_src = os.path.relpath(cands[-1], ROOT)
            pj = json.load(open(cands[-1]))
            prof = pj["kernels"]
            traffic_stale = pj.get("csrc_hash") != csrc_hash()
            prefixes = {"stem_conv1_dw_pw_dw_fused": ("stem2_fused_kernel", "stem3_", "stem4_", "stem5_"), "stem_conv1_dw_pw_fused": "stem_fused_kernel", "conv1_3x3x3_s2": "conv3x3_c3",
                        "depthwise3x3": "dwconv3x3_kernel",
                        "pointwise1x1_f32mfma": "pwconv_f32_",
                        "pointwise1x1_f16split": lambda k: k.startswith("pwconv_f16s_kernel") or (k.startswith("pwconv_ps_") and k.rstrip().endswith(", 0>")),
                        "fused_pw1x1_dw3x3_f16split": lambda k: k.startswith("pwconv_ps_") and not k.rstrip().endswith(", 0>"), "gap": "hsefr::gap_kernel",
                        "fused_dw3x3_pw1x1": "dwpw_fused_kernel", "fused_dw3x3_pw1x1_f16split": ("dwpw3_f16s_kernel", "dwpw2_f16s_kernel", "dwpw_f16s_kernel")}
            for cls, pre in prefixes.items():
                rows = [v for k, v in prof.items() if (pre(k) if callable(pre) else k.startswith(pre))]      # (startswith takes a tuple too)
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

    def layer_weight_bytes(L):
        return 4 * sum(a.size for a in (L.w, L.scale, L.shift, L.w2, L.shift2, L.w0, L.shift0, L.w3, L.scale3, L.shift3) if a is not None)

    kernels = []
    if per_op is not None:
        for name, pred in classes.items():
            idx = [i for i, L in enumerate(plan.layers) if pred(L)]
            if not idx:
                continue
            ms = float(per_op[idx].sum())
            nbytes = sum(4 * (int(np.prod(plan.layers[i].in_shape)) + int(np.prod(plan.layers[i].out_shape))) for i in idx) * B \
                + sum(layer_weight_bytes(plan.layers[i]) for i in idx)
            flops = sum(plan.layer_flops(plan.layers[i]) for i in idx) * B          # the class's OWN layers
            # a fused block also saves writing + re-reading the depthwise result: report both byte counts
            unfused_extra = sum(2 * 4 * int(np.prod(plan.layers[i].out_shape[:2])) * plan.layers[i].in_shape[2] for i in idx
                                if plan.layers[i].kind in (lowering.OP_DWPW_F32, lowering.OP_DWPW_F16S, lowering.OP_PWDW_PS)) * B \
                + sum(2 * 2 * 4 * int(np.prod(plan.layers[i].out_shape[:2])) * 32 for i in idx
                      if plan.layers[i].kind == lowering.OP_STEM_F16S) * B \
                + sum(2 * 4 * ((plan.layers[i].in_shape[0] + 1) // 2) * ((plan.layers[i].in_shape[1] + 1) // 2) * (32 + 32 + 64) for i in idx
                      if plan.layers[i].kind in (lowering.OP_STEM2_F16S, lowering.OP_STEM3_F16S)) * B
            launches = len(idx)
            note = None
            if name == "pointwise1x1_f32mfma":
                bound, achieved, peak, unit = "mfma", flops / (ms * 1e-3) / 1e12, MFMA_F32_PEAK_TF, "TFLOP/s"
            elif name in ("pointwise1x1_f16split", "fused_pw1x1_dw3x3_f16split"):
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
                            "traffic_stale": traffic_stale,
                            "mfma_util_pmc": None if not mfma_by_class.get(name) else round(mfma_by_class[name], 4),
                            "note": note})
    dominant = max(kernels, key=lambda k: k["ms_per_step"]) if kernels else None
    dw = next((k for k in kernels if k["kernel"] == "depthwise3x3"), None)

    def roof(k):
        if k is None:
            return None
        return {"kernel": k["kernel"], "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"],
                "unit": k["unit"], "frac": k["frac"], "traffic": k["traffic"], "traffic_stale": k["traffic_stale"],
                "avg_launch_us": k["avg_launch_us"],
                "launches_per_step": k["launches_per_step"], "mfma_util_pmc": k.get("mfma_util_pmc"),
                **({"peak_note": "the dense f16 peak at 2.4 GHz / 3 split products; inside back-to-back forwards this kernel's waves count 1.67 GHz "
                                 "(tools/ps_clock_net.py, DESIGN.md lesson 56): the chip's power management, not the kernel, sets that clock"}
                   if k["bound"] == "mfma" else {})}

    # the depthwise class, honestly split: the layers that run as their own kernel vs the ones inside fused kernels
    roofline_dw = roof(dw)
    if roofline_dw is not None:
        standalone = [L for L in plan.layers if L.kind == lowering.OP_DWCONV3X3]
        roofline_dw["covers"] = "the %d STANDALONE depthwise layers only: %s" % (
            len(standalone), ", ".join("%dx%dx%d/s%d" % (L.in_shape[0], L.in_shape[1], L.in_shape[2], L.stride) for L in standalone))
        inside = []
        for k in kernels:
            if k["kernel"] in ("stem_conv1_dw_pw_dw_fused", "stem_conv1_dw_pw_fused", "fused_dw3x3_pw1x1", "fused_dw3x3_pw1x1_f16split"):
                Ls = [L for L in plan.layers if classes[k["kernel"]](L)]
                dws = []
                for L in Ls:
                    if L.kind in (lowering.OP_STEM2_F16S, lowering.OP_STEM3_F16S):
                        h1 = (L.in_shape[0] + 1) // 2
                        dws += ["%dx%dx32/s1" % (h1, h1), "%dx%dx64/s2" % (h1, h1)]
                    elif L.kind == lowering.OP_STEM_F16S:
                        dws += ["%dx%dx32/s1" % (L.out_shape[0], L.out_shape[1])]
                    else:
                        dws += ["%dx%dx%d/s%d" % (L.in_shape[0], L.in_shape[1], L.in_shape[2], L.stride)]
                inside.append({"kernel": k["kernel"], "depthwise_layers": dws, "frac_of_hbm_roofline": k["frac"],
                               "unfused_equivalent_frac": None if not k["unfused_equivalent_gbs"] else round(k["unfused_equivalent_gbs"] / HBM_PEAK_GBS, 4),
                               "ms_per_step": k["ms_per_step"]})
        roofline_dw["inside_fused_kernels"] = inside
        roofline_dw["note"] = ("`frac` is the standalone class; the largest depthwise layers (96x96, 48x48, 24x24x256) run inside the fused "
                               "kernels listed under inside_fused_kernels, whose own fractions are the honest figure for them")

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle.torch_cpu import time_fused_loop, time_reference_loop
        # threads actually used: the cores this process may run on, capped at 32 (batch-1 convolutions
        # of this size stop scaling long before that; oversubscribing a cgroup-limited box is far slower)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        from hse_facerec_tf_amd.decode_pool import cpu_quota
        quota = cpu_quota()                  # the container's cgroup CPU quota (round-5 boxes: 16 of the 256 CPUs listed): threads beyond it are throttled
        cores = max(1, min(avail, 32, int(quota) if quota else 32))
        bud = args.cpu_baseline_seconds
        out_t = "global_pooling/Mean:0"
        fps, n_img = time_reference_loop(AGE_GENDER_PB, out_t, x_host[:32], cores, budget_s=bud, batch=1)
        fps256, n256 = time_reference_loop(AGE_GENDER_PB, out_t, x_host, cores, budget_s=bud, batch=B)
        ffps, fn_img = time_fused_loop(AGE_GENDER_PB, out_t, x_host[:32], cores, budget_s=bud, batch=1)
        ffps256, fn256 = time_fused_loop(AGE_GENDER_PB, out_t, x_host, cores, budget_s=bud, batch=B)
        cpu_baseline = {"value": round(ffps, 2), "unit": "faces/s", "cores": cores, "kind": "port", "host_cpus_listed": avail, "cgroup_quota_cpus": quota,
                        "sample": "%d images of the same synthetic %dx%dx3 batch through the reference's batch-1 loop (facerec_test.py:394) "
                                  "on the FUSED CPU port of the same frozen graph (oracle/torch_cpu.py FusedChainCPU: per-channel scales "
                                  "folded into the kernels, bias in the convolution, Relu/Minimum/Maximum as one in-place clamp, "
                                  "channels_last, torch-CPU/oneDNN fp32) -- the faster of the two CPU ports; not TensorFlow" % (fn_img, S, S),
                        "variants": {
                            "fused_batch1": {"value": round(ffps, 2), "images": fn_img,
                                             "what": "fused port, one image per call (the reference's calling pattern)"},
                            "fused_batch%d" % B: {"value": round(ffps256, 2), "images": fn256, "what": "fused port, whole batches: the CPU's best case"},
                            "op_by_op_batch1": {"value": round(fps, 2), "images": n_img,
                                                "what": "the frozen graph node by node (158 ops per image, Relu/Minimum/Maximum as three passes): "
                                                        "how an unoptimised graph executor runs it; round 1-2's cpu_baseline"},
                            "op_by_op_batch%d" % B: {"value": round(fps256, 2), "images": n256, "what": "node by node, whole batches"}},
                        "note": "a reported baseline, not the target: a large GPU/CPU ratio says nothing about kernel quality"}

    other = None
    if world == 1 and not args.no_other_configs:
        other = run_other_configs(args, dev)
    latency = None
    if world == 1 and not args.no_latency:
        try:
            latency = run_latency_batch1(args, dev)
        except Exception as e:
            latency = {"error": repr(e)}
    pipeline = None
    if world == 1 and not args.no_pipeline:
        try:
            pipeline = run_pipeline(args, tfi, dev)
        except Exception as e:          # a failing side measurement must not take the headline line down
            pipeline = {"error": repr(e)}

    def side(prefix):
        for o in other or []:
            if o.get("config", "").startswith(prefix) and "value" in o:
                return o["value"]
        return None

    line = {
        "metric": "faces/sec embedding-extract (MobileNet-192, bs=256)",
        "value": round(value, 1), "unit": "faces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (four U(-128,128) batches fed round-robin: 452 MB > the 256 MiB Infinity Cache)",
        "config": {"workload": "MobileNet-v1 192x192x3 embeddings (1024-D), batch %d per GPU, fp32 -- BASELINE configs[1]" % B,
                   "global_batch": B * world, "input": [S, S, 3],
                   "arithmetic": "fp32 activations, weights and accumulators; depthwise in fp32 FMA; "
                                 + ("conv1 AND the pointwise products as two-term f16 splits of both operands on the f16 MFMA (3 products "
                                    "per fp32 product, error <= 3*2^-22 per product: fp32-grade, 1e-6 end to end vs the fp64 oracle; conv1's "
                                    "input is bounded by the declared input_bound and checked on the device)"
                                    if any(L.kind == lowering.OP_STEM3_F16S for L in plan.layers) else
                                    "conv1 on the fp32 MFMA; pointwise products as two-term f16 splits of both operands on the f16 MFMA "
                                    "(3 products per fp32 product, error <= 3*2^-22 per product: fp32-grade)"
                                    if any(L.a_log2 for L in plan.layers) else "conv1 and pointwise products on the fp32 MFMA"),
                   "weights": "trunk of age_gender_tf2_new-01-0.14-0.92_quantized.pb (the reference's only shipped graph)",
                   "parallelism": "%d independent replicas, gallery sharded by image, one all-gather of embeddings" % world,
                   "backend": None if not grouped else ("RCCL (torch 'nccl')" if args.backend == "nccl" else "gloo"),
                   "process_group": bool(grouped),
                   "rccl_ranks": rccl_ranks, "rccl_version": rccl_version, "ranks": ranks_info,
                   "distinct_gpus": None if not ranks_info else len({(r["host"], r["pci_bus_id"]) for r in ranks_info}),
                   "allgather_gbs": None if not allgather_ms else round(world * B * int(out.shape[1]) * 4 / (allgather_ms * 1e-3) / 1e9, 2),
                   # the knobs that change WHAT is benchmarked: effective values (keyword arguments only: the product reads
                   # no plan option from the environment)
                   "pw_math": "f16split" if any(L.a_log2 for L in plan.layers) else "f32",
                   "input_bound": tfi.input_bound,
                   "plan_kinds": [int(L.kind) for L in plan.layers],
                   "op_events": "second pass of the same %d steps, HIP events on the forward stream" % args.steps if use_events else None,
                   # one number per side config, measured in this run (details under other_configs / sustained / config5)
                   "resnet50_bf16_faces_per_s": side("BASELINE configs[2]: ResNet-50"),
                   "resnet50_f32_faces_per_s": side("BASELINE configs[2] in the fp32-grade mode"),
                   "agegender_bs512_faces_per_s": side("BASELINE configs[3]"),
                   "mobilenet192_strict_f32_faces_per_s": side("BASELINE configs[1] with pw_math='f32'"),
                   "mtcnn_ms": side("MTCNN detection"),
                   "sustained_faces_per_s": None if sustained is None else sustained["value"],
                   "config5_total_ms": None if not config5 or "total_ms" not in config5 else config5["total_ms"]},
        "per_rank_faces_per_s": [round(v, 1) for v in per_rank_fps],
        "ms_per_step_instrumented": None if instrumented_ms is None else round(instrumented_ms, 4),
        "roofline": roof(dominant), "roofline_depthwise": roofline_dw, "kernels": kernels,
        "cpu_baseline": cpu_baseline,
        "sustained": sustained,
        "allgather_ms": None if allgather_ms is None else round(allgather_ms, 4),
        "config5": config5,
        "other_configs": other,
        "latency_batch1": latency,
        "pipeline": pipeline,
        "device_bytes": eng.device_bytes,
        "csrc_hash": csrc_hash(),
    }
    print(json.dumps(line))
    sys.stdout.flush()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
