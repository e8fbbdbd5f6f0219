#!/usr/bin/env python3
"""bench.py -- symmetric D1+D2 PSNR throughput of the CloudPair/MetricCalculator hot path.

    python bench.py --gpus N --steps K --warmup W [--engine auto|brute|grid] [--points 1000000]

One *step* = one pass of the hot path over one synthetic cloud pair already resident in HBM:
the search-structure build (the reference's two KD-tree builds, cloud_pair.py:65), both directional exact
1-NN sweeps (cloud_pair.py:67-78) with the D2 point-to-plane projection (metric.py:146-153) fused in, the
np.sum / np.max reductions of the D1 and D2 columns, the cross-rank exchange (N > 1) and the PSNR /
symmetric aggregation on the host -- i.e. GeoMSE, GeoPSNR and GeoHausdorffDistance for point_to_plane in
{False, True}, left, right and symmetric, evaluated through the product's own MetricCalculator DAG.  The
PSNR peak (max extent of A's minimal OBB, CPU/Qhull code in the reference) is injected: it is not part of
the GPU path.

metric  = Mpoints/s = (N_ref + N_deg) / time per step, whole job over all ranks.
N > 1   = one process per GPU (torch.distributed, backend nccl = RCCL), the pair split by direction first and
          by query rows inside each half (pccm_set_shard_dir); total problem size fixed (BASELINE.json:
          N_ref = N_deg = 1M at 1/2/4/8 GPUs) -> "scaling": "strong".

Extra records on the same line (N = 1): `full_report` (the step plus the self search behind Min/MaxSqrtDistance,
which options.py:36-37 always requests), `pair_stream` (two resident pairs taking turns: throughput of a sequence), `cold_pair` (a fresh context, nothing inherited), `brute` (the
brute-force engine north_star names: k1_scan against the fp32 vector roofline), `end_to_end` (fresh pair incl.
H2D), `cpu_baseline` / `cpu_reference_pattern` and the same-run parity gate against the oracle.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector == f32-MFMA dense peak
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec peak
FLOP_PER_PAIR = 8            # 3 sub + 1 mul + 2 fma
PROFILE_ROUND = "r04"


def synth(n):
    """SURVEY.md section 8(d) synthetic inputs (independent uniform clouds, unit normals)."""
    a = np.random.default_rng(1234).random((n, 3), dtype=np.float32)
    b = np.random.default_rng(5678).random((n, 3), dtype=np.float32)

    def unit(seed):
        g = np.random.default_rng(seed).standard_normal((n, 3), dtype=np.float32)
        return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)

    return a, b, unit(4321), unit(8765)


def synth_content(target=800_000, seed=11):
    """A configs[4]-class pair (BASELINE.json: 8i longdress vs decoded rates; the real files are not here): a 10-bit voxelised
    closed surface of `target` points (integer coordinates, exact distance ties are the rule) and a "decoded" version of it --
    voxel jitter, duplicates merged, a tenth of the points dropped.  No normals (PLY content has none)."""
    rng = np.random.default_rng(seed)
    m = 6 * target
    u = rng.random(m) * 2 * np.pi
    v = np.arccos(2 * rng.random(m) - 1)
    r = 250 + 22 * np.sin(3 * u) * np.sin(5 * v) + 6 * np.sin(17 * u + 3 * v)
    p = np.stack([512 + r * np.sin(v) * np.cos(u), 512 + 0.62 * r * np.sin(v) * np.sin(u), 512 + r * np.cos(v)], 1)
    a = np.unique(np.round(p).astype(np.float32), axis=0)
    a = a[rng.permutation(len(a))[:target]]
    b = np.unique((a + np.rint(rng.normal(0, 0.45, a.shape))).astype(np.float32), axis=0)
    b = b[rng.random(len(b)) >= 0.1]
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def cpu_baseline(a, b, na, nb):
    """The oracle (CPU restatement: exact kd-tree 1-NN in C/OpenMP + NumPy reductions) timed on this
    host for the SAME workload as one GPU step.  Reported baseline, not a target."""
    from oracle import oracle as orc
    orc.build()
    threads = orc.num_threads()
    t0 = time.perf_counter()
    pair = orc.OraclePair(a, b, na, nb, method="kdtree")
    rep = pair.report(hausdorff=False, point_to_plane_=True, peak=1.0)
    hl, hr = pair.geo_hausdorff(True, False), pair.geo_hausdorff(False, False)
    dt = time.perf_counter() - t0
    n = a.shape[0] + b.shape[0]
    return {"value": n / dt / 1e6, "unit": "Mpoints/s", "cores": threads, "kind": "port",
            "sample": f"full workload once: {a.shape[0]} vs {b.shape[0]} points, 2 kd-tree builds + 2 sweeps "
                      f"+ D1/D2 reductions, {dt:.2f} s"}, rep, (hl, hr), pair


def cpu_reference_pattern(a, b, na, nb, sample=20000):
    """The reference's own calling pattern on the CPU, on a prefix of the workload: one Python-level nearest-neighbour
    call per point out of np.apply_along_axis (cloud_pair.py:16-32) and one np.dot per point for D2 (metric.py:146-153),
    single thread -- against the oracle's kd-tree instead of Open3D's (absent here).  Scaled linearly to a rate."""
    from oracle import oracle as orc
    m = min(sample, a.shape[0], b.shape[0])
    a64, b64, nb64 = a.astype(np.float64), b.astype(np.float64), nb.astype(np.float64)
    tree = orc.KDTree(b64)                                     # the full searched cloud, as in the reference
    t0 = time.perf_counter()
    found = np.apply_along_axis(lambda p: tree.search_1nn(p), 1, a64[:m])          # (idx, d2) per point
    idx = found[:, 0].astype(np.int64)
    err = a64[:m] - b64[idx]
    proj = np.zeros(m)
    for i in range(m):                                                         # metric.py:148-152
        proj[i] = np.dot(err[i], nb64[i])
    mse1, mse2 = np.sum(found[:, 1]) / m, np.sum(np.square(proj)) / m
    dt = time.perf_counter() - t0
    tree.close()
    # one direction of m points measured; a step is both directions over all N points
    return {"value": m / dt / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "reference calling pattern on the oracle's kd-tree",
            "sample": f"{m} of {a.shape[0]} points of one direction (search in the full other cloud, kd-tree build excluded), "
                      f"{dt:.2f} s; rate assumed linear in the number of query points", "d1_mse_of_sample": float(mse1),
            "d2_mse_of_sample": float(mse2)}


def committed_traffic(kernel, n, world):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE,
    separate rocprofv3 --pmc runs over this same command, corrected as MI355X_MICROARCH.md prescribes: a PMC pass
    cannot run inside this process).  Only quoted for the configuration it was collected on."""
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_traffic.json")) as fh:
            pmc = json.load(fh).get(kernel, {})
        if pmc.get("points") == n and world == 1:
            return pmc["hbm_bytes_per_launch"], f"from the committed profile profiles/{PROFILE_ROUND}/pmc_traffic.json (not measured in this run): " + pmc["note"]
    except (OSError, ValueError, KeyError):
        pass
    return None, "no PMC profile committed for this engine/size"


def self_launch(nproc):
    """Run this same command line under torch.distributed.run with one rank per GPU (what the driver's launch line does) and
    pass its output through.  The parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 200 (5 with --engine brute, whose step is ~0.2 s)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 5 (1 with --engine brute)")
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--engine", default="auto", choices=["auto", "brute", "grid"])
    ap.add_argument("--content-only", action="store_true",
                    help="profiling runs: time only the `content` record's step (voxelised-surface pair, D1 + Hausdorff) and exit")
    ap.add_argument("--content-full-only", action="store_true",
                    help="profiling runs: time only the `content_full` record's step (the same pair, D1 + D2 + colour rows) and exit")
    ap.add_argument("--content-cli-only", action="store_true", help="only the `content_cli` record (files -> report, stage by stage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip full_report / cold_pair / brute / end_to_end (profiling runs)")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--shard-mode", default="direction", choices=["direction", "rows"], help="how N > 1 ranks split the pair")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process (never an exec), before
        # anything in this process has touched HIP or torch.cuda; relay rank 0's JSON line and the child's return code
        sys.exit(self_launch(args.gpus))

    lib = os.path.join(ROOT, "open_pcc_metric_amd", "csrc", "libpccm.so")
    if not os.path.exists(lib) and int(os.environ.get("LOCAL_RANK", "0")) == 0:
        # a fresh checkout (built artefacts are git-ignored): build as __graft_entry__.build() does; the product itself
        # never builds or falls back -- without the library it raises
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.dirname(lib), "-j4"], check=True, stdout=subprocess.DEVNULL)
    for _ in range(600):                      # the other ranks of a multi-GPU launch wait for rank 0's build
        if os.path.exists(lib):
            break
        time.sleep(0.5)
    import torch
    import torch.distributed as dist
    import open_pcc_metric_amd.metric as m
    from open_pcc_metric_amd import _native as nat
    from open_pcc_metric_amd.calculator import MetricCalculator
    from open_pcc_metric_amd.cloud_pair import CloudPair
    from open_pcc_metric_amd.options import CalculateOptions, transform_options
    from open_pcc_metric_amd.point_cloud import PointCloud

    if args.steps is None:
        args.steps = 5 if args.engine == "brute" else 200
    if args.warmup is None:
        args.warmup = 1 if args.engine == "brute" else 5
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                         # a launcher's world size wins over the flag
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: open_pcc_metric_amd has no CPU path")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    group = None
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
        group = dist.group.WORLD

    def content_record(steps):
        """One step on PCC-like content: both clouds resident, grid build + both sweeps + D1 MSE / PSNR / Hausdorff rows."""
        ca, cb = synth_content()
        copts = CalculateOptions(color=None, hausdorff=True, point_to_plane=False)
        # (staged_io: the clouds of this record are freed behind it -- see pccm_set_io_staged; uploads are not in the timed region)
        with CloudPair(PointCloud(ca), PointCloud(cb), extent=[511.0, 322.0, 505.0], device=local, nn_engine=args.engine, staged_io=True,
                       use_graph=not args.no_graph) as cp:
            ce = cp._engine

            def cstep():
                cp.recompute()
                return MetricCalculator(cp).calculate(transform_options(copts)[2:]).as_dict()

            for _ in range(4):
                cstep()
            ce.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                cres = cstep()
            ce.sync()
            dt = (time.perf_counter() - t0) / steps
            cp._use_graph = False
            ce.profile(True)
            ce.profile_reset()
            for _ in range(5):
                cstep()
            ce.sync()
            kus = {k: round(ce.profile_get(k)[0] / 5 * 1e3, 1) for k in nat.KERNEL_CLASSES if ce.profile_get(k)[1]}
            ce.profile(False)
            return {"ms_per_step": round(dt * 1e3, 4), "value": round((len(ca) + len(cb)) / dt / 1e6, 2), "unit": "Mpoints/s",
                    "points": [len(ca), len(cb)], "kernel_us_per_step": kus, "grid_cells": ce.nn_stats(0)["splits"],
                    "mse_left": float(cres[("GeoMSE", True, False)]),
                    "note": "10-bit voxelised closed surface vs a jittered / thinned copy (BASELINE configs[4]-class content; the real "
                            "longdress files are not in the container), D1 MSE / PSNR / Hausdorff, both directions + symmetric; "
                            "distances only, so the searches run on the voxel bricks (pccm_vox.hip; PCCM_VOX=0: the per-thread lattice kernel)"}

    def content_full_record(steps):
        """The configs[4]-shaped report on PCC-like content with everything resident: D1 + D2 (the matched point's normal: the clouds
        differ in size) + colour (ycc) + Hausdorff rows, i.e. searches that must return the matched ROW."""
        ca, cb = synth_content()
        rng = np.random.default_rng(77)

        def unit(n):
            g = rng.standard_normal((n, 3)).astype(np.float32)
            return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)

        def colours(p):
            c = np.stack([128 + 100 * np.sin(p[:, 0] / 37.0), 128 + 100 * np.cos(p[:, 1] / 23.0), 128 + 90 * np.sin(p[:, 2] / 51.0)], 1)
            u8 = np.clip(np.rint(c + rng.normal(0, 6, c.shape)), 0, 255).astype(np.uint8)
            return u8

        pa, pb = PointCloud(ca, unit(len(ca)), colours(ca) / 255.0), PointCloud(cb, unit(len(cb)), colours(cb) / 255.0)
        copts = CalculateOptions(color="ycc", hausdorff=True, point_to_plane=True)
        with CloudPair(pa, pb, extent=[511.0, 322.0, 505.0], device=local, nn_engine=args.engine, normal_index="neighbour", staged_io=True,
                       use_graph=not args.no_graph) as cp:
            ce = cp._engine

            def cstep():
                cp.recompute()
                return MetricCalculator(cp).calculate(transform_options(copts)).as_dict()

            for _ in range(4):
                cstep()
            ce.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                cres = cstep()
            ce.sync()
            dt = (time.perf_counter() - t0) / steps
            cp._use_graph = False
            ce.profile(True)
            ce.profile_reset()
            for _ in range(5):
                cstep()
            ce.sync()
            kus = {k: round(ce.profile_get(k)[0] / 5 * 1e3, 1) for k in nat.KERNEL_CLASSES if ce.profile_get(k)[1]}
            ce.profile(False)
            return {"ms_per_step": round(dt * 1e3, 4), "value": round((len(ca) + len(cb)) / dt / 1e6, 2), "unit": "Mpoints/s",
                    "points": [len(ca), len(cb)], "rows": len(cres), "kernel_us_per_step": kus, "grid_cells": ce.nn_stats(0)["splits"],
                    "d2_mse_left": float(cres[("GeoMSE", True, True)]), "color_mse_left_y": float(cres[("ColorMSE", True, "ycc")][0]),
                    "note": "the content pair with unit normals and uchar colours resident: every row of --color ycc --hausdorff "
                            "--point-to-plane --normal-index neighbour (32 rows: D1, D2 with the matched point's normal, colour, Hausdorff, "
                            "the self search); matched rows come from the voxel-brick search (round 4; PCCM_VOX=0: a rebuilt grid + the "
                            "per-thread lattice kernel)"}

    def content_cli_record():
        """BASELINE configs[4]'s literal shape through the product's front door: PLY files without normals, uchar colours, one
        reference cloud against one and against three decoded versions -- read, upload, normal estimation, minimal OBB, searches
        and the full report (--color ycc --hausdorff --point-to-plane --normal-index neighbour), stage by stage (host clock, a
        stream sync behind every stage).  Three rates: the reference cloud's share of the work is done once (CloudPair.with_reconst)."""
        import tempfile
        from open_pcc_metric_amd.io import read_point_cloud, write_point_cloud
        ca, cb = synth_content()
        rng = np.random.default_rng(78)

        def colours(p):
            c = np.stack([128 + 100 * np.sin(p[:, 0] / 37.0), 128 + 100 * np.cos(p[:, 1] / 23.0), 128 + 90 * np.sin(p[:, 2] / 51.0)], 1)
            return np.clip(np.rint(c + rng.normal(0, 6, c.shape)), 0, 255).astype(np.uint8) / 255.0

        decoded = [cb]
        for step in (2, 4):
            q = np.unique((np.round(ca / step) * step).astype(np.float32), axis=0)
            decoded.append(np.ascontiguousarray(q[rng.random(len(q)) >= 0.03]))
        copts = CalculateOptions(color="ycc", hausdorff=True, point_to_plane=True)
        with tempfile.TemporaryDirectory() as tmp:
            ref = os.path.join(tmp, "ref.ply")
            write_point_cloud(ref, PointCloud(ca, None, colours(ca)), coord_dtype="float")
            paths = []
            for k, d in enumerate(decoded):
                paths.append(os.path.join(tmp, f"dec{k}.ply"))
                write_point_cloud(paths[-1], PointCloud(d, None, colours(d)), coord_dtype="float")

            cli_eng = nat.acquire_engine(local)      # one context for every run below (the pool hands out whichever was released last)

            _kept = []

            def run(paths_):
                st = {"read": 0.0, "upload_and_searches": 0.0, "normals": 0.0, "extent": 0.0, "report": 0.0}
                t_all = time.perf_counter()
                t0 = time.perf_counter()
                origin = read_point_cloud(ref)
                st["read"] += time.perf_counter() - t0
                pair = None
                for pth in paths_:
                    t0 = time.perf_counter()
                    dec = read_point_cloud(pth)
                    t1 = time.perf_counter()
                    first = pair is None
                    if os.environ.get("BENCH_CLI_KEEP"):
                        _kept.append((dec, pair, origin))          # (diagnosis: nothing the runtime may have pinned is given back)
                    if first:
                        cli_eng.reset()
                        pair = CloudPair(origin, dec, nn_engine=args.engine, normal_index="neighbour", _engine=cli_eng, staged_io=True)   # (as handler.py)
                    else:
                        pair = pair.with_reconst(dec)
                    skip = os.environ.get("BENCH_CLI_SKIP", "")
                    if first and "profile" not in skip:
                        pair._engine.profile(True)
                        pair._engine.profile_reset()
                    pair._engine.sync()
                    t2 = time.perf_counter()
                    if os.environ.get("BENCH_CLI_DEBUG"):
                        print("cli pair", len(dec.points), "stage ms", round((t2 - t1) * 1e3, 2), {k: round(pair._engine.profile_get(k)[0] * 1e3, 1) for k in ("grid_build", "grid_query", "grid_finish")},
                              [pair._engine.nn_stats(d) for d in (0, 1)], file=sys.stderr)
                    if "normals" not in skip:
                        pair._require_normals(0)
                        pair._require_normals(1)
                    pair._engine.sync()
                    t3 = time.perf_counter()
                    if "extent" not in skip:
                        pair.get_extent()
                    t4 = time.perf_counter()
                    text = ""
                    if "report" not in skip:
                        with np.errstate(divide="ignore"):
                            text = MetricCalculator(pair).calculate(transform_options(copts)).as_df().to_string()
                    t5 = time.perf_counter()
                    for key, dtv in (("read", t1 - t0), ("upload_and_searches", t2 - t1), ("normals", t3 - t2), ("extent", t4 - t3), ("report", t5 - t4)):
                        st[key] += dtv
                kus = {k: round(pair._engine.profile_get(k)[0] * 1e3, 1) for k in nat.KERNEL_CLASSES if pair._engine.profile_get(k)[1]}
                pair._engine.profile(False)
                pair.close()
                total = time.perf_counter() - t_all
                return {**{k: round(v * 1e3, 3) for k, v in st.items()}, "total_ms": round(total * 1e3, 3), "report_rows": text.count("\n"),
                        "kernel_us_behind_the_first_upload": kus}

            run(paths)                                       # warm: contexts and their buffers at every size, kernels, file cache
            one = run(paths[:1])
            three = run(paths)
            separate = sum(run([pth])["total_ms"] for pth in paths)
            nat.release_engine(cli_eng)
        return {"points": [len(ca)] + [len(d) for d in decoded], "one_rate_ms": one, "three_rates_ms": three,
                "three_separate_runs_ms": round(separate, 3),
                "note": "PLY (binary, float xyz + uchar rgb, no normals) -> full report as text; stages in ms on the host clock with a stream sync "
                        "behind each; three_rates = one reference cloud against three decoded clouds through CloudPair.with_reconst (the "
                        "reference is read, uploaded, given normals, a minimal OBB and a self search once); three_separate_runs = the sum of "
                        "three one-rate runs"}

    if args.content_only:
        print(json.dumps({"content": content_record(args.steps or 50)}), flush=True)
        return
    if args.content_cli_only:
        print(json.dumps({"content_cli": content_cli_record()}), flush=True)
        return
    if args.content_full_only:
        print(json.dumps({"content_full": content_full_record(args.steps or 50)}), flush=True)
        return

    n = args.points
    a, b, na, nb = synth(n)
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], device=local, nn_engine=args.engine,
                     group=group, use_graph=not args.no_graph, shard_mode=args.shard_mode)   # H2D + ingest happen here, untimed
    eng = pair._engine
    options = CalculateOptions(color=None, hausdorff=False, point_to_plane=True)
    hd_rows = [("GeoHausdorffDistance", True, False), ("GeoHausdorffDistance", False, False)]

    def headline_metrics():
        # Min/MaxSqrtDistance (the self search) are reported apart: `full_report`
        return transform_options(options)[2:] + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]

    def full_metrics():
        return transform_options(options) + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]

    def step(p=pair, metrics=headline_metrics):
        p.recompute()
        return MetricCalculator(p).calculate(metrics()).as_dict()

    def fence():
        if world > 1:
            dist.barrier()
        eng.sync()
        torch.cuda.synchronize()

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    nwarm = max(args.warmup, 3 if not args.no_graph else 0)
    for _ in range(nwarm):   # eager, capture, first replay
        result = step()
    # a full collection of this process (torch, pandas, ... : millions of long-lived objects) is a ~50 ms pause that
    # would land in a random step; park what exists now in the permanent generation.  Nothing of a step is skipped.
    gc.collect()
    gc.freeze()
    eng.profile(args.no_graph)   # HIP events time eager launches; a hipGraph replay cannot carry them (ROCm 7.2)
    eng.profile_reset()
    elapsed, result = timed(step, args.steps)
    prof_leg = "HIP events over the timed region"
    if not args.no_graph:
        # kernel durations: the same steps issued eagerly right after the timed region, same process,
        # same resident data (a kernel runs the same whether a graph or the host launched it)
        pair._use_graph = False
        eng.profile(True)
        eng.profile_reset()
        for _ in range(min(args.steps, 10)):
            result = step()
        prof_leg = f"HIP events over {min(args.steps, 10)} eager steps issued right after the timed hipGraph region"
        pair._use_graph = True
    eng.profile(False)
    prof_steps = args.steps if args.no_graph else min(args.steps, 10)

    ms_per_step = elapsed / args.steps * 1e3
    value = (2 * n) / (elapsed / args.steps) / 1e6

    # dominant kernel of this rank, HIP events on the library's stream
    stats = [eng.nn_stats(d) for d in (0, 1)]
    scan_ms, scan_n = eng.profile_get("scan")
    gq_ms, gq_n = eng.profile_get("grid_query")
    prof = {k: eng.profile_get(k) for k in nat.KERNEL_CLASSES}
    q_rows = sum(eng.shard_range(d)[1] - eng.shard_range(d)[0] for d in (0, 1)) / 2.0     # query rows per direction on this rank
    roofline = None
    if scan_n:
        pairs_per_launch = sum(s["pairs"] for s in stats) / 2.0
        avg_ms = scan_ms / scan_n
        tflops = pairs_per_launch * FLOP_PER_PAIR / (avg_ms * 1e-3) / 1e12
        compulsory = 12.0 * (q_rows + n) + 12.0 * q_rows
        roofline = {"bound": "mfma", "achieved": round(tflops, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / FP32_PEAK_TFLOPS, 4), "traffic": None,
                    "kernel": "k1_scan", "avg_launch_ms": round(avg_ms, 4), "launches": scan_n,
                    "algorithmic_flop_per_launch": pairs_per_launch * FLOP_PER_PAIR,
                    "note": "fp32 vector-ALU bound brute-force scan (no MFMA issued: the exact difference form "
                            "(q-r)^2 is not a contraction); gfx950 f32-MFMA dense peak == fp32 VALU peak = 157.3 TFLOP/s",
                    "hbm_compulsory": {"bytes_per_launch": compulsory,
                                       "achieved_GBs": round(compulsory / (avg_ms * 1e-3) / 1e9, 3),
                                       "peak_GBs": HBM_PEAK_GBS,
                                       "frac": round(compulsory / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}}
    elif gq_n:
        avg_ms = gq_ms / gq_n
        ncells = float(stats[0]["splits"])      # grid engine: pccm_nn_stats reports the cells of the grid it searched
        # one launch serves BOTH directions (pccm_nn_pair).  DESIGN.md section 3, per direction: a query's 16-byte record in,
        # ONE 16-byte result out (the matched record: squared distance and row-indexed projection are formed by the reduction,
        # which reads rows and normals in row order -- round 3; round 2 gathered a 16/24-byte normal per query here and
        # stored {d2, projection}); per searched point its 16-byte record once; cell starts of both clouds, 4 B/cell each
        alg_bytes = 2.0 * (32.0 * q_rows + 16.0 * n + 8.0 * ncells)
        # SURVEY.md section 8(d)'s layout-independent compulsory bytes of search + projection: 12 (N_q + N_r) + 24 N_q, + 12 N
        # normals -- kept as the yardstick of `frac_compulsory` although 12 N_q of normals and 8 of the 24 output bytes now
        # belong to the reduction's pass (its own figure: `reduce_roofline`)
        compulsory = 2.0 * (12.0 * (q_rows + n) + 24.0 * q_rows + 12.0 * q_rows)
        traffic, traffic_note = committed_traffic("k_brick_query", n, world)
        roofline = {"bound": "hbm", "achieved": round(alg_bytes / (avg_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "frac_compulsory": round(compulsory / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "traffic": traffic,
                    "kernel": "k_brick_query (ring-1 search of both directions, one launch; results = matched records)",
                    "avg_launch_ms": round(avg_ms, 4), "launches": gq_n, "algorithmic_bytes_per_launch": alg_bytes,
                    "compulsory_bytes_per_launch": compulsory, "traffic_note": traffic_note,
                    "arithmetic": "fp32 candidate filter in LDS (packed fp32, 13 VALU instructions per pair of candidates, LDS reads software-pipelined), fp32 stop rule",
                    "limiter": "priced against HBM as the contract prescribes, but not bound by it (1.5 of 8 TB/s): PMC busy counters "
                               "(profiles/r04/grid_1M_busy_counters.json) show instruction throughput spread over all pipes -- per wave 930 VALU "
                               "(67 % of the launch per SIMD at four clocks each, 43 % at the guide's two / four), 506 SALU + 123 branches, 176 LDS; "
                               "waves parked 45 % of their cycles, ready-but-unissued 29 %; time = 41 us + 7.7 ns per workgroup over brick sizes "
                               "(per-brick set-up is 39 % of the vector instructions); pipelined LDS reads -5 %, staggered starts / two pairs per "
                               "trip / resident workgroups with prefetch: no gain (DESIGN.md section 3)"}

    reduce_roofline = None
    red_ms, red_n = eng.profile_get("reduce")
    if gq_n and red_n:
        # the reductions' pass over matched records: per row of either cloud its 16-byte result record, its own 16-byte coordinate
        # word and the other cloud's row-indexed normal (16 bytes) -- all streamed in row order
        rbytes = 2.0 * q_rows * 48.0
        reduce_roofline = {"bound": "hbm", "achieved": round(rbytes / (red_ms / red_n * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(rbytes / (red_ms / red_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "avg_launch_ms": round(red_ms / red_n, 4),
                           "algorithmic_bytes_per_launch": rbytes,
                           "kernel": "k_unit_lean (distance + row-indexed projection from matched records, NumPy-order sums and extrema of "
                                     "the D1 and D2 columns of both directions)"}

    line = {
        "metric": "Mpoints/s for symmetric D1+D2 PSNR, N_ref=N_deg=%s" % (f"{n // 1_000_000}M" if n % 1_000_000 == 0 else n),
        "value": round(value, 4), "unit": "Mpoints/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "warmup_run": nwarm, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} vs {n} uniform-random fp32 xyz + unit normals, symmetric D1+D2 MSE/PSNR + D1 Hausdorff "
                               "(BASELINE.json configs[1]+[2])",
                   "engine": args.engine, "sharding": "none" if world == 1 else f"{args.shard_mode} x{args.gpus}",
                   "hip_graph": not args.no_graph,
                   "fallback_queries": [s["fallback_queries"] for s in stats],
                   ("grid_cells" if gq_n else "scan_splits"): [s["splits"] for s in stats]},
        "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "backend": (args.backend if world > 1 else None),
        "roofline": roofline, "roofline_measured_by": prof_leg, "reduce_roofline": reduce_roofline,
        "kernel_us_per_step": {k: round(v[0] / prof_steps * 1e3, 1) for k, v in prof.items() if v[1]},
        "result_sample": {"GeoMSE_sym_d1": float(result[("SymmetricMetric", "GeoMSE", True, False, "GeoMSE", False, False)]),
                          "GeoPSNR_sym_d2": float(result[("SymmetricMetric", "GeoPSNR", True, True, "GeoPSNR", False, True)])},
    }
    if world > 1:
        # what each rank's GPU spent per step (HIP events): the replicated part of a step is the build of the cloud
        # a rank searches; everything else shrinks with the rows it owns
        mine = {"rank": rank, "rows": [list(eng.shard_range(d)) for d in (0, 1)], **line["kernel_us_per_step"]}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        line["per_rank_kernel_us_per_step"] = allr

    if world > 1 and not args.no_extras:
        # for the record, next to the contract's strong-scaling figure: the same workload as N INDEPENDENT pairs, one whole pair
        # per rank and step, no collective in the data path (what sequence.evaluate_pairs does with a codec study's frames)
        with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], device=local, nn_engine=args.engine,
                       use_graph=not args.no_graph) as solo:
            for _ in range(nwarm):
                step(solo)
            dt, _ = timed(lambda: step(solo), args.steps)
        line["independent_pairs"] = {"ms_per_step": round(dt / args.steps * 1e3, 4), "value": round(world * 2 * n / (dt / args.steps) / 1e6, 2),
                                     "unit": "Mpoints/s", "scaling": "weak", "pairs_per_step": world,
                                     "note": "every rank evaluates its own whole pair per step (no sharding, no collective); "
                                             "time = slowest rank, value = all ranks' points"}

    extras = world == 1 and not args.no_extras
    if extras:
        # (1) the full report: the same step plus the self search behind Min/MaxSqrtDistance (options.py:36-37)
        for _ in range(nwarm):
            step(metrics=full_metrics)
        dt, full = timed(lambda: step(metrics=full_metrics), max(10, args.steps // 4))
        k = max(10, args.steps // 4)
        line["full_report"] = {"ms_per_step": round(dt / k * 1e3, 4), "value": round(2 * n / (dt / k) / 1e6, 2), "unit": "Mpoints/s",
                               "steps": k, "note": "step + self search of the origin cloud (third sweep, A vs A) + its min/max "
                                                   "reduction: every row transform_options() returns, plus the D1 Hausdorff rows"}
        for _ in range(nwarm):                         # back to the headline report (re-captures its graph)
            result = step()

        # (1a) a STREAM of resident pairs: two pairs in flight on two contexts of the same GPU, the launch of one issued before
        # the report of the other is made -- what hides the ~50 us the GPU idles between two steps of ONE pair (host wake-up,
        # the Python behind the last kernel, the next graph launch).  Both pairs are the headline pair (second one: the clouds swapped).
        with CloudPair(PointCloud(b, nb), PointCloud(a, na), extent=[1.0, 1.0, 1.0], device=local, nn_engine=args.engine,
                       use_graph=not args.no_graph) as other:
            def finish(p):
                return MetricCalculator(p).calculate(headline_metrics()).as_dict()
            for _ in range(nwarm + 1):
                pair.recompute(); r0 = finish(pair)
                other.recompute(); r1 = finish(other)
            ks = max(20, args.steps)
            fence()
            other._engine.sync()
            t0 = time.perf_counter()
            pair.recompute()
            for _ in range(ks):
                other.recompute()                    # the next pair's launch ...
                s0 = finish(pair)                    # ... before this pair's report
                pair.recompute()
                s1 = finish(other)
            s0 = finish(pair)
            other._engine.sync()
            fence()
            dts = (time.perf_counter() - t0) / (2 * ks + 1)
            same = all(s0[k] == r0[k] for k in r0) and all(s1[k] == r1[k] for k in r1)
            line["pair_stream"] = {"ms_per_pair": round(dts * 1e3, 4), "value": round(2 * n / dts / 1e6, 2), "unit": "Mpoints/s", "pairs": 2 * ks + 1,
                                   "rows_equal_to_the_single_pair_runs": bool(same),
                                   "note": "two resident 1M + 1M pairs on two contexts of one GPU, taking turns: each pair's step (grid build, both "
                                           "sweeps, reductions, report) as in `value`, but the other pair's hipGraph is launched before this pair's "
                                           "report is read -- throughput of a sequence of pairs, not the latency of one; never `value`"}
        for _ in range(nwarm):
            result = step()

        # (1b) PCC-like content: voxelised surfaces take the per-thread search, not the brick kernel
        line["content"] = content_record(30)
        line["content_full"] = content_full_record(20)
        line["content_cli"] = content_cli_record()

        # (2) end to end, for the record (never `value`): a FRESH pair per iteration -- upload of both clouds and their
        # normals from pageable host memory, ingest, both sweeps, the same report -- through the pooled context
        reps = 5
        t_e2e = 0.0
        for it in range(reps + 1):                   # the first iteration warms the pooled context and is not counted
            if it == 1:
                t_e2e = time.perf_counter()
            with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], device=local, nn_engine=args.engine, staged_io=False) as fresh:
                MetricCalculator(fresh).calculate(headline_metrics()).as_dict()
        dt = (time.perf_counter() - t_e2e) / reps
        # ... and with every upload in front of the first search (round 3's order), to see what the new order hides
        t_ser = 0.0
        for it in range(reps + 1):
            if it == 1:
                t_ser = time.perf_counter()
            with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], device=local, nn_engine=args.engine, _uploads_first=True, staged_io=False) as fresh:
                MetricCalculator(fresh).calculate(headline_metrics()).as_dict()
        dt_serial = (time.perf_counter() - t_ser) / reps
        # ... and what a loop over FILES sees: new host arrays for every pair, given back behind it (the copies are not timed).  Direct:
        # the runtime pins each array; when it is freed the driver evicts and restores the GPU queues, and every other pair stands
        # still for ~30 ms.  Staged (what CloudPair chooses for a context's second pair on): through the context's own pinned buffers.
        def fresh_loop(staged, reps=8):
            out, cl, fp = [], None, None
            for it in range(reps + 2):
                # as in `for f in files: cloud = read(f); pair = CloudPair(...)`: the next clouds are made while the previous ones are
                # still referenced, so they land elsewhere and the previous ones are given back afterwards
                nxt = (PointCloud(np.array(a), np.array(na)), PointCloud(np.array(b), np.array(nb)))
                cl = nxt
                t0 = time.perf_counter()
                with CloudPair(*cl, extent=[1.0, 1.0, 1.0], device=local, nn_engine=args.engine, staged_io=staged) as fp:
                    MetricCalculator(fp).calculate(headline_metrics()).as_dict()
                if it >= 2:
                    out.append((time.perf_counter() - t0) * 1e3)
            return out
        # ... and a sequence of fresh pairs through the product's own evaluate_pairs (two host threads, each with a pooled context:
        # one pair's uploads run beside the other's kernels and report)
        from open_pcc_metric_amd.sequence import evaluate_pairs
        seq_items = [(PointCloud(a, na), PointCloud(b, nb)) for _ in range(8)]
        evaluate_pairs(seq_items[:2], options, device=local, workers=2, extent=[1.0, 1.0, 1.0], nn_engine=args.engine)
        t_seq = time.perf_counter()
        seq_rows = evaluate_pairs(seq_items, options, device=local, workers=2, extent=[1.0, 1.0, 1.0], nn_engine=args.engine)
        dt_seq = (time.perf_counter() - t_seq) / len(seq_items)
        h2d_bytes = a.nbytes + b.nbytes + na.nbytes + nb.nbytes
        # the upload alone, same buffers, same context: what PCIe and the pageable-memory path allow
        ue = nat.Engine(local)
        ue.set_cloud(0, a); ue.set_normals(0, na); ue.set_cloud(1, b); ue.set_normals(1, nb); ue.sync()
        t_up = time.perf_counter()
        for _ in range(reps):
            ue.set_cloud(0, a); ue.set_normals(0, na); ue.set_cloud(1, b); ue.set_normals(1, nb)
        ue.sync()
        up = (time.perf_counter() - t_up) / reps
        ue.close()
        fresh_staged, fresh_direct = fresh_loop(True), fresh_loop(False)      # (last: the direct loop leaves mappings behind whose tear-down disturbs whatever runs next)
        # the same fresh pair with the points alone uploaded before the searches start (the round-3 order for comparison is gone:
        # CloudPair announces normals and flushes them behind the sweeps); what a resident pair costs for the same work once,
        # eagerly (no graph), is the non-PCIe share
        line["end_to_end"] = {"ms_per_pair": round(dt * 1e3, 4), "value": round(2 * n / dt / 1e6, 2), "unit": "Mpoints/s",
                              "h2d_ms": round(up * 1e3, 4), "h2d_bytes": int(h2d_bytes), "h2d_GBs": round(h2d_bytes / up / 1e9, 1),
                              "ms_per_pair_uploads_first": round(dt_serial * 1e3, 4), "hidden_ms": round((dt_serial - dt) * 1e3, 4),
                              "fresh_arrays_ms_per_pair": {"direct": [round(x, 2) for x in fresh_direct], "staged": [round(x, 2) for x in fresh_staged],
                                                           "mean_direct": round(sum(fresh_direct) / len(fresh_direct), 2),
                                                           "mean_staged": round(sum(fresh_staged) / len(fresh_staged), 2)},
                              "ms_per_pair_in_a_sequence": round(dt_seq * 1e3, 4), "sequence_rows_equal": all(r[k] == result[k] for r in seq_rows for k in result if k in r),
                              "note": "ms_per_pair etc.: the SAME host arrays every iteration, handed to the runtime as they are (staged_io=False): what rounds 1-3 measured; fresh_arrays_ms_per_pair: new arrays per pair, freed behind it; ms_per_pair_in_a_sequence: eight fresh pairs through evaluate_pairs (two host threads with a pooled context each; every row of transform_options(), i.e. with the self search); otherwise: fresh CloudPair per iteration through the pooled context: H2D of 2 clouds (pageable fp32), ingest, grid "
                                      "decisions inherited, both sweeps from the caller's row order (no spatial copy: a pair that is searched "
                                      "once never makes one), the normals' H2D (2 x 12 MB, announced before and flushed behind the sweeps: "
                                      "pccm_set_normals_deferred) on a copy stream beside them, report; h2d_ms = the four uploads + ingests "
                                      "alone, serialised (PCIe Gen5 x16: 63 GB/s spec -> 0.76 ms for these bytes at best); ms_per_pair_uploads_first = the "
                                      "same pair with all four uploads in front of the first search (round 3's order), hidden_ms the difference: "
                                      "the searches are ~0.13 ms of GPU work, which is all an upload of 0.55 ms can hide"}

        # (3) a cold pair: brand-new context (no pooled allocations), nothing inherited from an earlier pair
        # (PCCM_GRID_NO_REUSE=1: the cell-edge decision with its histogram passes and host round trips runs), one report
        os.environ["PCCM_GRID_NO_REUSE"] = "1"
        cold = []
        for _ in range(3):
            ce = nat.Engine(local)
            t0 = time.perf_counter()
            cp = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], nn_engine=args.engine, _engine=ce)
            MetricCalculator(cp).calculate(headline_metrics()).as_dict()
            cold.append(time.perf_counter() - t0)
            ce.close()
        del os.environ["PCCM_GRID_NO_REUSE"]
        line["cold_pair"] = {"ms_per_pair": round(min(cold) * 1e3, 3), "ms_all": [round(c * 1e3, 3) for c in cold],
                             "note": "new context per pair (hipMalloc of every buffer, stream, events), upload, ingest, grid decisions "
                                     "taken from scratch, sweeps, report; context teardown not included"}

        # (4) the brute-force engine (the formulation north_star names), a few steps: k1_scan against the fp32 vector peak
        if args.engine != "brute":
            with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], device=local, nn_engine="brute") as bp:
                be = bp._engine
                step(bp)
                be.profile(True)
                be.profile_reset()
                bsteps = 2
                t0 = time.perf_counter()
                for _ in range(bsteps):
                    bres = step(bp)
                be.sync()
                bdt = (time.perf_counter() - t0) / bsteps
                sms, sn = be.profile_get("scan")
                be.profile(False)
                pairs = sum(be.nn_stats(d)["pairs"] for d in (0, 1)) / 2.0
                tfl = pairs * FLOP_PER_PAIR / (sms / sn * 1e-3) / 1e12 if sn else 0.0
                line["brute"] = {"ms_per_step": round(bdt * 1e3, 2), "value": round(2 * n / bdt / 1e6, 2), "unit": "Mpoints/s", "steps": bsteps,
                                 "k1_scan_ms": round(sms / sn, 3) if sn else None, "k1_scan_TFLOPs": round(tfl, 2),
                                 "k1_scan_frac_of_fp32_vector_peak": round(tfl / FP32_PEAK_TFLOPS, 4),
                                 "same_rows_as_grid_engine": all(bres[key] == result[key] for key in result)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, rep, hd, opair = cpu_baseline(a, b, na, nb)
        line["cpu_baseline"] = base
        line["cpu_reference_pattern"] = cpu_reference_pattern(a, b, na, nb)
        # same-run parity gate: every row of the step equals the oracle bit for bit
        bad = [k for k, v in rep.items() if k in result and not (result[k] == v)]
        bad += [k for k, v in zip(hd_rows, hd) if not (result[k] == v)]
        if extras:
            mn, mx = opair.min_max_sqrt()
            bad += [k for k, v in ((("MinSqrtDistance",), mn), (("MaxSqrtDistance",), mx)) if not (full[k] == v)]
        line["parity_vs_oracle"] = "bit-exact" if not bad else f"MISMATCH in {bad}"
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
