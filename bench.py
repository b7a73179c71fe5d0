#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on N GPUs of one node.

A "step" is one pass of the whole path over one synthetic forest cloud that is already
resident in HBM: Morton keys + radix sort + octree/LOD -> kNN structure + k=31 neighbours +
curvature (treee/src/calculations.rs:328-373) -> map_to_u32 -> splat into a 1920x1080
depth+id buffer.  At N > 1 every rank owns one x/z tile and the step is the sharded path
(SURVEY.md section 8 row E): ghost-point all-gather -> the rank's piece of the ONE octree of
the whole cloud (common cube, Morton-cell ownership, one all-to-all, replicated top:
treee_shard_octree_build_dev) -> kNN structure over own + ghost points, curvature of the
own points -> splat of the own points, MIN all-reduce of the packed frame, resolve.
Default: weak scaling, every rank owns one 15 M-point 200 m x 200 m tile.
--total-points T: strong scaling, T / N points per rank at the same density
(BASELINE.json configs[3]: --total-points 150000000).

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      for the dominant kernel (the kNN/curvature kernel), measured live with HIP
                events on the stream the kernel runs on
  cpu_baseline  the CPU oracle (a port of the reference algorithms) timed on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

K = 31  # treee/src/calculations.rs:278
F32_MAX = float(np.finfo(np.float32).max)  # calculations.rs:279
W, H = 1920, 1080
PLOT = 200.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# algorithmic bytes per point (SURVEY.md section 8 D / BASELINE.md section 3)
# algorithmic bytes per point; the kNN structure's own order adds a key conversion (8), a second sort (68), the
# gather of the 16-byte records (4 + 16 + 16 + 4) and the bucket boxes (16)
BYTES = {"keys_sort_octree": 20 + 68 + 40, "keys_sort_octree_knn_structure": 20 + 68 + 40 + 8 + 68 + 40 + 16, "knn_curve": 340, "splat": 16}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=15_000_000, help="points per GPU")
    ap.add_argument("--total-points", type=int, default=0, help="strong scaling: points of the WHOLE cloud, cut into --gpus tiles of the same density "
                    "(configs[3]: 150000000); overrides --points / --plot")
    ap.add_argument("--transport", choices=["auto", "rccl", "host"], default="auto", help="N > 1: the library's RCCL communicator (default under nccl) or "
                    "host callbacks over the torch group")
    ap.add_argument("--stream-tiles", action="store_true", help="configs[4]'s form of a step: every rank's points arrive as LAS tiles of <= 15 M records (20 B) in pinned "
                    "host memory and are decoded into the rank's resident cloud at the START OF EVERY STEP (PCIe-inclusive; the copy of tile i + 1 under the "
                    "decode of tile i), then the step runs as usual: --gpus 8 --total-points 1005000000 --stream-tiles")
    ap.add_argument("--plot", type=float, default=PLOT, help="edge of the square tile in metres (632 with --points 150000000 keeps the density)")
    ap.add_argument("--halo", type=float, default=2.0)
    ap.add_argument("--splat-unsorted", action="store_true", help="draw the caller's cloud fragment by fragment (treee_splat_dev, rounds 1-2) instead of "
                    "sorted records through the per-workgroup LDS tile: same frame, slower")
    ap.add_argument("--splat-octree", action="store_true", help="draw the octree's Morton-sorted records (treee_splat_octree_dev, round 3) instead of the records "
                    "the kNN structure holds in Hilbert order (treee_splat_knn_dev): same frame, 0.156 vs 0.145 ms")
    ap.add_argument("--three-calls", action="store_true", help="octree + kNN structure as treee_sort_dev / treee_knn_from_octree_dev / treee_octree_complete_dev on two "
                    "contexts (rounds 3-5) instead of the one call treee_sort_build_knn_dev: same arrays")
    ap.add_argument("--no-overlap", action="store_true", help="build the octree's node table and the kNN structure one after the other on one stream")
    ap.add_argument("--cpu-sample", type=int, default=3_000_000, help="points of the CPU-baseline sample (about 10-30 s of oracle time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (ingest, per-segment kNN, eye-dome, out-of-core stream): profiling runs")
    ap.add_argument("--dry-run", action="store_true", help="bring the ranks up, build the tile plan and the library's communicator, exchange a few words through it "
                    "and print the JSON skeleton -- no GPU work (what the CPU suite runs: the launch path of --gpus N)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python3 bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): this process starts the N ranks
    itself -- fresh child processes with the environment torch.distributed.run would give them -- and never touches the GPU
    (nothing here or above calls into HIP; `import torch` does not).  Rank 0's JSON line reaches stdout through the inherited
    descriptor.  A rank that dies takes the others down instead of leaving them in a collective."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TREEE_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                for other in live:  # (exactly the processes started above)
                    other.terminate()
    return rc


class Stage:
    """HIP-event timer on torch's current stream (the stream every treee kernel runs on)."""

    def __init__(self):
        self.t = {}

    def run(self, name, fn, record):
        if not record:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.t.setdefault(name, []).append((a, b))
        return out

    def mean_ms(self):
        torch.cuda.synchronize()
        return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in self.t.items()}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # before anything touches the GPU: the ranks are fresh processes
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal knobs (NOT used by the driver): several ranks on one GPU over gloo
    backend = os.environ.get("TREEE_BENCH_BACKEND", "gloo" if args.dry_run else "nccl")
    if args.dry_run:
        return dry_run(args, rank, world, backend)
    if os.environ.get("TREEE_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import treee_amd as T
    from treee_amd import distributed as D
    from treee_amd import synth

    T.lib()
    ctx = T.default_context(dev)
    n = args.points
    plot = args.plot
    if args.total_points:
        n = args.total_points // world
        plot = (n / (15_000_000 / PLOT ** 2)) ** 0.5  # the 15 M configuration's 375 points / m^2
    origin = D.tile_origin(rank, world, plot)
    bounds = D.tile_bounds(rank, world, plot)
    pts, seg = synth.forest_cloud_torch(n, dev, plot=plot, seed=synth.SEED_FOREST + rank, origin=origin)
    ingest = None
    if args.stream_tiles:
        from treee_amd import stream

        tile_points = 15_000_000
        tiles = stream.split_into_las_tiles(pts, tile_points)
        ingest = stream.TileIngest(dev, n, tile_points)
        del pts
        ingest.ingest(tiles)
        pts = ingest.xyz[:n]  # the resident cloud: rewritten (with the same records) by every step's ingest
    cam = T.Camera3D(W / H)
    proj = cam.projection()
    span = plot * max(D.tile_grid(world))
    view = T.view_from_transform(T.rotation_x(-0.6) @ T.translation(0.0, 0.0, 100.0 + 0.9 * span))
    env = T.PointCloudEnvironment(0, 0xFFFFFFFF, 0.1)
    packed = torch.empty((H, W), dtype=torch.int64, device=dev)
    depth = torch.empty((H, W), dtype=torch.float32, device=dev)
    ids = torch.empty((H, W), dtype=torch.int32, device=dev)
    curve = torch.empty((n + n // 8,), dtype=torch.float32, device=dev)
    ctx.reserve(int(n * 1.15) * 48 + (64 << 20))
    overlap = not args.no_overlap
    # (the one-call build owns its side streams; a second context only for the three-call form and the sharded step)
    side = torch.cuda.Stream(dev) if ((overlap and args.three_calls) or world > 1) else None
    ctx2 = T.Context(dev, side) if side is not None else None
    if side is not None:
        ctx2.reserve(int(n * 1.15) * 24 + (32 << 20))  # node table, leaf ids, LOD lists and the scans' scratch of the side stream
    stages = Stage()
    side_stages = Stage()  # events recorded on the side stream (torch's current stream inside `with torch.cuda.stream(side)`)
    halo = args.halo
    state = {}
    comm = None
    if world > 1:
        # the library's own communicator: RCCL over xGMI under the nccl backend (torch only carries the unique id), host
        # callbacks over the torch group otherwise (gloo rehearsals) or when some rank cannot open RCCL.  Comm.rccl agrees
        # over the torch group before it initialises anything, so every rank takes the same branch here.
        if args.transport == "host" or (args.transport == "auto" and backend != "nccl"):
            comm = D.Comm.host()
        elif args.transport == "rccl":
            comm = D.Comm.rccl()
        else:
            comm = D.Comm.auto()

    def step(record):
        nonlocal halo
        free_state()  # previous step's handles: blocks return to the context pool, stream-ordered, no sync
        if ingest is not None:
            stages.run("ingest_decode_pcie", lambda: ingest.ingest(tiles), record)
        if world > 1:
            return step_sharded(record)
        # ONE Morton sort per step, two consumers: the kNN structure (its own Hilbert re-sort + box tree) is built on this
        # stream while the octree's node table -- ~50 small dependent launches -- is completed on a second one
        # (treee_sort_dev / treee_knn_from_octree_dev / treee_octree_complete_dev); --no-overlap builds them one after the other
        if overlap and not args.three_calls:
            # one call (treee_sort_build_knn_dev): the library owns the second stream and the order of the host's launches
            oc, tree = stages.run("keys_sort_octree_knn_structure", lambda: T.Octree.with_knn(pts, ctx=ctx), record)
        elif overlap:
            def both():
                o = T.Octree(pts, ctx=ctx, defer=True)
                t = T.KDTree.from_octree(o, ctx)
                o.complete(ctx2)
                torch.cuda.current_stream().wait_stream(side)
                return o, t

            oc, tree = stages.run("keys_sort_octree_knn_structure", both, record)
        else:
            oc = stages.run("keys_sort_octree", lambda: T.Octree(pts, ctx=ctx), record)
            tree = stages.run("knn_build", lambda: T.KDTree.from_octree(oc, ctx), record)
        cv = curve[:n]
        stages.run("knn_curve", lambda: tree.curvature_dev(K, F32_MAX, ctx, out=cv), record)
        val = stages.run("map_to_u32", lambda: T.map_to_u32_dev(cv, ctx), record)

        def splat():
            # sorted records (one 16-byte load per point), resolved in an LDS tile per workgroup before the frame's atomics: the
            # kNN structure's, in Hilbert order (--splat-octree: the octree's, in Morton order; --splat-unsorted: the caller's
            # cloud fragment by fragment as in rounds 1-2).  Same frame every way.
            if args.splat_unsorted:
                T._lib.check(T.lib().treee_splat_dev(ctx.handle, T.context.ptr(pts), T.context.ptr(val), n, T.context.ptr(view), T.context.ptr(proj),
                                                     env.scale, env.min, env.max, W, H, T.context.ptr(packed), 0, 0, None, None))
            elif args.splat_octree:
                T._lib.check(T.lib().treee_splat_octree_dev(ctx.handle, oc._h, T.context.ptr(val), T.context.ptr(view), T.context.ptr(proj),
                                                            env.scale, env.min, env.max, W, H, T.context.ptr(packed), 0, 0, None, None))
            else:
                T._lib.check(T.lib().treee_splat_knn_dev(ctx.handle, tree._h, T.context.ptr(val), T.context.ptr(view), T.context.ptr(proj),
                                                         env.scale, env.min, env.max, W, H, T.context.ptr(packed), 0, 0, None, None))
            T._lib.check(T.lib().treee_splat_resolve_dev(ctx.handle, T.context.ptr(packed), T.context.ptr(val), n, W, H, T.context.ptr(depth),
                                                         T.context.ptr(ids)))

        stages.run("splat", splat, record)
        state["oc"], state["tree"], state["curve"], state["cloud_n"] = oc, tree, cv, n
        return oc, tree

    def step_sharded(record):
        if state.get("gx") is None:
            state["gx"] = D.GhostExchanger(pts, bounds, halo, rank, world, ctx=ctx, comm=comm)
        cloud = stages.run("ghost_allgather", lambda: state["gx"].exchange(halo), record)
        m = cloud.shape[0]
        # the rank's piece of the whole cloud's octree: own points only (ghosts are neighbours, not nodes)
        oc = stages.run("sharded_octree", lambda: T.ShardedOctree(pts, comm, ctx=ctx), record)
        state["owned"] = int(oc.rank_info.count)

        # The environment culls nothing (min 0, max u32::MAX), so the packed (depth, index) frame depends on positions
        # only: it is drawn now on the second stream and min-reduced there, i.e. beside the kNN structure's build (a dozen
        # streaming launches that leave CUs free).  It does NOT overlap the kNN kernel: that kernel fills every CU (504 of
        # 512 VGPRs per SIMD lane, profiles/README.md "Concurrency experiments"), so whatever of the reduce has not run
        # when it starts runs after it -- the stage `frame_reduce_wait_resolve` is that remainder, timed, and the side
        # stream's own duration is reported as `frame_draw_reduce_side_stream`.  (One communicator, two streams: the
        # octree's all-to-all has completed -- the build ends with a synchronisation -- before the reduce is enqueued,
        # and the next step's exchange comes after `resolve` has joined the side stream.)
        def draw():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                side_stages.run("frame_draw_reduce_side_stream", lambda: (
                    T._lib.check(T.lib().treee_splat_dev(ctx2.handle, T.context.ptr(pts), None, n, T.context.ptr(view), T.context.ptr(proj),
                                                         env.scale, env.min, env.max, W, H, T.context.ptr(packed), 0, rank * n, None, None)),
                    D.frame_reduce_min(packed, comm, ctx2)), record)

        stages.run("splat_and_frame_reduce_enqueue", draw, record)
        # ONE Morton sort of the tile per step, as at N = 1: the sharded octree has the own points in Morton order of the
        # common cube; the kNN structure merges the ghosts in with its own (Hilbert) sort (treee_knn_from_shard_dev)
        tree = stages.run("knn_build", lambda: T.KDTree.from_shard(oc, cloud[n:], ctx), record)
        oc.release_local()  # (the rank's own sorted copy has served: 24 B per point back to the pool before the kNN kernel runs)
        cv = curve[:m] if m <= curve.shape[0] else torch.empty((m,), dtype=torch.float32, device=dev)
        # only the shard's own points are queried: the ghosts behind them are neighbours, not queries
        stages.run("knn_curve", lambda: tree.curvature_dev(K, F32_MAX, ctx, out=cv, n_own=n), record)
        val = stages.run("map_to_u32", lambda: T.map_to_u32_dev(cv[:n], ctx), record)

        def resolve():
            torch.cuda.current_stream().wait_stream(side)
            # (the reduced frame's indices are global: rank * n + local; this rank resolves its own winners' values and
            # leaves the others' -- the id plane of the whole frame is the union over the ranks)
            T._lib.check(T.lib().treee_splat_resolve_dev(ctx.handle, T.context.ptr(packed), T.context.ptr(val), n, W, H, T.context.ptr(depth), None))

        stages.run("frame_reduce_wait_resolve", resolve, record)
        state["oc"], state["tree"], state["curve"], state["cloud_n"] = oc, tree, cv, m
        return oc, tree

    def free_state():
        for k in ("oc", "tree"):
            if state.get(k) is not None:
                state[k].close()
                state[k] = None

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()

    # multi-GPU exactness of the unbounded-radius kNN: the halo is verified outside the timing and widened until
    # every own query's k-th neighbour lies inside it (the verdict is collective, so all ranks loop alike)
    halo_verified = world == 1
    if world > 1:
        for _ in range(6):
            step(False)
            tree = state["tree"]
            d, _, _ = tree.self_neighbors_dev(K, F32_MAX, ctx, want_index=False)
            halo_verified = bool(D.halo_is_sufficient(pts, d[:n, K - 1], bounds, halo, rank, world))
            del d
            free_state()
            if halo_verified:
                break
            halo *= 2.0
            if rank == 0:
                print(f"# halo widened to {halo} m", file=sys.stderr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        D._all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = stages.mean_ms()
    ms.update(side_stages.mean_ms())
    comm_info = comm.info() if comm is not None else None

    if rank == 0:
        total_points = n * world * args.steps
        value = total_points / elapsed / 1e6
        knn_ms = ms["knn_curve"]
        m_knn = n  # queries per launch (at N > 1 the ghosts behind the own points are neighbours only)
        achieved = BYTES["knn_curve"] * m_knn / (knn_ms * 1e-3) / 1e9
        stage_roof = {}
        for name, b in BYTES.items():
            if name not in ms:
                continue
            pts_in = m_knn if name == "knn_curve" else n
            extra = 24 * W * H if name == "splat" else 0
            gbs = (b * pts_in + extra) / (ms[name] * 1e-3) / 1e9
            stage_roof[name] = {"ms": round(ms[name], 3), "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        for name in ms:
            if name not in stage_roof:
                stage_roof[name] = {"ms": round(ms[name], 3)}
        out = {
            "metric": f"Mpoints/s octree+kNN+splat ({n * world // 1_000_000}M-point synthetic forest, {n // 1_000_000}M per GPU)",
            "value": round(value, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.total_points else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"forest {n} pts/GPU on {plot:.0f}m tile: morton radix sort + octree/LOD + kNN k=31 r=f32::MAX + curvature + splat {W}x{H}",
                "points_per_gpu": n,
                "k": K,
                "framebuffer": [W, H],
                "parallelism": f"spatial x/z tiles {D.tile_grid(world)[0]}x{D.tile_grid(world)[1]}, ghost all-gather halo {halo} m; sharded octree (common cube, "
                               f"Morton-cell ownership: rank 0 holds {n} points and owns {state.get('owned')} of the sorted order)" if world > 1 else "one GPU",
                "transport": None if comm is None else ("rccl" if comm_info[2] else "host-staged callbacks over the torch group"),
                "comm_ranks": None if comm is None else comm_info[1],  # what the library's communicator itself reports (treee_comm_info)
                "launched_by": "bench.py (spawned its own ranks)" if os.environ.get("TREEE_BENCH_SPAWNED") else ("torch.distributed.run" if world > 1 else "python"),
                "total_points": n * world,
                "halo_verified": halo_verified,
                "input": (f"streamed every step: {len(tiles)} LAS tiles of <= {tile_points} records (20 B) per rank from pinned host memory, decoded into the rank's "
                          "resident cloud (PCIe-inclusive)") if ingest is not None else "resident in HBM before the timed region",
            },
            "roofline": {
                "kernel": "k_knn (kNN k=31 + covariance/eigen -> curve)",
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                **committed_counters(m_knn),
                "algorithmic_bytes_per_point": BYTES["knn_curve"],
                "points_per_launch": m_knn,
                "kernel_ms": round(knn_ms, 3),
            },
            "stages": stage_roof,
            "end_to_end_GBps": round(((BYTES["keys_sort_octree"] + BYTES["knn_curve"] + BYTES["splat"]) * n + 24 * W * H) / (elapsed / args.steps) / 1e9 * world, 1),
        }
        if world == 1 and n == 15_000_000 and not args.no_extras:
            out["hbm_copy_GBps"] = measured_copy_gbps(dev)
            out["ingest"] = ingest_stage(n, dev, ctx)
            out["per_segment_knn"] = segment_stage(pts, seg, ctx)
            out["calculations"] = calculations_stage(pts, seg, ctx, cpu=not args.no_cpu_baseline)
            out["segmentation_host"] = segmentation_stage()
            out["eye_dome"] = eye_dome_stage(depth, ctx)
            out["out_of_core"] = out_of_core_stage(dev, n, proj)
            free_state()
            torch.cuda.empty_cache()
            out["config3_150M_one_gpu"] = whole_150m_stage()
    free_state()
    if world > 1:
        if comm is not None:
            comm.close()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the contract's two side objects at every N: rank 0 measures them once the ranks have parted (the CPU baseline is
        # 10-30 s of host work that no other rank should sit in a collective for)
        if world > 1 and not args.no_extras:
            out["hbm_copy_GBps"] = measured_copy_gbps(dev)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(out), flush=True)


def dry_run(args, rank, world, backend):
    """The launch path of `--gpus N` without a GPU (tests/test_host_logic.py runs it with two ranks): rendezvous over
    `backend`, the tile plan, the library's communicator (host callbacks over the group -- RCCL needs devices), a shard plan
    computed by every rank from all-gathered counts, and rank 0's JSON skeleton.  No hot-path work, no `value`."""
    if world > 1:
        dist.init_process_group(backend)
    import treee_amd as T
    from treee_amd import distributed as D

    T.lib()
    comm = D.Comm.host()
    info = comm.info()
    # a few words through the torch group the callbacks ride on, and the library's plan arithmetic on them
    counts = np.zeros((world, 8), np.uint32)
    counts[rank, rank % 8] = 1000 + rank
    if world > 1:
        t = torch.from_numpy(counts.view(np.int32).copy())
        dist.all_reduce(t)
        counts = t.numpy().view(np.uint32)
    plan = D.ShardPlan(counts, 1)
    owned = [int(plan.rank(r).count) for r in range(world)]
    stream_plan = None
    if args.stream_tiles:  # what `--stream-tiles` would hand every rank (pure arithmetic: no tile is built without a device)
        n_rank = (args.total_points // world) if args.total_points else args.points
        full, rest = divmod(n_rank, 15_000_000)
        stream_plan = {"points_per_rank": n_rank, "tiles_per_rank": full + (1 if rest else 0), "tile_points": [15_000_000] * full + ([rest] if rest else []),
                       "pinned_host_bytes_per_rank": n_rank * 20, "total_points": n_rank * world}
    plan.close()
    comm.close()
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": "Mpoints/s octree+kNN+splat", "value": None, "unit": "Mpoints/s", "n_gpus": world,
                          "config": {"parallelism": f"spatial x/z tiles {D.tile_grid(world)[0]}x{D.tile_grid(world)[1]}", "comm_ranks": info[1],
                                     "transport": "host-staged callbacks over the torch group", "owned_points_per_rank": owned, "stream_tiles": stream_plan,
                                     "launched_by": "bench.py (spawned its own ranks)" if os.environ.get("TREEE_BENCH_SPAWNED") else "torch.distributed.run"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def whole_150m_stage():
    """BASELINE.json configs[3]'s cloud -- 150 M points, 632 m plot, the 15 M configuration's density -- as ONE GPU runs it (the N = 1 point
    of the strong-scaling curve `--total-points 150000000 --gpus N` draws): the same step in a child process of this one (its own JSON
    line, reduced here).  Beside the main line, not part of `value`."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--points", "150000000", "--plot", "632", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"workload": d["config"]["workload"], "Mpoints_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "stages_ms": {k: v["ms"] for k, v in d["stages"].items()}, "knn_roofline_frac": d["roofline"]["frac"]}
    except Exception as e:  # (a side measurement must not take the main line down)
        return {"error": f"{type(e).__name__}: {e}"}


def measured_copy_gbps(dev):
    """Practical HBM ceiling beside the 8 TB/s spec: device-to-device copy of 1 GiB, read + write bytes."""
    a = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return round(5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def ingest_stage(n, dev, ctx):
    """Rows A8/A9 beside the step (not part of `value`): decode n LAS format-0 records (20 B, laz.rs:202-213) that are
    resident in HBM, then y-slice keys + binning (loading.rs:97-101).  Algorithmic bytes: decode reads the 12
    coordinate bytes of a record and writes 12; binning reads y (4), writes + sorts (key, index) pairs in
    ceil(bits/8) passes of 16 B, and writes the order (4)."""
    import treee_amd as T
    from treee_amd import laz

    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rec = torch.zeros((n, 5), dtype=torch.int32, device=dev)
    rec[:, 0] = torch.randint(-100_000, 100_000, (n,), generator=g, device=dev, dtype=torch.int32)
    rec[:, 1] = torch.randint(-100_000, 100_000, (n,), generator=g, device=dev, dtype=torch.int32)
    rec[:, 2] = torch.randint(0, 30_000, (n,), generator=g, device=dev, dtype=torch.int32)
    raw = rec.view(torch.uint8).reshape(-1)
    chunk = laz.Chunk(raw, 20, (0.0, 0.0, 0.0), (0.001, 0.001, 0.001), (0.0, 0.0, 15.0))
    res = {}
    for name, fn in (("decode", lambda: chunk.read(ctx)), ("slice_bin", None)):
        if fn is None:
            pts = chunk.read(ctx)
            fn = lambda: laz.slice_bin(pts, ctx)  # noqa: E731
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 3
    layers = out[2].shape[0] - 1
    passes = max(1, (max(1, (layers - 1).bit_length()) + 7) // 8)
    b_dec, b_bin = 24, 4 + 8 + 16 * passes + 4
    return {
        "workload": f"{n} LAS fmt-0 records (20 B) resident in HBM -> xyz f32 -> {layers} y-slices",
        "decode_ms": round(res["decode"], 3),
        "decode_GBps": round(b_dec * n / (res["decode"] * 1e-3) / 1e9, 1),
        "decode_frac": round(b_dec * n / (res["decode"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "slice_bin_ms": round(res["slice_bin"], 3),
        "slice_bin_GBps": round(b_bin * n / (res["slice_bin"] * 1e-3) / 1e9, 1),
        "slice_bin_frac": round(b_bin * n / (res["slice_bin"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "Mpoints_per_s": round(n / ((res["decode"] + res["slice_bin"]) * 1e-3) / 1e6, 1),
    }


def eye_dome_stage(depth, ctx):
    """the eye-dome pass (render/src/eye_dome.wgsl:37-67) over the step's depth frame, and over an
    8K frame where the launch is long enough to read a bandwidth from (4 B read + 4 B written per pixel)."""
    import treee_amd as T

    res = {}
    big = depth.repeat(4, 4).contiguous()  # 7680 x 4320
    for name, d in (("1080p", depth), ("8k", big)):
        T.eye_dome_dev(d, 0.7, ctx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            T.eye_dome_dev(d, 0.7, ctx)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[name] = {"ms": round(ms, 4), "achieved_GBps": round(8 * d.numel() / (ms * 1e-3) / 1e9, 1), "frac": round(8 * d.numel() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return res


def segment_stage(pts, seg, ctx):
    """The kNN + curvature loop the way the reference calls it -- one KD-tree per segment (calculations.rs:97-118,
    329) -- for all segments of the same cloud in one pass (treee_knn_build_segments_dev + treee_curvature_dev).
    Beside the step, not part of `value`."""
    import treee_amd as T

    n = pts.shape[0]
    nseg = int(seg.max().item()) + 1
    out = torch.empty((n,), dtype=torch.float32, device=pts.device)
    res = {}
    tree = None
    for name in ("build", "curve"):
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "build":
                if tree is not None:
                    tree.close()
                tree = T.KDTree.new_segments(pts, seg, nseg, ctx)
            else:
                tree.curvature_dev(K, F32_MAX, ctx, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[name] = min(ts[1:])
    tree.close()
    return {
        "workload": f"{n} points in {nseg} segments (ground + trees), k={K}, r=f32::MAX, neighbours inside the segment only",
        "build_ms": round(res["build"], 3),
        "knn_curve_ms": round(res["curve"], 3),
        "Mpoints_per_s": round(n / ((res["build"] + res["curve"]) * 1e-3) / 1e6, 1),
    }


def segmentation_stage(sample=1_500_000):
    """SURVEY 8 F1, beside the step: the Segmenting phase (segmenting.rs:110-235) is host code in the reference and in the
    library (serial per slice; slices over a thread pool).  Timed on a forest patch of the step's density."""
    import treee_amd as T
    from treee_amd import synth

    plot = PLOT * (sample / 15_000_000) ** 0.5
    p, _ = synth.forest_cloud_torch(sample, "cuda", plot=plot)
    host = p.cpu().numpy()
    t0 = time.perf_counter()
    sg, nseg = T.segment_slices(host, seed=1)
    dt = time.perf_counter() - t0
    return {"workload": f"{sample}-point forest patch ({plot:.0f} m), max_distance 0.75: slices -> hulls -> centroid hand-off -> Voronoi cells -> ids",
            "seconds": round(dt, 3), "Mpoints_per_s": round(sample / dt / 1e6, 3), "segments": int(nseg) - 1, "host_threads": _host_threads(),
            "assigned_fraction": round(float((sg > 0).mean()), 4)}


def calculations_stage(pts, seg, ctx, cpu=True):
    """SURVEY 8 F4, beside the step: the Calculations phase the curvature loop sits in -- SegmentData::new +
    SegmentInformation::new / update (calculations.rs:133-327) and DisplayData::new (interactive/mod.rs:1002-1039) for
    every segment of the same cloud (treee_segments_calculate_dev, treee_display_values_dev).  The CPU figure is the
    oracle on the first segments of the cloud holding ~1.5 M points (the ground segment is among them), one thread per
    segment the way Calculations::new spreads them over rayon (calculations.rs:97) -- wall time of the longest."""
    import treee_amd as T

    n = pts.shape[0]
    nseg = int(seg.max().item()) + 1
    curve = torch.rand((n,), dtype=torch.float32, device=pts.device)
    ts, tv = [], []
    for _ in range(3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        r = T.segments_calculate(pts, seg, nseg, ctx=ctx)
        e1.record()
        T.display_values_dev(seg, nseg, r.classification, curve, r.expansion, r.height, ctx=ctx)
        e2.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        tv.append(e1.elapsed_time(e2))
    out = {
        "workload": f"{n} points in {nseg} segments: min/max, slice hulls (0.1 m), trunk/crown heights, classification, expansion, height; then the 4 display values",
        "calculate_ms": round(min(ts[1:]), 3),
        "display_values_ms": round(min(tv[1:]), 3),
        "Mpoints_per_s": round(n / (min(ts[1:]) * 1e-3) / 1e6, 1),
        # against the HBM roofline on the phase's algorithmic bytes (16 B read -- xyz + segment id -- and 12 B
        # written -- class, expansion, height -- per point; display values: 20 B read, 16 B written).  Small by construction: the
        # phase is the reference's SEQUENTIAL hull insertion per slice, bound by instruction issue and dependent latency
        "calculate_GBps": round(28 * n / (min(ts[1:]) * 1e-3) / 1e9, 1),
        "calculate_frac": round(28 * n / (min(ts[1:]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "display_values_GBps": round(36 * n / (min(tv[1:]) * 1e-3) / 1e9, 1),
        "display_values_frac": round(36 * n / (min(tv[1:]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
    }
    if cpu:
        from oracle import oracle as O

        O.build()
        sg = seg.cpu().numpy().astype(np.uint32)
        host = pts.cpu().numpy()
        counts = np.bincount(sg, minlength=nseg)
        last = int(np.searchsorted(np.cumsum(counts), 1_500_000)) + 1
        total, longest = 0, 0.0
        t_all = time.perf_counter()
        for s_id in range(min(last, nseg)):
            q = host[sg == s_id]
            t0 = time.perf_counter()
            mn, mx, info, cls = O.segment_data(q)
            O.segment_update(q, cls, mn[1], mx[1], info)
            longest = max(longest, time.perf_counter() - t0)
            total += len(q)
        out["cpu_oracle"] = {"points": total, "segments": min(last, nseg), "one_thread_s": round(time.perf_counter() - t_all, 3),
                             "longest_segment_s": round(longest, 3), "kind": "port"}
    return out


def out_of_core_stage(dev, n, proj, tiles=67):
    """BASELINE.json configs[4] on one GPU: 67 forest tiles = 1.005 G points streamed from pinned host memory as
    uncompressed LAS format-0 records (20 B) through decode -> Morton sort -> octree/LOD -> splat (and, second line,
    with kNN k=31 + curvature per tile), PCIe copies overlapped with compute on a second stream
    (treee_amd/stream.py).  PCIe-inclusive; beside the step, not part of `value`."""
    import treee_amd as T
    from treee_amd import stream, synth

    base = []
    for i in range(3):
        p, _ = synth.forest_cloud_torch(n, dev, plot=PLOT, seed=synth.SEED_FOREST + 100 + i)
        base.append(stream.las_tile_from_cloud(p))
        del p
    view = T.view_from_transform(T.rotation_x(-0.6) @ T.translation(0.0, 0.0, 100.0 + 0.9 * 9 * PLOT))
    env = T.PointCloudEnvironment(0, 0xFFFFFFFF, 0.1)

    def gen(count):
        for i in range(count):
            b = base[i % len(base)]  # placed on a 9 x 8 grid through the LAS header offset
            yield stream.Tile(b.records, b.n, b.stride, b.scale, ((i % 9) * PLOT, (i // 9) * PLOT, 0.0), (4 * PLOT, 3.5 * PLOT, 0.0))

    res = {"workload": f"{tiles} LAS tiles x {n} points from pinned host memory, 20 B records",
           "per_tile_mode": "octree_splat / octree_knn_splat: PER TILE, NO GHOST POINTS -- one octree per tile (its own cube), kNN inside a tile only "
                            "(a point near a tile border sees one side); whole_cloud is the one-cloud result"}
    for name, k in (("octree_splat", 0), ("octree_knn_splat", K)):
        ts = stream.TileStream(dev, n, width=W, height=H, knn_k=k)
        ts.run(gen(3), view, proj, env)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ts.run(gen(tiles), view, proj, env)
        dt = time.perf_counter() - t0
        ts.close()
        res[name] = {"points": r["points"], "seconds": round(dt, 3), "Mpoints_per_s": round(r["points"] / dt / 1e6, 1),
                     "pcie_GBps": round(r["points"] * 20 / dt / 1e9, 1)}
    # configs[4] as ONE cloud: the same tiles decoded into one resident buffer (12 B x 1.005 G = 12 GB of HBM), then one sort /
    # octree / LOD with one cube and global numbering, the kNN structure from that sort, curvature with neighbours across tile
    # borders, one splat (stream.WholeCloudStream).  PCIe-inclusive wall time of the second of two runs.
    try:
        torch.cuda.empty_cache()
        ws = stream.WholeCloudStream(dev, n * tiles, n, width=W, height=H, knn_k=K)
        # two full runs: the first touches 130 GB of freshly allocated HBM (the resident cloud, both sorted copies, the node
        # table ...), which is several times slower than reusing it -- reported as first_run_seconds; `seconds` is the second
        # run, the state a resident service is in (the library's pool keeps the arrays: csrc/common.h pool_idle_limit)
        t0 = time.perf_counter()
        ws.run(gen(tiles), view, proj, env)
        first = time.perf_counter() - t0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ws.run(gen(tiles), view, proj, env)
        dt = time.perf_counter() - t0
        st = r["stages_ms"]
        b_struct, b_knn = BYTES["keys_sort_octree_knn_structure"], BYTES["knn_curve"]
        res["whole_cloud"] = {
            "mode": r["mode"], "points": r["points"], "seconds": round(dt, 3), "Mpoints_per_s": round(r["points"] / dt / 1e6, 1), "first_run_seconds": round(first, 3),
            "pcie_GBps_during_ingest": round(r["points"] * 20 / (st["ingest_decode"] * 1e-3) / 1e9, 1),
            "stages_ms": {k: round(v, 2) for k, v in st.items()}, "octree_nodes": r["nodes"], "lod_entries": r["lod"], "cube_size_m": round(r["size"], 1),
            "knn_curve_frac_of_hbm": round(b_knn * r["points"] / (st["knn_curve"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "structure_frac_of_hbm": round(b_struct * r["points"] / (st["sort_octree_knn_structure"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "hbm_in_use_GB": round((torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 1e9, 1),  # cloud + both structures + workspace, handles alive
        }
        ws.close()
        del ws
        torch.cuda.empty_cache()
    except Exception as e:  # (a side measurement must not take the main line down)
        res["whole_cloud"] = {"error": f"{type(e).__name__}: {e}"}
    # the same stream from LAZ tiles (configs[4] as BASELINE.json words it): decompressed on the host's threads, one tile
    # ahead of the GPU -- what bounds it is the host, as in the reference (laz.rs:134-160)
    import copy

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import laz_fixture  # test infrastructure: the LAZ compressor (the reference never writes LAZ; tests/cpp/laz_encode_fixture.cpp)

    t0 = time.perf_counter()
    p, _ = synth.forest_cloud_torch(n, dev, plot=PLOT, seed=synth.SEED_FOREST + 100)
    ztile = laz_fixture.laz_tile_from_cloud(p)
    del p
    enc_s = time.perf_counter() - t0

    def zgen(count):
        for i in range(count):
            z = copy.copy(ztile)  # (placed through the LAS header offset like the tiles above; the image is shared)
            z.offset, z.center = np.asarray(((i % 9) * PLOT, (i // 9) * PLOT, 0.0)), np.asarray((4 * PLOT, 3.5 * PLOT, 0.0))
            yield z

    ztiles = min(tiles, 24)
    laz_threads = _host_threads()
    res["laz"] = {"workload": f"{ztiles} LAZ tiles x {n} points (format 0, 50 000-point chunks), {len(ztile.src.data) / n:.2f} B/point compressed",
                  "host_threads": laz_threads, "encode_seconds_per_tile": round(enc_s, 2)}
    for name, k in (("octree_splat", 0), ("octree_knn_splat", K)):
        ts = stream.TileStream(dev, n, width=W, height=H, knn_k=k)
        ts.laz_threads = 0  # (the library's own choice, reported above)
        ts.run(zgen(2), view, proj, env)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ts.run(zgen(ztiles), view, proj, env)
        dt = time.perf_counter() - t0
        ts.close()
        res["laz"][name] = {"points": r["points"], "seconds": round(dt, 3), "Mpoints_per_s": round(r["points"] / dt / 1e6, 1)}
    return res


def _host_threads():
    """what the library's host-side phases start by themselves (treee::host_threads): the hardware's threads, or the cgroup's
    CPU quota when that is smaller"""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def _latest_profile(pattern):
    """newest committed summary profiles/rNN_vM_<pattern> (by round, then version)"""
    import glob
    import re

    best, best_key = None, None
    for path in glob.glob(os.path.join(ROOT, "profiles", f"r*_{pattern}")):
        m = re.match(r"r(\d+)(?:_v(\d+))?_", os.path.basename(path))
        if not m:
            continue
        key = (int(m.group(1)), int(m.group(2) or 0))
        if best_key is None or key > best_key:
            best, best_key = path, key
    return best


def committed_counters(points):
    """`traffic` (HBM bytes per launch: 2 x FETCH_SIZE raw -- the gfx950 correction of MI355X_MICROARCH.md -- + WRITE_SIZE) and
    `valu_busy` (share of SIMD issue cycles that issue a VALU instruction: waves per SIMD x SQ_ACTIVE_INST_VALU /
    SQ_WAVE_CYCLES) of the dominant kernel.  PMC passes cannot run inside this process (rocprofv3 wraps the command), so both
    come from the newest COMMITTED summaries (profiles/collect.sh, profiles/knn_pmc.sh) -- and only if those are not older
    than the kernel: each file records the sha256 of csrc/knn.hip it was measured with, and a summary of another kernel
    source (or of another cloud size) gives null.  The files used are named beside the numbers."""
    import hashlib

    out = {"traffic": None, "valu_busy": None, "counters_from": None}
    try:
        sha = hashlib.sha256(open(os.path.join(ROOT, "treee_amd", "csrc", "knn.hip"), "rb").read()).hexdigest()[:16]
        used = {}
        tp = _latest_profile("traffic.json")
        t = json.load(open(tp))
        if t.get("knn_hip_sha16") == sha and t["points_per_launch"] == points:
            out["traffic"] = int(2 * t["fetch_size_bytes_raw"] + t["write_size_bytes"])
            used["traffic"] = os.path.basename(tp)
        cp = _latest_profile("knn_sq_pmc.json")
        c = json.load(open(cp))
        if c.get("knn_hip_sha16") == sha and c.get("points_per_launch", 15_000_000) == points:
            out["valu_busy"] = round(float(c.get("waves_per_simd", 3)) * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 3)
            used["valu_busy"] = os.path.basename(cp)
        why = (f"were not measured with this csrc/knn.hip ({sha})" if t.get("knn_hip_sha16") != sha or c.get("knn_hip_sha16") != sha
               else f"describe launches of {t['points_per_launch']} points, this run's have {points}")
        out["counters_from"] = used or f"stale: newest committed summaries ({os.path.basename(tp)}, {os.path.basename(cp)}) {why}"
    except Exception as e:
        out["counters_from"] = f"unavailable: {type(e).__name__}"
    return out


def cpu_baseline(sample):
    """The CPU oracle (port of the reference algorithms) on a forest patch of the same density,
    timed on this host: as-reference = 1 thread (calculations.rs:332 is a sequential iterator)."""
    from oracle import oracle as O
    from treee_amd import synth

    O.build()
    plot = PLOT * (sample / 15_000_000) ** 0.5
    d_pts, d_seg = synth.forest_cloud_torch(sample, "cuda" if torch.cuda.is_available() else "cpu", plot=plot)
    pts, seg = d_pts.cpu().numpy(), d_seg.cpu().numpy().astype(np.uint32)
    cores = _host_threads()  # (the CPUs this process may really use: a one-GPU box shows 256 and grants 16)
    t0 = time.perf_counter()
    mn = pts.min(0)
    keys = O.morton_keys(pts, mn, float((pts.max(0) - mn).max()))
    sk, perm = O.sort_keys(keys)
    O.octree_build(sk, 256, 256)
    t1 = time.perf_counter()
    curve = O.curvature(pts, K, F32_MAX, threads=1)
    t2 = time.perf_counter()
    val = np.array([O.map_to_u32(v) for v in curve[:1]], np.uint32)  # noqa: F841 (elementwise, negligible)
    view = np.eye(4, dtype=np.float32)
    view[2, 3] = -(100.0)
    O.splat(pts, seg, np.ascontiguousarray(view.T).reshape(16), O.perspective(W / H, 45.0 * 2 * np.pi / 360.0, 0.1, 1e4), 0.1, 0, 0xFFFFFFFF, W, H)
    t3 = time.perf_counter()
    O.curvature(pts, K, F32_MAX, threads=cores)
    t4 = time.perf_counter()
    one = sample / (t3 - t0) / 1e6
    return {
        "value": round(one, 4),
        "unit": "Mpoints/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample}-point forest patch at the 15M config's density (375 pts/m^2); sort+octree {t1 - t0:.2f}s, KDTree build + 31-NN + curvature {t2 - t1:.2f}s (1 thread, as the reference), software splat {t3 - t2:.2f}s",
        "knn_curve_all_cores": {"cores": cores, "Mpoints_per_s": round(sample / (t4 - t3) / 1e6, 4)},
    }


if __name__ == "__main__":
    main()
