"""GPU (-m gpu): the PRODUCT's sharded forward, two ranks sharing cuda:0.

The GPU box has one MI355X, so both ranks run on the same device and the process group is gloo (it reduces and
gathers HIP tensors through host staging; RCCL refuses two ranks on one GPU).  What runs is exactly what runs over
RCCL on a node: MVSNet.set_view_shard -> per-rank FeatureNet on the local views, K1 on the local source views,
all_reduce(SUM) of the partial similarity volumes, and -- with shard_rows -- H-slab regularisation + all-gather of the
regression outputs.  Asserted against the same process's unsharded forward."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, H=512, W=160, V=4, ndepths=(8, 8, 8), use_wino=True):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from dmvsnet_amd import MVSNet, ops, shard_source_views, synth
        ops.use_wino = use_wino

        # (default case: stage-1 volume 128 rows -- slabs of 64 + 32 halo rows are real cuts)
        ndepths, ratios = list(ndepths), [3, 2, 1]
        net = MVSNet(ndepths, ratios, verbose=False)
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 5))
        net = net.to("cuda:0")
        net.return_prob_volume = False
        imgs, proj, dv = synth.synth_inputs(H, W, V, 5)
        args = (imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
        full = net(*args)
        full = {k: full[k].clone() for k in ("depth", "photometric_confidence", "depth_sub_plus", "depth_values_c")}

        def rel(a, b):
            return ((a - b).abs().mean() / b.abs().mean()).item()

        res = {"rank": rank, "views": shard_source_views(V, world, rank)}
        net.set_view_shard(dist.group.WORLD, rank, world)                     # v1: view shard + all_reduce
        out = net(*args)
        res["v1"] = {k: rel(out[k], full[k]) for k in full}
        v1_depth = out["depth"].clone()
        net.set_view_shard(dist.group.WORLD, rank, world, shard_rows=True)    # v2: + H-slab regularisation, fed by
        out = net(*args)                                                      # reduce_scatter + halo exchange
        res["v2"] = {k: rel(out[k], full[k]) for k in full}
        res["v2_equals_v1"] = bool(torch.equal(out["depth"], v1_depth))
        res["v2_vs_v1"] = rel(out["depth"], v1_depth)
        v2_depth = out["depth"].clone()
        net.set_view_shard(dist.group.WORLD, rank, world, shard_rows=True, row_collective="all_reduce")
        out = net(*args)
        res["v2_allreduce_close"] = rel(out["depth"], v2_depth)
        res["v2_allreduce_equal"] = bool(torch.equal(out["depth"], v2_depth))
        res["slab_rows"] = [MVSNet.row_slabs((H // 4) << s, world)[0][rank] for s in range(3)]
        res["shapes"] = {k: tuple(out[k].shape) == tuple(full[k].shape) for k in full}
        torch.cuda.synchronize()
        q.put(res)
    except Exception as e:   # noqa: BLE001 -- report to the parent instead of hanging its queue
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc() + repr(e)})
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_product_view_shard_two_ranks_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in range(world)]
    for p in procs:
        p.join(60)
    for r in res:
        assert "error" not in r, r.get("error")
    res.sort(key=lambda r: r["rank"])
    assert res[0]["views"] == [1, 3] and res[1]["views"] == [2]
    for r in res:
        assert all(r["shapes"].values()), r
        # only the order of the view sum differs; the confidence is a steep function of the spread of
        # the four regressed depths (sigmoid(interval / std)), so its relative error is ~10x that of the depths
        for k, v in r["v1"].items():
            assert v < (1e-4 if k == "photometric_confidence" else 1e-6), ("view shard", k, v)
        for k, v in r["v2"].items():
            assert v < (1e-4 if k == "photometric_confidence" else 1e-6), ("view shard + row slabs", k, v)
        # the direct-form kernels reproduce the replicated regularisation bit for bit; the Winograd layers do so only where
        # a slab's 2x2 output tiling coincides with the full volume's (even slab offsets at every U-Net level)
        assert r["v2_equals_v1"] or r["v2_vs_v1"] < 1e-6, ("H-slab regularisation vs replicated", r["v2_vs_v1"])
        assert r["v2_allreduce_equal"], "two ranks: a + b in either collective is the same sum"


@pytest.mark.timeout(900)
def test_product_view_shard_four_ranks_eleven_views():
    """BASELINE configs[2] as it is sharded over 4 GPUs (11 views -> 3/3/2/2 source views per rank), at a quarter of the
    linear size, four ranks on cuda:0: v1 (all-reduce), v2 with both row collectives; the stage-1 volume has 72 rows =
    3 slabs of 24, so rank 3 owns an EMPTY slab there (and a short one at stage 2)."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 288, 416, 11, (16, 8, 8))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=780) for _ in range(world)]
    for p in procs:
        p.join(60)
    for r in res:
        assert "error" not in r, r.get("error")
    res.sort(key=lambda r: r["rank"])
    assert [r["views"] for r in res] == [[1, 5, 9], [2, 6, 10], [3, 7], [4, 8]]
    assert res[3]["slab_rows"][0][1] - res[3]["slab_rows"][0][0] == 0, res[3]["slab_rows"]   # the empty slab
    for r in res:
        assert all(r["shapes"].values()), r
        for mode in ("v1", "v2"):
            for k, v in r[mode].items():
                assert v < (1e-4 if k == "photometric_confidence" else 1e-6), (mode, k, v)
        # four partial sums: the collectives may associate them differently (fp32 re-association only)
        assert r["v2_allreduce_close"] < 1e-6, r


def _run(world, args, timeout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(60)
    for r in res:
        assert "error" not in r, r.get("error")
    res.sort(key=lambda r: r["rank"])
    return res


@pytest.mark.timeout(600)
def test_row_slabs_direct_form_bit_identical():
    """What protects MVSNet.ROW_HALO: with the direct-form kernels (ops.use_wino = False) the H-slab
    regularisation must reproduce the replicated run BIT FOR BIT -- a halo one row short, or a slab off the 8-row grid,
    changes bits here, where the Winograd default only promises rel < 1e-6."""
    for r in _run(2, (512, 160, 4, (8, 8, 8), False), 480):
        assert r["v2_equals_v1"], ("direct-form H-slab regularisation differs from the replicated run", r["v2_vs_v1"])
        assert r["v2_allreduce_equal"]
        for k, v in r["v2"].items():
            assert v < (1e-4 if k == "photometric_confidence" else 1e-6), (k, v)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("V", [11, 5])
def test_product_view_shard_eight_ranks(V):
    """BASELINE configs[3] / [4] name 8 GPUs.  Eight ranks on cuda:0 (gloo), c3 at a quarter of the linear size:
    V = 11 -> 2/2/1/1/1/1/1/1 source views per rank; V = 5 -> ranks 4..7 own NO source view (the nsrc == 0 branch of
    ops.warp_corr, an empty projection slice, FeatureNet on the reference view alone).  v1 (all-reduce) and v2 (row slabs
    of 16 / 24 / 40 rows: shorter than the halo, several neighbours per side; three ranks with an empty stage-1 slab)."""
    res = _run(8, (288, 416, V, (16, 8, 8)), 1300)
    views = [r["views"] for r in res]
    if V == 11:
        assert [len(v) for v in views] == [2, 2, 1, 1, 1, 1, 1, 1], views
    else:
        assert views == [[1], [2], [3], [4], [], [], [], []], views
    assert sum(1 for r in res if r["slab_rows"][0][1] == r["slab_rows"][0][0]) == 3     # 72 rows = 4 x 16 + 8
    for r in res:
        assert all(r["shapes"].values()), r
        for mode in ("v1", "v2"):
            for k, v in r[mode].items():
                assert v < (1e-4 if k == "photometric_confidence" else 2e-6), (r["rank"], mode, k, v)
        assert r["v2_allreduce_close"] < 2e-6, r


def _worker_hybrid(rank, world, port, q, vg, H, W, V, ndepths):
    """Hybrid ("config 2 at 8 GPUs must combine with (1)"): world / vg view groups of vg ranks; every group works
    on ITS OWN reference view (seed = group index), its ranks share the depth map (v2: row slabs over the sub-group)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from dmvsnet_amd import MVSNet, shard_source_views, synth
        groups = [dist.new_group(ranks=list(range(g * vg, (g + 1) * vg))) for g in range(world // vg)]
        gi, gr = rank // vg, rank % vg
        net = MVSNet(list(ndepths), [3, 2, 1], verbose=False)
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), 5))
        net = net.to("cuda:0")
        net.return_prob_volume = False
        imgs, proj, dv = synth.synth_inputs(H, W, V, 20 + gi)       # a different reference view per group
        args = (imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
        full = net(*args)["depth"].clone()
        net.set_view_shard(groups[gi], gr, vg, shard_rows=True)
        out = net(*args)["depth"]
        torch.cuda.synchronize()
        q.put({"rank": rank, "group": gi, "views": shard_source_views(V, vg, gr),
               "rel": ((out - full).abs().mean() / full.abs().mean()).item(),
               "sig": full[0, 5::37, 3::41].flatten()[:32].cpu().tolist()})   # a few pixels of the group's own depth map
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc() + repr(e)})
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_hybrid_two_view_groups_of_four_ranks_five_views():
    """5 views on 8 ranks as 2 view groups x 4 -- every rank owns exactly one source view (the plain 8-way
    shard leaves ranks 4-7 empty), the two groups run different reference views at the same time, and each group's depth map
    equals its own unsharded forward (sub-group reduce_scatter + halo exchange with global-rank peers)."""
    world, vg = 8, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hybrid, args=(r, world, port, q, vg, 288, 416, 5, (16, 8, 8))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1300) for _ in range(world)]
    for p in procs:
        p.join(60)
    for r in res:
        assert "error" not in r, r.get("error")
    res.sort(key=lambda r: r["rank"])
    assert [r["views"] for r in res] == [[1], [2], [3], [4]] * 2
    assert all(len(r["views"]) >= 1 for r in res)
    for r in res:
        assert r["rel"] < 2e-6, r
    # the two groups really worked on different depth maps (different reference views), the ranks of a group on the same one
    assert max(abs(a - b) for a, b in zip(res[0]["sig"], res[4]["sig"])) > 1.0
    assert res[0]["sig"] == res[3]["sig"] and res[4]["sig"] == res[7]["sig"]


@pytest.mark.timeout(900)
def test_bench_hybrid_view_groups():
    """bench.py --gpus 8 --mode view-shard-rows --view-group 4: two depth maps in flight per step, weak scaling over the groups."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--share-gpu",
           "--mode", "view-shard-rows", "--view-group", "4", "--config", "c3_small", "--steps", "2", "--warmup", "1", "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["n_ranks"] == 8 and res["view_group"] == 4 and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"] + 1e-3     # 2 groups x steps / time
    assert "2 view groups" in res["config"]["parallelism"]


@pytest.mark.timeout(900)
def test_bench_eight_ranks_view_shard_rows():
    """bench.py --gpus 8 in the latency mode v2, the eight ranks sharing cuda:0 over gloo: the launcher, the rank
    bookkeeping (n_ranks counted by a collective) and the single JSON line."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--share-gpu",
           "--mode", "view-shard-rows", "--config", "c3_small", "--steps", "2", "--warmup", "1", "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 8 and res["n_ranks"] == 8 and res["scaling"] == "strong"
    assert res["latency_mode"]["value"] > 0 and res["throughput_mode"]["value"] > 0


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world,config,vg", [(2, "c3_small", 2), (8, "c2_small", 4)])
def test_bench_replicas_under_torchrun(world, config, vg):
    """The driver's scaling command line -- `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, NO --mode
    -- with the ranks sharing cuda:0 over gloo: rendezvous from the environment, barrier + max-over-ranks timing, ONE JSON line
    from rank 0 whose `value` is the replicas rate (weak scaling, no data-path collective) AND whose `latency_mode` is
    north_star's partition: the hybrid view shard (5 views on 8 ranks = 2 view groups x 4), with the bytes every collective
    moved per rank and the sharded depth against the unsharded forward, both measured in the run."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--dist-backend", "gloo", "--share-gpu", "--config", config] + (["--no-kernel-timing"] if world > 2 else [])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1400)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == world and res["n_ranks"] == world and res["scaling"] == "weak" and res["steps"] == 3 and res["warmup"] == 1
    assert res["value"] > 0 and abs(res["value"] - world * 1000.0 / res["ms_per_step"]) < 1e-6 * res["value"] + 1e-3
    assert res["vs_baseline"] is None and ("roofline" in res or world > 2)
    # both sub-records: the replicas rate is `value`; the latency mode is the view shard with collectives on the data path
    thr, lat = res["throughput_mode"], res["latency_mode"]
    assert abs(thr["value"] - res["value"]) < 1e-9 * res["value"]
    assert lat["mode"] == "view-shard-rows" and lat["view_group"] == vg == res["view_group"] and lat["view_groups"] == world // vg
    assert lat["value"] > 0 and abs(lat["value"] - (world // vg) * 1000.0 / lat["ms_per_map"]) < 1e-6 * lat["value"] + 1e-3
    comm = lat["collectives_per_map_per_rank"]
    assert comm["reduce_scatter"]["calls_per_map"] == 6 and comm["all_gather"]["calls_per_map"] == 6   # main + refine pass of 3 stages
    assert comm["reduce_scatter"]["bytes_received_per_rank"] > 0 and comm["halo_p2p"]["calls_per_map"] == 6
    # a rank receives 1 / vg of what it hands to the reduce_scatter (padded slabs)
    assert comm["reduce_scatter"]["bytes_sent_per_rank"] == vg * comm["reduce_scatter"]["bytes_received_per_rank"]
    assert 0.0 <= lat["depth_rel_vs_unsharded"] < lat["depth_rel_vs_unsharded_bound"] == 2e-6
    assert "legs_s" in res and res["legs_s"]["latency_mode"] > 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fault", ["raise:1", "hang:1", "raise:0"])
def test_bench_line_survives_a_failing_latency_leg(fault):
    """The driver's multi-GPU command must leave its record even if the secondary leg dies: `latency_mode` (the hybrid view
    shard, never run over RCCL / xGMI before the driver's first multi-GPU node) runs LAST and guarded.  A rank that raises
    inside it, or one that never comes back (a collective that does not return), must cost nothing but the sub-record: ONE
    JSON line from rank 0 with the replicas `value`, `latency_mode.error`, exit code 0."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--dist-backend", "gloo", "--share-gpu", "--config", "c3_small", "--no-kernel-timing",
           "--inject-latency-fault", fault, "--latency-budget-s", "25"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["n_ranks"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert abs(res["throughput_mode"]["value"] - res["value"]) < 1e-9 * res["value"]
    lat = res["latency_mode"]
    assert lat["value"] is None and lat["error"] and lat["mode"] == "view-shard-rows" and "collectives_per_map_per_rank" not in lat
