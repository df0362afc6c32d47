"""CPU, world_size 2 over gloo: the multi-GPU sharding of bench.py (row-cyclic PointSource grids + one all-reduce of
the emissivity histogram) is correct by construction: the union of the rank grids is exactly the refined global
grid and the all-reduced histogram equals the single-process one.  The per-rank engine here is the oracle (no GPU
in this container); on the GPU box the same ranks call libkrtrace and all-reduce over RCCL."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_cases as gc
import oracle_lib as ol
from raytrace_cpu_amd import capi

sys.path.insert(0, gc.ROOT)
import bench  # noqa: E402

D = 0.08       # base grid spacing: 25 x 25 rays per rank
NR = 30


def _hist(spec, n_primary):
    o = ol.oracle()
    rays = ol.oracle_pointsource(spec)
    o.kro_redshift_start_f64(bench.SPIN, 0.0, 0, 0, ol.ptr(rays), len(rays))
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max = capi.RK4, bench.R_MAX
    out, st = ol.oracle_trace(p, rays, nthreads=2)
    o.kro_range_phi_f64(-np.pi, np.pi, ol.ptr(out), len(out))
    o.kro_redshift_f64(bench.SPIN, -1.0, 0, 0, 0, ol.ptr(out), len(out))
    b = bench.emis_bins(capi, o.kro_kerr_isco(bench.SPIN, 1), n_primary)
    b.nr = NR
    b.dr = float(np.exp(np.log(bench.R_DISC / b.r_min) / NR))
    cnt = np.zeros(NR, dtype=np.int64)
    flux, emis, sg, stt = (np.zeros(NR) for _ in range(4))
    dc = C.c_int64()
    o.kro_reduce_emissivity_f64(C.byref(b), ol.ptr(out), len(out), ol.ptr(cnt), ol.ptr(flux), ol.ptr(emis), ol.ptr(sg), ol.ptr(stt), C.byref(dc))
    h = np.concatenate([cnt.astype(np.float64), flux, emis, sg, stt, [float(dc.value)]])
    return h, out, st


def _n_primary(world):
    full = bench.make_spec(capi, D, refine=world)
    return int(((full.cosalphamax - full.cosalpha0) / full.dcosalpha) * ((full.betamax - full.beta0) / full.dbeta)), full


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_primary, _ = _n_primary(world)
    h, out, st = _hist(bench.make_spec(capi, D, rank, world), n_primary)
    t = torch.from_numpy(h.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)          # the path's one exchange
    tot = torch.tensor([float(st["rays_traced"]), float(st["steps_total"])], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    live = out["steps"] != -1
    q.put((rank, t.numpy().copy(), tot.numpy().copy(), np.stack([out["alpha"][live], out["beta"][live]], 1)))
    dist.barrier()
    dist.destroy_process_group()


def test_row_cyclic_shards_reduce_to_the_global_histogram():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    n_primary, full = _n_primary(world)
    h_full, out_full, st_full = _hist(full, n_primary)
    # (1) every rank holds the same reduced histogram, (2) it equals the single-process histogram of the global grid
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][1][:NR], h_full[:NR])                 # counts: exact
    np.testing.assert_allclose(res[0][1], h_full, rtol=1e-12)                   # sums: order of addition only
    # (3) the shards partition the global ray set: same (cos alpha, beta) pairs, no overlap
    live = out_full["steps"] != -1
    glob = np.stack([out_full["alpha"][live], out_full["beta"][live]], 1)
    union = np.concatenate([r[3] for r in res])
    assert len(union) == len(glob) == int(res[0][2][0]) == st_full["rays_traced"]
    key = lambda a: np.lexsort((a[:, 1], a[:, 0]))
    np.testing.assert_allclose(union[key(union)], glob[key(glob)], rtol=0, atol=1e-12)
    assert int(res[0][2][1]) == st_full["steps_total"]


# ---- image plane: ray-cyclic shards, all-reduce of the seven planes (bench.py --workload imageplane) ---------------
IMG = 8


def _image_planes(rays):
    o = ol.oracle()
    spin = bench.SPIN
    p = capi.default_params(-spin)
    p.integrator, p.r_max = capi.RK4, 11000.0
    out, st = ol.oracle_trace(p, rays, nthreads=2)
    o.kro_redshift_f64(-spin, -1.0, 1, 0, 0, ol.ptr(out), len(out))
    o.kro_range_phi_f64(-np.pi, np.pi, ol.ptr(out), len(out))
    b = capi.ImageBins()
    b.x0, b.y0, b.img_dx, b.img_dy = -30.0, -30.0, 60.0 / IMG, 60.0 / IMG
    b.r_isco, b.r_disc = o.kro_kerr_isco(spin, 1), 30.0
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = 3.0, 4.0, 3.0, 10.0, 3.0
    b.img_nx, b.img_ny, b.flip_image, b.pad = IMG, IMG, 1, 0
    npix = IMG * IMG
    n = np.zeros(npix, dtype=np.int32)
    planes = [np.zeros(npix) for _ in range(6)]
    dc = C.c_int64()
    o.kro_reduce_image_f64(C.byref(b), ol.ptr(out), len(out), ol.ptr(n), *[ol.ptr(x) for x in planes], C.byref(dc))
    return np.concatenate([n.astype(np.float64)] + planes + [[float(dc.value)]]), st


def _image_init():
    spec = ol.imageplane_spec(10000.0, 80.0, -30.0, 30.0, 60.0 / 23, -30.0, 30.0, 60.0 / 23, bench.SPIN)
    rays = ol.oracle_imageplane(spec)
    ol.oracle().kro_redshift_start_f64(-bench.SPIN, 0.0, 1, 0, ol.ptr(rays), len(rays))
    return rays


def _image_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = _image_init()[rank::world].copy()      # what kr_imageplane_init_strided_dev_f64(spec, first = rank, stride = world) generates
    h, st = _image_planes(mine)
    t = torch.from_numpy(h.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    q.put((rank, t.numpy().copy(), st["rays_traced"]))
    dist.barrier()
    dist.destroy_process_group()


def test_ray_cyclic_image_shards_reduce_to_the_global_planes():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_image_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full, st_full = _image_planes(_image_init())
    npix = IMG * IMG
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][1][:npix], full[:npix])          # per-pixel ray counts: exact
    assert res[0][1][-1] == full[-1] > 0                                   # disc rays
    np.testing.assert_allclose(res[0][1], full, rtol=1e-12)
    assert res[0][2] + res[1][2] == st_full["rays_traced"]


# ---- image plane, pixel-column-cyclic shards + gather (bench.py --image-exchange gather, the default) ---------------------
IMG_G = 24          # image = ray grid spacing (img_dx = dx), as in the application: one ray column per pixel column


def _image_planes_g(rays, img):
    o = ol.oracle()
    spin = bench.SPIN
    p = capi.default_params(-spin)
    p.integrator, p.r_max = capi.RK4, 11000.0
    out, st = ol.oracle_trace(p, rays, nthreads=2)
    o.kro_redshift_f64(-spin, -1.0, 1, 0, 0, ol.ptr(out), len(out))
    o.kro_range_phi_f64(-np.pi, np.pi, ol.ptr(out), len(out))
    b = capi.ImageBins()
    b.x0, b.y0, b.img_dx, b.img_dy = -30.0, -30.0, 60.0 / img, 60.0 / img
    b.r_isco, b.r_disc = o.kro_kerr_isco(spin, 1), 30.0
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = 3.0, 4.0, 3.0, 10.0, 3.0
    b.img_nx, b.img_ny, b.flip_image, b.pad = img, img, 1, 0
    npix = img * img
    n = np.zeros(npix, dtype=np.int32)
    planes = [np.zeros(npix) for _ in range(6)]
    dc = C.c_int64()
    o.kro_reduce_image_f64(C.byref(b), ol.ptr(out), len(out), ol.ptr(n), *[ol.ptr(x) for x in planes], C.byref(dc))
    return np.concatenate([n.astype(np.float64)] + planes + [[float(dc.value)]]), st


def _image_init_g():
    spec = ol.imageplane_spec(10000.0, 80.0, -30.0, 30.0, 60.0 / IMG_G, -30.0, 30.0, 60.0 / IMG_G, bench.SPIN)
    rays = ol.oracle_imageplane(spec)
    ol.oracle().kro_redshift_start_f64(-bench.SPIN, 0.0, 1, 0, ol.ptr(rays), len(rays))
    return rays


def _image_gather_worker(rank, world, port, q):
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nx = ny = IMG_G + 1
    idx = bench.image_shard_indices(nx, ny, rank, world)           # the slots kr_imageplane_init_emit_runs_dev_f64 fills on this rank
    mine = _image_init_g()[idx].copy()
    h, st = _image_planes_g(mine, IMG_G)
    t = torch.from_numpy(h.copy())
    fake = types.SimpleNamespace(N=IMG_G, world=world, rank=rank, exchange="gather")
    bench.ImagePlaneWorkload.exchange_planes(fake, t, dist)        # the bench's own exchange code, over gloo
    q.put((rank, t.numpy().copy(), st["rays_traced"], idx))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pixel_column_cyclic_image_shards_gather_to_the_global_planes(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_image_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    init = _image_init_g()
    full, st_full = _image_planes_g(init, IMG_G)
    # the shards partition the ray grid ...
    allidx = np.sort(np.concatenate([r[3] for r in res]))
    np.testing.assert_array_equal(allidx, np.arange((IMG_G + 1) ** 2))
    assert sum(r[2] for r in res) == st_full["rays_traced"]
    # ... own disjoint pixel columns, and rank 0 holds the single-process planes after the gather: no sum was needed, so even
    # the floating-point planes are bit-identical as long as each pixel's rays were added in the same order (one rank, one pixel)
    got = res[0][1]
    npix = IMG_G * IMG_G
    np.testing.assert_array_equal(got[:npix], full[:npix])
    assert got[-1] == full[-1] > 0
    np.testing.assert_allclose(got, full, rtol=1e-13)


def test_strong_scaling_rows_partition_the_base_grid():
    """bench.py --scaling strong: rank r of N takes rows r, r + N, ... of the N = 1 grid itself."""
    base = ol.oracle_pointsource(bench.make_spec(capi, D))
    live = base["steps"] != -1
    glob = np.stack([base["alpha"][live], base["beta"][live]], 1)
    for world in (2, 3):
        parts = []
        for r in range(world):
            rays = ol.oracle_pointsource(bench.make_spec(capi, D, r, world, strong=True))
            m = rays["steps"] != -1
            parts.append(np.stack([rays["alpha"][m], rays["beta"][m]], 1))
        union = np.concatenate(parts)
        assert len(union) == len(glob)
        key = lambda a: np.lexsort((a[:, 1], a[:, 0]))
        np.testing.assert_allclose(union[key(union)], glob[key(glob)], rtol=0, atol=1e-12)
        sizes = [len(x) for x in parts]
        assert max(sizes) - min(sizes) <= len(glob) // len(np.unique(glob[:, 0])) + 1      # balanced to within one row
