#!/usr/bin/env python3
"""bench.py -- headline benchmark: rays/s (and RK steps/s) to disc hit, PointSource emissivity a = 0.998.

One "step" = one pass of the hot path over one batch of synthetic rays, entirely in HBM:
    PointSource init -> redshift_start -> run_raytrace (RK4, theta_max = pi/2, r_max = 1000)
    -> range_phi -> redshift -> emissivity histogram  [-> RCCL all-reduce of the histogram when N > 1]
Workload at N = 1: BASELINE.json configs[1] -- par_example/emissivity.par_example source with --source_h=10
(source = 0 10 1E-3 1.5707, V = 0, spin = 0.998), cos(alpha) in [-0.995, 0.995), beta in [-pi, pi), a
3163 x 3163 grid ~ 1e7 rays, fixed-step RK4, fp64.  For N > 1 the grid is refined N-fold in cos(alpha) and
rank r owns rows r, r+N, r+2N, ... (row-cyclic: neighbouring rows cost the same, so ranks stay balanced),
i.e. per-GPU work is fixed (weak scaling); the only exchange is the all-reduce of the 5*Nr+1 histogram words.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R] [--integrator rk4|rk45|euler]
        N > 1 without RANK / WORLD_SIZE in the environment: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
        itself, as a CHILD process (one rank per GPU over RCCL), relays rank 0's JSON line and exits with the child's code; it touches no GPU
        itself.  Under torch.distributed.run (the driver's multi-GPU command) it is a rank.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# overlapping launches need hardware queues of their own (raytrace_cpu_amd/csrc/kr_capi.hip); must be set before the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

SPIN = 0.998
SOURCE = [0.0, 10.0, 1e-3, 1.5707]
R_MAX = 1000.0
NR, R_DISC, GAMMA = 100, 500.0, 2.0             # emissivity.cpp defaults (Nr = 100, r_esc default 500 for r_disc, gamma = 2)
FLOP_PER_STEP = {"euler": 100.0, "rk4": 340.0, "rk45": 590.0}   # algorithmic fp64 FLOP per step / per RK45 attempt
FP64_VECTOR_PEAK_TFLOPS = 78.6                   # MI355X vector fp64: 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0


def self_launch_argv(argv, gpus, port, python=None):
    """The command a rank-less `bench.py --gpus N` (N > 1) starts as a child: torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1 (the container's hostname may not resolve), the same bench arguments."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}", "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.join(ROOT, "bench.py"), *argv]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv, gpus):
    """Runs before anything touches the GPU (no torch import, no HIP call in this process): the reference's parallel loop needs no
    launcher (raytracer.cpp:104), so neither does `bench.py --gpus N`.  Relays the child's stdout -- rank 0's ONE JSON line -- and its
    stderr as they come; returns the child's exit code."""
    import subprocess
    cmd = self_launch_argv(argv, gpus, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL across processes needs it)
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in child.stdout:
        # (anything a library of the ranks still prints on stdout -- RCCL has a version banner -- goes to stderr: stdout is the JSON line's)
        out = sys.stdout if line.startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return child.wait()


def make_spec(capi, d, rank=0, world=1, refine=1, strong=False):
    """PointSource spec.  Base grid: spacing d in cos(alpha) and d*pi/0.995 in beta (same point count on both axes).
    world > 1, weak scaling: rank `rank` owns rows rank, rank+world, ... of the grid refined world-fold in cos(alpha).
    world > 1, strong=True: the rows rank, rank+world, ... of the BASE grid itself (fixed total work).
    refine > 1 (with world = 1): that whole refined grid, used for the flux normalisation."""
    s = capi.PointSourceSpec()
    for i in range(4):
        s.pos[i] = SOURCE[i]
    s.V, s.spin, s.tol, s.E = 0.0, SPIN, 100.0, 1.0
    if strong:
        s.cosalpha0, s.cosalphamax, s.dcosalpha = -0.995 + rank * d, 0.995, d * world
    else:
        s.cosalpha0, s.cosalphamax, s.dcosalpha = -0.995 + rank * (d / world), 0.995, d / refine
    s.beta0, s.betamax, s.dbeta = -math.pi, math.pi, d * math.pi / 0.995
    return s


def image_column_shard(nx, ny, rank, world, run_cols=1):
    """Pixel-column-cyclic shard of an nx x ny ray grid stored column after column (ray = i * ny + j): rank r owns the runs of
    `run_cols` ray columns number r, r + world, ...  Returns (first, stride, run, count) for kr_imageplane_init_emit_runs_dev_f64,
    whose slot k holds source ray first + (k // run) * stride + k % run."""
    run = run_cols * ny
    n_runs = -(-nx // run_cols)
    mine = range(rank, n_runs, world)
    count = sum(min(run_cols, nx - b * run_cols) for b in mine) * ny
    return rank * run, world * run, run, count


def image_shard_indices(nx, ny, rank, world, run_cols=1):
    """The source-ray index of every slot of that shard (what the kernel computes; used by the CPU tests)."""
    first, stride, run, count = image_column_shard(nx, ny, rank, world, run_cols)
    k = np.arange(count, dtype=np.int64)
    return first + (k // run) * stride + k % run


def grid_spacing_for(rays):
    return 1.99 / (math.sqrt(rays) - 1.0)


def emis_bins(capi, kerr_isco, n_primary):
    b = capi.EmisBins()
    b.r_isco = kerr_isco
    b.r_min = kerr_isco
    b.dr = math.exp(math.log(R_DISC / b.r_min) / NR)
    b.gamma, b.spin, b.num_primary_rays = GAMMA, SPIN, float(n_primary)
    b.nr, b.logbin = NR, 1
    return b


ARITHMETIC_NOTE = {
    "strict": "strict: IEEE fp64, reference association, every ray",
    "hybrid": "hybrid: rays whose theta-dot^2 or h is a cancellation residue -> strict IEEE path (side launch on SIMDs of their own); all others -> shared-reciprocal / Newton rcp,rsq / FMA path",
    "fast": "fast (opt-in): shared-reciprocal / Newton rcp,rsq / FMA path for every ray; ill-conditioned rays may end differently",
}


def omp_set_threads(n):
    """The OpenMP runtime the CPU checkers run on (libgomp, shared by oracle/_ref and oracle/libkr_oracle.so): team size of the next parallel regions."""
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def device_pipeline_histogram(lib, capi, api, spec, p, bins):
    """The device-resident pipeline bench.py times (device PointSource + redshift_start -> trace -> range_phi + redshift + histogram; the fused
    passes) once on `spec`: returns (histogram words, kr_stats)."""
    import torch
    n = api.pointsource_count(spec)[0]
    rays = torch.empty(n * capi.RAY_F64.itemsize, dtype=torch.uint8, device="cuda")
    res = torch.zeros(5 * bins.nr + 1, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    vp = C.c_void_p
    capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(spec), 0, 1, 0.0, 0, 0, vp(rays.data_ptr()), n, vp(stream)), "init_emit")
    st = api.trace_dev(p, rays.data_ptr(), n, stream=stream, want_stats=True)
    capi.check(lib, lib.kr_post_emissivity_dev_f64(SPIN, -1.0, 0, 0, 0, -math.pi, math.pi, C.byref(bins), vp(rays.data_ptr()), n, vp(res.data_ptr()), vp(stream)), "post")
    torch.cuda.synchronize()
    return res.cpu().numpy(), st


def histogram_check(h, nr, cnt, flux, emis, sg, stt, disc_count):
    """Device histogram words h = [count | flux | emis | sum_redshift | sum_time | disc_count] against the CPU reducer's arrays."""
    got_cnt = np.rint(h[:nr]).astype(np.int64)
    same = got_cnt == cnt
    worst, excluded_sum_bins = 0.0, int((~same & (cnt > 0)).sum())
    for k, w in enumerate((flux, emis, sg, stt)):
        g = h[(k + 1) * nr:(k + 2) * nr]
        m = same & (cnt > 0)
        if m.any():
            worst = max(worst, float(np.max(np.abs(g[m] - w[m]) / np.abs(w[m]))))
    return {"bins": int(nr), "bins_count_mismatch": int((~same).sum()), "max_count_diff": int(np.abs(cnt - got_cnt).max()), "max_rel_diff_on_matching_bins": worst,
            "bins_excluded_from_the_sum_check": excluded_sum_bins, "disc_rays_device": int(round(float(h[5 * nr]))), "disc_rays_cpu": int(disc_count), "tolerance": 1e-6}


def cpu_baseline(args, capi, api, integrator, d_full, flags=0, timed=None):
    """Times the CPU path on this box's host cores -- the reference's own run_raytrace (oracle/_ref) where it was built, else the oracle port --
    and checks the GPU results ray by ray and bin by bin against it.  Two legs:
      * thread sweep: the reference's `omp parallel for schedule(dynamic)` with an `omp atomic` progress counter (raytracer.cpp:104-124) does not
        scale to every hardware thread of a 256-thread host, so a ~1e6-ray sample of the workload is timed at 16 / 32 / 64 / 128 / 256 threads (those the
        box has) and the best team size is kept;
      * the headline grid ITSELF (3162^2 rays, what `value` of the GPU line is quoted on) at that team size -- unless the sweep says it would take
        more than ~90 s, or --cpu-sample-rays asks for a sample; the checks then run on the same rays the timing ran on.
    Three checks against that CPU run: `pipeline_bins_check` -- the histogram and step total of the TIMED device-resident pipeline itself (timed =
    {"hist", "steps"} of the last timed pass; device-built rays, fused passes), or of one more run of that pipeline on the sample grid when the CPU
    leg ran on a sample; `bins_check` and `rays_check` -- the reference-constructed rays traced through the host-pointer entry points."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol   # bench.py's cpu_baseline leg is one of the three places allowed to touch oracle/
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    kind = "reference" if ol.ref() is not None else "port"
    p = capi.default_params(SPIN)
    p.integrator, p.r_max = integrator, R_MAX

    def run_cpu(d, threads):
        spec = make_spec(capi, d)
        omp_set_threads(threads)
        if kind == "reference":
            src = ol.RefSource(spec)                       # the reference's own PointSource<double>
            src.lib.ref_redshift_start(src.h, 0.0, 0, 0)
            init = src.snapshot()
            t0 = time.perf_counter()
            src.run(p)                                     # Raytracer::run_raytrace, OpenMP
            wall = time.perf_counter() - t0
            out = src.snapshot()
            src.close()
        else:
            init = ol.oracle_pointsource(spec)
            ol.oracle().kro_redshift_start_f64(SPIN, 0.0, 0, 0, ol.ptr(init), len(init))
            t0 = time.perf_counter()
            out, _ = ol.oracle_trace(p, init, nthreads=threads)
            wall = time.perf_counter() - t0
        return spec, init, out, wall

    # leg 1: team size
    sweep = {}
    teams = sorted({t for t in (16, 32, 64, 128, 256) if t <= cores} | {cores}) if cores > 16 else [cores]
    d_sweep = grid_spacing_for(min(1.0e6, 25000.0 * cores))
    for t in teams:
        _, _, out_t, wall_t = run_cpu(d_sweep, t)
        live = out_t["steps"] != -1
        sweep[t] = {"rays": int(live.sum()), "wall_s": wall_t, "rays_per_sec": int(live.sum()) / wall_t, "rays_per_sec_per_thread": int(live.sum()) / wall_t / t}
    best = max(sweep, key=lambda t: sweep[t]["rays_per_sec"])
    # leg 2: the headline grid itself (or a sample)
    full_rays = (math.floor(1.99 / d_full) + 1) ** 2
    est_full_s = full_rays / sweep[best]["rays_per_sec"]
    on_headline_grid = not args.cpu_sample_rays and est_full_s <= 90.0
    d = d_full if on_headline_grid else grid_spacing_for(args.cpu_sample_rays or 25000 * cores)
    spec, init, cpu_rays, wall = run_cpu(d, best)
    omp_set_threads(cores)
    valid = cpu_rays["steps"] != -1
    n_valid = int(valid.sum())
    steps = int(np.abs(cpu_rays["steps"][valid].astype(np.int64)).sum())
    # parity of the emissivity bins on this sample: GPU (same init rays, through the C ABI) vs CPU
    o = ol.oracle()
    o.kro_range_phi_f64(-math.pi, math.pi, ol.ptr(cpu_rays), len(cpu_rays))
    o.kro_redshift_f64(SPIN, -1.0, 0, 0, 0, ol.ptr(cpu_rays), len(cpu_rays))
    n_primary = int(((spec.cosalphamax - spec.cosalpha0) / spec.dcosalpha) * ((spec.betamax - spec.beta0) / spec.dbeta))
    bins = emis_bins(capi, api.lib().kr_kerr_isco(SPIN, 1), n_primary)
    p.flags = flags                                    # the arithmetic mode the headline was measured in
    gpu_rays, _ = api.trace(p, init)
    api.range_phi(gpu_rays)
    api.redshift(SPIN, -1.0, 0, 0, gpu_rays)
    got = api.reduce_emissivity(bins, gpu_rays)
    nr = bins.nr
    cnt = np.zeros(nr, dtype=np.int64)
    flux, emis, sg, stt = (np.zeros(nr) for _ in range(4))
    dc = C.c_int64()
    o.kro_reduce_emissivity_f64(C.byref(bins), ol.ptr(cpu_rays), len(cpu_rays), ol.ptr(cnt), ol.ptr(flux), ol.ptr(emis), ol.ptr(sg), ol.ptr(stt), C.byref(dc))
    # ... and the pipeline bench.py TIMES: device-built rays (kr_pointsource_init_emit_dev_f64), fused passes, histogram left in HBM
    if on_headline_grid and timed is not None:
        pipe_h, pipe_steps, pipe_what = timed["hist"], int(timed["steps"]), "the last TIMED pass itself"
    else:
        pipe_h, pst = device_pipeline_histogram(api.lib(), capi, api, spec, p, bins)
        pipe_steps, pipe_what = int(pst["steps_total"]), "one more run of the timed pipeline, on the CPU leg's sample grid"
    pipeline_bins_check = histogram_check(pipe_h, nr, cnt, flux, emis, sg, stt, dc.value)
    pipeline_bins_check.update({"what": pipe_what + ": device PointSource + redshift_start -> trace -> range_phi + redshift + histogram, against the reference's constructor + run_raytrace "
                                "+ the oracle's O(N) passes and reducer on the same grid", "rk_steps_device": pipe_steps, "rk_steps_cpu": steps, "rk_steps_equal": pipe_steps == steps})
    same = cnt == got["count"]
    worst = 0.0
    for k, w in (("flux", flux), ("emis", emis), ("sum_redshift", sg), ("sum_time", stt)):
        g = got[k][same & (cnt > 0)]
        ww = w[same & (cnt > 0)]
        if len(ww):
            worst = max(worst, float(np.max(np.abs(g - ww) / np.abs(ww))))
    # ... and ray by ray: integer outcome fields equal, positions / redshift within 1e-9 relative
    ints_same = np.ones(len(init), dtype=bool)
    for k in ("status", "rdot_flips", "equatorial_crossings", "steps"):
        ints_same &= gpu_rays[k] == cpu_rays[k]
    close = np.ones(len(init), dtype=bool)
    with np.errstate(invalid="ignore"):
        for k in ("r", "theta", "redshift"):
            close &= ~(np.abs(gpu_rays[k] - cpu_rays[k]) > 1e-9 * np.maximum(np.abs(cpu_rays[k]), 1e-300))
    on_disc = valid & ((cpu_rays["status"] & 1) != 0)

    def worst_rel(mask):
        w = 0.0
        with np.errstate(invalid="ignore"):
            for k in ("r", "theta", "redshift"):
                d = np.abs(gpu_rays[k][mask] - cpu_rays[k][mask]) / np.maximum(np.abs(cpu_rays[k][mask]), 1e-300)
                if len(d):
                    w = max(w, float(np.nanmax(d)))
        return w
    rays_check = {"rays": n_valid, "integer_fields_differ": int((valid & ~ints_same).sum()),
                  "disc_rays": int(on_disc.sum()), "disc_rays_beyond_1e-9": int((on_disc & ints_same & ~close).sum()),
                  "disc_rays_worst_rel_diff(r, theta, redshift)": worst_rel(on_disc & ints_same),
                  "bit_identical_r_theta_frac": float(((gpu_rays["r"] == cpu_rays["r"]) & (gpu_rays["theta"] == cpu_rays["theta"]))[valid].mean())}
    # the same sample once more on the strict arithmetic (IEEE + - x / sqrt, correctly rounded sin / cos): how many rays carry the CPU's bits
    # in EVERY output of the trace + redshift (what the hybrid launch gives its flagged rays; all rays with --arithmetic strict)
    p.flags = 0
    strict_rays, _ = api.trace(p, init)
    api.range_phi(strict_rays)
    api.redshift(SPIN, -1.0, 0, 0, strict_rays)
    bits = valid.copy()
    for k in ("t", "r", "theta", "phi", "redshift"):
        bits &= (strict_rays[k].view(np.int64) == cpu_rays[k].view(np.int64)) | (np.isnan(strict_rays[k]) & np.isnan(cpu_rays[k]))
    for k in ("status", "rdot_flips", "equatorial_crossings", "steps"):
        bits &= strict_rays[k] == cpu_rays[k]
    rays_check["strict_arithmetic_bit_identical_frac(t, r, theta, phi, redshift, integer fields)"] = float(bits[valid].mean())
    unit = "rays/s"
    return {
        "value": n_valid / wall, "unit": unit, "cores": best, "host_threads_available": cores, "kind": kind,
        "sample": (f"the headline grid itself ({math.isqrt(len(init))}^2 grid points)" if on_headline_grid else
                   f"same source and angular ranges on a {math.isqrt(len(init))}^2-ish grid") +
                  f": {n_valid} rays, {steps} steps, run_raytrace only ({'reference sources, g++ -O2 -fopenmp -ffp-contract=off' if kind == 'reference' else 'oracle C port, gcc -O2 -fopenmp'}), "
                  f"OpenMP team of {best} threads = the best of the sweep below",
        "steps_per_sec": steps / wall, "wall_s": wall, "rays_per_sec_per_thread": n_valid / wall / best,
        "thread_sweep": {"what": f"run_raytrace on a {sweep[best]['rays']}-ray sample of the same source at each team size", "teams": {str(t): v for t, v in sweep.items()}},
        "pipeline_bins_check": pipeline_bins_check,
        "bins_check": {"bins": int(nr), "bins_count_mismatch": int((~same).sum()), "max_count_diff": int(np.abs(cnt - got["count"]).max()),
                       "max_rel_diff_on_matching_bins": worst, "bins_excluded_from_the_sum_check": int((~same & (cnt > 0)).sum()), "tolerance": 1e-6},
        "rays_check": rays_check,
    }


class EmissivityWorkload:
    """BASELINE configs[1] (and, with --integrator rk45, the app's own configs[0] integrator)."""
    name = "emissivity"

    def __init__(self, args, lib, capi, api, rank, world):
        self.lib, self.capi, self.api = lib, capi, api
        self.method = {"euler": capi.EULER, "rk4": capi.RK4, "rk45": capi.RK45}[args.integrator]
        self.d = grid_spacing_for(args.rays or 1e7)
        strong = args.scaling == "strong"
        self.spec = make_spec(capi, self.d, rank, world, strong=strong)
        self.n, self.n_ca, self.n_b = api.pointsource_count(self.spec)
        full = make_spec(capi, self.d, refine=1 if strong else world)     # the global grid, for the flux normalisation
        n_primary = int(((full.cosalphamax - full.cosalpha0) / full.dcosalpha) * ((full.betamax - full.beta0) / full.dbeta))
        self.bins = emis_bins(capi, lib.kr_kerr_isco(SPIN, 1), n_primary)
        self.p = capi.default_params(SPIN)
        self.p.integrator, self.p.r_max = self.method, R_MAX
        self.result_words = 5 * NR + 1
        self.describe = (f"PointSource emissivity lamp-post h=10 a=0.998 V=0, {args.integrator.upper()}, theta_max=pi/2 r_max=1000 "
                         f"(BASELINE configs[1]); grid {self.n_ca}x{self.n_b} per GPU")
        self.fused = not args.separate_passes
        self.pipeline = ("[pointsource_init+redshift_start]+trace+[range_phi+redshift+emissivity_histogram] ([..] = one fused pass each)" if self.fused
                         else "pointsource_init+redshift_start+trace+range_phi+redshift+emissivity_histogram")
        self.sharding = f"row-cyclic over {world} rank(s), " + ("fixed global grid (strong scaling)" if strong else "grid refined with the rank count (weak scaling)")
        self.scaling = args.scaling

    def step(self, d_rays, d_res, stream):
        lib, capi, vp = self.lib, self.capi, C.c_void_p
        n = self.n
        if self.fused:      # same per-ray arithmetic, two passes over the records instead of five (tests: test_fused_pipeline_ends_...)
            capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(self.spec), 0, 1, 0.0, 0, 0, vp(d_rays), n, vp(stream)), "init_emit")
            st = self.api.trace_dev(self.p, d_rays, n, stream=stream, want_stats=True)
            capi.check(lib, lib.kr_post_emissivity_dev_f64(SPIN, -1.0, 0, 0, 0, -math.pi, math.pi, C.byref(self.bins), vp(d_rays), n, vp(d_res), vp(stream)), "post")
            return st
        capi.check(lib, lib.kr_pointsource_init_dev_f64(C.byref(self.spec), vp(d_rays), n, vp(stream)), "init")
        capi.check(lib, lib.kr_redshift_start_dev_f64(SPIN, 0.0, 0, 0, vp(d_rays), n, vp(stream)), "redshift_start")
        st = self.api.trace_dev(self.p, d_rays, n, stream=stream, want_stats=True)
        capi.check(lib, lib.kr_range_phi_dev_f64(-math.pi, math.pi, vp(d_rays), n, vp(stream)), "range_phi")
        capi.check(lib, lib.kr_redshift_dev_f64(SPIN, -1.0, 0, 0, 0, vp(d_rays), n, vp(stream)), "redshift")
        capi.check(lib, lib.kr_reduce_emissivity_dev_f64(C.byref(self.bins), vp(d_rays), n, vp(d_res), vp(stream)), "reduce")
        return st

    def summary(self, h):
        return {"disc_hits": float(h[5 * NR])}


class ImagePlaneWorkload:
    """BASELINE configs[3]: imageplane_disc_image geometry (par_example: dist 1e4, incl 80, +-30, r_disc 30, q = 3), img_N = N = 4096
    -> 4097^2 rays, RK4 (RK45 never returns on the (0,0) pixel in the reference), ray-cyclic shards, 7 image planes reduced."""
    name = "imageplane"

    def __init__(self, args, lib, capi, api, rank, world):
        self.lib, self.capi, self.api = lib, capi, api
        self.method = {"euler": capi.EULER, "rk4": capi.RK4, "rk45": capi.RK45}[args.integrator]
        N1 = int(round(math.sqrt(args.rays))) - 1 if args.rays else 4096
        # strong scaling: the N1 x N1 image itself is divided among the ranks; weak: the grid grows so that every rank keeps (N1+1)^2 rays
        strong = args.scaling == "strong"
        N = N1 if (strong or world == 1) else int(round(math.sqrt(world) * (N1 + 1))) - 1
        s = capi.ImagePlaneSpec()
        s.dist, s.inc_deg, s.x0, s.xmax, s.y0, s.ymax = 10000.0, 80.0, -30.0, 30.0, -30.0, 30.0
        s.dx = s.dy = 60.0 / N
        s.spin, s.phi0, s.precision = SPIN, 0.0, 100.0
        self.spec, self.N = s, N
        total, nx, ny = api.imageplane_count(s)
        self.total, self.nx, self.ny, self.rank, self.world = total, nx, ny, rank, world
        self.exchange = args.image_exchange if world > 1 else "none"
        if args.image_exchange == "gather":
            # pixel-column-cyclic: rank r owns the ray columns (= pixel columns, img_dx = dx) r, r + world, ...: disjoint pixel sets,
            # neighbouring columns cost the same -> balanced; the planes are gathered on rank 0, nothing is summed
            self.first, self.stride, self.run, self.n = image_column_shard(nx, ny, rank, world)
            self.sharding = f"pixel-column-cyclic over {world} rank(s) (disjoint pixel sets; planes gathered on rank 0 over the direct xGMI links)"
        else:
            self.first, self.stride, self.run = rank, world, 1
            self.n = (total - rank + world - 1) // world
            self.sharding = f"ray-cyclic over {world} rank(s) (planes all-reduced)"
        self.sharding += ", fixed image (strong scaling)" if strong else ", image grown with the rank count (weak scaling)"
        self.scaling = args.scaling
        b = capi.ImageBins()
        b.x0, b.y0, b.img_dx, b.img_dy = s.x0, s.y0, 60.0 / N, 60.0 / N
        b.r_isco, b.r_disc = lib.kr_kerr_isco(SPIN, 1), 30.0
        b.q1, b.rb1, b.q2, b.rb2, b.q3 = 3.0, 4.0, 3.0, 10.0, 3.0
        b.img_nx, b.img_ny, b.flip_image, b.pad = N, N, 1, 0
        self.bins = b
        self.p = capi.default_params(-SPIN)              # ImagePlane stores the spin negated (imageplane.cpp:12)
        self.p.integrator, self.p.r_max = self.method, 1.1 * s.dist
        self.p.flags |= (3 << 8)                         # KR_FLAG_BLOCKS_PER_CU(3): no long-ray tail on this workload (max ~2000 steps)
        self.result_words = 7 * N * N + 1
        self.describe = (f"ImagePlane disc image dist=1e4 incl=80 a=0.998 x,y in +-30, {nx}x{ny} rays, img {N}x{N}, {args.integrator.upper()}, "
                         f"r_max=1.1*dist (BASELINE configs[3])")
        self.fused = not args.separate_passes
        self.pipeline = ("[imageplane_init+redshift_start]+trace+[redshift+range_phi+image_planes] ([..] = one fused pass each)" if self.fused
                         else "imageplane_init+redshift_start+trace+redshift+range_phi+image_planes")

    def step(self, d_rays, d_res, stream):
        lib, capi, vp = self.lib, self.capi, C.c_void_p
        n = self.n
        if self.fused:
            capi.check(lib, lib.kr_imageplane_init_emit_runs_dev_f64(C.byref(self.spec), self.first, self.stride, self.run, 0.0, 1, 0, vp(d_rays), n, vp(stream)), "init_emit")
            st = self.api.trace_dev(self.p, d_rays, n, stream=stream, want_stats=True)
            capi.check(lib, lib.kr_post_image_dev_f64(-SPIN, -1.0, 1, 0, 0, -math.pi, math.pi, C.byref(self.bins), vp(d_rays), n, vp(d_res), vp(stream)), "post")
            return st
        if self.run != 1:
            raise SystemExit("--separate-passes supports the ray-cyclic shards only (use --image-exchange allreduce)")
        capi.check(lib, lib.kr_imageplane_init_strided_dev_f64(C.byref(self.spec), self.first, self.stride, vp(d_rays), n, vp(stream)), "init")
        capi.check(lib, lib.kr_redshift_start_dev_f64(-SPIN, 0.0, 1, 0, vp(d_rays), n, vp(stream)), "redshift_start")
        st = self.api.trace_dev(self.p, d_rays, n, stream=stream, want_stats=True)
        capi.check(lib, lib.kr_redshift_dev_f64(-SPIN, -1.0, 1, 0, 0, vp(d_rays), n, vp(stream)), "redshift")
        capi.check(lib, lib.kr_range_phi_dev_f64(-math.pi, math.pi, vp(d_rays), n, vp(stream)), "range_phi")
        capi.check(lib, lib.kr_reduce_image_dev_f64(C.byref(self.bins), vp(d_rays), n, vp(d_res), vp(stream)), "reduce")
        return st

    def exchange_planes(self, res, dist):
        """The image pipeline's one exchange.  all-reduce: every rank ends with the summed planes (0.94 GB through the ring at 4096^2).
        gather: the ranks own disjoint pixel columns, so rank 0 just collects them -- (world-1)/world of the planes arrive over its
        7 direct xGMI links in parallel, nothing is added; the other ranks' `res` keeps their own columns only."""
        import torch
        if self.exchange == "allreduce":
            dist.all_reduce(res, op=dist.ReduceOp.SUM)
            return
        N, W, r = self.N, self.world, self.rank
        cols = -(-N // W)                                    # columns per rank, padded to the same size for the collective
        planes = res[:7 * N * N].view(7, N, N)
        mine = torch.zeros(7 * cols * N + 1, dtype=res.dtype, device=res.device)
        own = planes[:, r::W, :]
        mine[:7 * cols * N].view(7, cols, N)[:, :own.shape[1], :] = own
        mine[-1] = res[-1]                                   # this rank's disc-ray count
        parts = [torch.empty_like(mine) for _ in range(W)] if r == 0 else None
        dist.gather(mine, parts, dst=0)
        if r == 0:
            count = 0.0
            for q, part in enumerate(parts):
                nq = len(range(q, N, W))
                planes[:, q::W, :] = part[:7 * cols * N].view(7, cols, N)[:, :nq, :]
                count = count + part[-1]
            res[-1] = count

    def summary(self, h):
        return {"disc_hits": float(h[-1]), "lit_pixels": int((h[: self.N * self.N] > 0).sum())}


class ReturnRadiationWorkload:
    """BASELINE configs[4]: disc -> disc returning radiation.  Nr source radii log-spaced r_isco .. 500 on the disc (theta = pi/2 - 1e-6,
    Keplerian V), ~1e6 rays each over beta in [0, pi), Euler to 1.1 r_esc (disc_source_photonfrac_r.cpp:74-135), one relaunch per
    radius, escape/return/lost weighted fractions per radius.  Radii are sharded across ranks; the Nr x 4 table is all-reduced."""
    name = "return_radiation"

    def __init__(self, args, lib, capi, api, rank, world):
        self.lib, self.capi, self.api = lib, capi, api
        self.method = {"euler": capi.EULER, "rk4": capi.RK4, "rk45": capi.RK45}[args.integrator]
        self.nr = args.radii * (world if args.scaling == "weak" else 1)      # weak: every rank keeps args.radii source radii
        self.scaling = args.scaling
        rays = args.rays or 1e6
        self.r_isco = lib.kr_kerr_isco(SPIN, 1)
        dr = math.exp(math.log(R_DISC / self.r_isco) / self.nr)
        self.radii = [(ir, self.r_isco * dr ** ir) for ir in range(self.nr) if ir % world == rank]
        d = 1.99 / (math.sqrt(rays) - 1.0)
        self.specs = []
        for ir, r_s in self.radii:
            s = capi.PointSourceSpec()
            for i, v in enumerate([0.0, r_s, math.pi / 2 - 1e-6, 1.5707]):
                s.pos[i] = v
            s.V, s.spin, s.tol, s.E = lib.kr_disc_velocity(r_s, SPIN, 1), SPIN, 100.0, 1.0
            s.cosalpha0, s.cosalphamax, s.dcosalpha = -0.995, 0.995, d
            s.beta0, s.betamax, s.dbeta = 0.0, math.pi, d * (math.pi / 2) / 0.995
            self.specs.append(s)
        self.counts = [api.pointsource_count(s)[0] for s in self.specs]
        self.n = max(self.counts)
        self.nstreams, self.streams, self.buffers, self.order, self.ordered = max(0, args.streams), None, None, None, False
        self.groups = max(1, int(getattr(args, "rr_groups", 4)))
        self.p = capi.default_params(SPIN)
        self.p.integrator, self.p.r_max = self.method, 1.1 * R_MAX
        if os.environ.get("KR_RR_BLOCKS"):                 # experiment: resident waves per SIMD of the merged main launch (scripts/gpu_euler_occ.sh)
            self.p.flags |= (int(os.environ["KR_RR_BLOCKS"]) & 0xF) << 8
        self.result_words = 4 * self.nr
        self.describe = (f"disc->disc returning radiation: {self.nr} source radii r_isco..500 x ~{int(rays)} rays (beta in [0,pi)), {args.integrator.upper()}, "
                         f"r_max=1.1*r_esc, per-radius relaunch (BASELINE configs[4])")
        self.pipeline = ("per radius: [pointsource_init+redshift_start]+trace+[range_phi+return_classification]; " +
                         (f"radii round-robin over {self.nstreams} stream(s), longest launches first, one counter read-back at the end" if self.nstreams else
                          f"all radii resident, traced as {self.groups} merged batch(es) on streams of their own (one side launch + one main launch per batch; "
                          "the O(N) passes of all radii of a batch in ceil(k / 24) + ceil(k / 32) launches)"))
        self.sharding = f"radii cyclic over {world} rank(s), " + ("fixed set of radii (strong scaling)" if args.scaling == "strong" else "radii added with the rank count (weak scaling)")

    def step(self, d_rays, d_res, stream):
        """Radii go round-robin over self.nstreams HIP streams (each with its own ray buffer), the radii whose launches have the
        longest tails first, so that the long-ray tail of one radius overlaps the bulk of the next ones; counters are collected
        once, after the last radius (kr_trace_async / kr_trace_wait).  --streams 1 is the old serial relaunch loop."""
        import torch
        lib, capi, vp = self.lib, self.capi, C.c_void_p
        if self.nstreams == 0:
            return self.step_merged(d_rays, d_res, stream)
        if self.streams is None:
            self.streams = [torch.cuda.Stream() for _ in range(self.nstreams)] if self.nstreams > 1 else []
            self.buffers = [torch.empty(self.n * capi.RAY_F64.itemsize, dtype=torch.uint8, device="cuda") for _ in self.streams]
            self.order = list(range(len(self.radii)))
        cur = torch.cuda.current_stream()
        lanes = [(s.cuda_stream, b.data_ptr()) for s, b in zip(self.streams, self.buffers)] or [(stream, d_rays)]
        if not self.ordered:
            lanes = [(stream, d_rays)]               # first (warm-up) pass: one launch at a time, to learn each radius's own launch time
        for s in self.streams:
            s.wait_stream(cur)                       # behind the zeroing of the result table
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(cur)
        tickets, done = [], []
        for k, j in enumerate(self.order):
            (ir, r_s), s = self.radii[j], self.specs[j]
            st_k, rays_k = lanes[k % len(lanes)]
            n = self.counts[j]
            capi.check(lib, lib.kr_pointsource_init_emit_dev_f64(C.byref(s), 0, 1, s.V, 0, 0, vp(rays_k), n, vp(st_k)), "init+redshift_start (fused)")
            if len(tickets) - len(done) >= 4 * len(lanes):          # bounded pipeline depth: collect the oldest launch's counters
                jj, tt = tickets[len(done)]
                done.append((jj, self.api.trace_wait(tt)))
            tickets.append((j, self.api.trace_async(self.p, rays_k, n, stream=st_k)))
            b = capi.ReturnBins()
            b.r_isco, b.r_disc, b.r_esc, b.source_r, b.source_phi = self.r_isco, R_DISC, R_MAX, r_s, 1.5707
            b.plane_iso, b.limb, b.weight_norm, b.pad = 1, 0, 1, 0
            capi.check(lib, lib.kr_post_return_dev_f64(-math.pi, math.pi, C.byref(b), vp(rays_k), n, vp(d_res + 32 * ir), vp(st_k)), "range_phi+reduce (fused)")
        for s in self.streams:
            cur.wait_stream(s)                       # whatever follows on the caller's stream (the all-reduce) sees every radius
        t1.record(cur)
        tot, per = None, {}
        done += [(j, self.api.trace_wait(t)) for j, t in tickets[len(done):]]
        for j, st in done:
            per[j] = st["kernel_ms"]
            if tot is None:
                tot = dict(st)
            else:
                for key in ("rays_traced", "steps_total", "rk45_attempts", "rk45_rejects", "rk45_stationary_steps", "rk45_extrapolated_steps", "kernel_ms"):
                    tot[key] += st[key]
        t1.synchronize()
        tot["sum_of_launch_ms"] = tot["kernel_ms"]
        tot["kernel_ms"] = t0.elapsed_time(t1)       # the span the overlapping launches took together, HIP events on the caller's stream
        if not self.ordered and self.streams:
            # first pass: from now on, longest launch first (a launch's length beyond the mean is its long-ray tail)
            self.order = sorted(self.order, key=lambda j: -per[j])
            self.ordered = True
        return tot

    def step_merged(self, d_rays, d_res, stream):
        """--streams 0: every radius keeps its own ray buffer in HBM (100 x 144 MB = 14.4 GB of 288) and the whole sweep over radii is
        ONE kr_trace_batch_async_f64 call -- one classification per radius, then one side launch and one main launch over all of them
        (trace_multi_kernel) -- on one stream; the per-radius classification passes follow."""
        import torch
        lib, capi, vp = self.lib, self.capi, C.c_void_p
        if self.buffers is None:
            self.buffers = [torch.empty(c * capi.RAY_F64.itemsize, dtype=torch.uint8, device="cuda") for c in self.counts]
        cur = torch.cuda.current_stream()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(cur)
        # (the 100 source and 100 reducer kernels stay on this one stream: spread over 2 / 4 auxiliary streams the pass took 236 / 252 ms instead
        # of 233 -- streams that wait on the merged batch slow it down, profiles/r03_ab_experiments.txt)
        groups = int(os.environ.get("KR_RR_GROUPS", self.groups))
        if groups > 1 and len(self.specs) >= 2 * groups:
            return self.step_grouped(d_res, stream, groups, cur, t0, t1)
        st_of = [stream] * len(self.specs)
        k = len(self.specs)
        if getattr(self, "batch_args", None) is None:
            # the host-side argument arrays of the two batched O(N) passes (all radii in ceil(k / 24) + ceil(k / 32) launches instead of 2 k)
            specs = (capi.PointSourceSpec * k)(*self.specs)
            V = (C.c_double * k)(*[s.V for s in self.specs])
            ptrs = (C.c_void_p * k)(*[b.data_ptr() for b in self.buffers])
            ns = (C.c_int64 * k)(*self.counts)
            bins = (capi.ReturnBins * k)()
            for j, (ir, r_s) in enumerate(self.radii):
                b = bins[j]
                b.r_isco, b.r_disc, b.r_esc, b.source_r, b.source_phi = self.r_isco, R_DISC, R_MAX, r_s, 1.5707
                b.plane_iso, b.limb, b.weight_norm, b.pad = 1, 0, 1, 0
            self.batch_args = (specs, V, ptrs, ns, bins)
        specs, V, ptrs, ns, bins = self.batch_args
        capi.check(lib, lib.kr_pointsource_init_emit_batch_dev_f64(k, specs, V, 0, 0, ptrs, ns, vp(stream)), "init+redshift_start (fused, all radii)")
        tickets = self.api.trace_batch_async([self.p] * k, [b.data_ptr() for b in self.buffers], self.counts, st_of)
        outs = (C.c_void_p * k)(*[d_res + 32 * ir for ir, _ in self.radii])
        capi.check(lib, lib.kr_post_return_batch_dev_f64(k, -math.pi, math.pi, bins, ptrs, ns, outs, vp(stream)), "range_phi+reduce (fused, all radii)")
        t1.record(cur)
        tot = self.api.trace_wait_many(tickets)          # one call for the hundred tickets: their counters summed
        t1.synchronize()
        tot["kernel_ms"] = t0.elapsed_time(t1)
        return tot

    def step_grouped(self, d_res, stream, groups, cur, t0, t1):
        """The radii in G interleaved groups (--rr-groups, default 4), each a merged batch of its own on a stream of its own, so that one group's
        memory-bound O(N) passes run beside another group's compute-bound trace: 100 radii x 1e6 rays 223.9 / 219.5 / 214.8-216.7 / 227 / 223 ms
        with 1 / 2 / 4 / 5 / 8 groups (more groups: smaller merged launches, more side launches); the same sums to the last bits."""
        import torch
        lib, capi, vp = self.lib, self.capi, C.c_void_p
        if getattr(self, "group_args", None) is None:
            self.group_streams = [torch.cuda.Stream() for _ in range(groups)]
            self.group_args = []
            for g in range(groups):
                idx = list(range(g, len(self.specs), groups))
                k = len(idx)
                specs = (capi.PointSourceSpec * k)(*[self.specs[j] for j in idx])
                V = (C.c_double * k)(*[self.specs[j].V for j in idx])
                ptrs = (C.c_void_p * k)(*[self.buffers[j].data_ptr() for j in idx])
                ns = (C.c_int64 * k)(*[self.counts[j] for j in idx])
                bins = (capi.ReturnBins * k)()
                for q, j in enumerate(idx):
                    b = bins[q]
                    b.r_isco, b.r_disc, b.r_esc, b.source_r, b.source_phi = self.r_isco, R_DISC, R_MAX, self.radii[j][1], 1.5707
                    b.plane_iso, b.limb, b.weight_norm, b.pad = 1, 0, 1, 0
                self.group_args.append((idx, specs, V, ptrs, ns, bins))
        tickets = []
        for g, (idx, specs, V, ptrs, ns, bins) in enumerate(self.group_args):
            gs = self.group_streams[g]
            gs.wait_stream(cur)
            k = len(idx)
            outs = (C.c_void_p * k)(*[d_res + 32 * self.radii[j][0] for j in idx])       # (the result table is double-buffered under torch.distributed: per call)
            capi.check(lib, lib.kr_pointsource_init_emit_batch_dev_f64(k, specs, V, 0, 0, ptrs, ns, vp(gs.cuda_stream)), "init")
            tickets += self.api.trace_batch_async([self.p] * k, [self.buffers[j].data_ptr() for j in idx], [self.counts[j] for j in idx], [gs.cuda_stream] * k)
            capi.check(lib, lib.kr_post_return_batch_dev_f64(k, -math.pi, math.pi, bins, ptrs, ns, outs, vp(gs.cuda_stream)), "post")
        for gs in self.group_streams:
            cur.wait_stream(gs)
        t1.record(cur)
        tot = self.api.trace_wait_many(tickets)
        t1.synchronize()
        tot["kernel_ms"] = t0.elapsed_time(t1)
        return tot

    def summary(self, h):
        t = h.reshape(self.nr, 4)
        with np.errstate(invalid="ignore", divide="ignore"):
            f = t[:, 1:] / t[:, :1]
        pick = [0, self.nr // 4, self.nr // 2, self.nr - 1]
        return {"fractions_escape_return_lost": {f"radius_index_{i}": [float(f[i, 1]), float(f[i, 0]), float(f[i, 2])] for i in pick}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="emissivity", choices=["emissivity", "imageplane", "return_radiation"])
    ap.add_argument("--radii", type=int, default=100, help="return_radiation: number of source radii")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = per-GPU work fixed (the grid / image / set of radii grows with N); strong = the N = 1 problem divided among the ranks")
    ap.add_argument("--image-exchange", default="gather", choices=["gather", "allreduce"],
                    help="imageplane, N > 1: gather = pixel-column-cyclic shards, planes collected on rank 0; allreduce = ray-cyclic shards, planes summed everywhere")
    ap.add_argument("--no-overlap-exchange", action="store_true", help="N > 1: run the exchange on the compute stream instead of beside the next pass")
    ap.add_argument("--rr-groups", type=int, default=4, help="return_radiation, --streams 0: the radii are traced as this many merged batches on streams of their own "
                                                              "(one group's O(N) passes beside another's trace); 1 = one merged batch")
    ap.add_argument("--streams", type=int, default=0, help="return_radiation: HIP streams the per-radius launches are spread over (1 = serial relaunch; 0 = all radii resident, one merged batch)")
    ap.add_argument("--rays", type=float, default=0, help="rays per GPU (default: 1e7 emissivity = BASELINE configs[1]; 4097^2 imageplane = configs[3])")
    ap.add_argument("--integrator", default="rk4", choices=["euler", "rk4", "rk45"])
    ap.add_argument("--arithmetic", default="auto", choices=["auto", "hybrid", "strict", "fast"],
                    help="hybrid (KR_FLAG_HYBRID: strict for ill-conditioned rays, fast for the rest), strict (flags = 0), fast (KR_FLAG_FAST_MATH); "
                         "auto = hybrid for euler / rk4, strict for rk45 (what the host mirror of the class API does)")
    ap.add_argument("--fast-math", action="store_true", help="same as --arithmetic fast")
    ap.add_argument("--separate-passes", action="store_true", help="emissivity / imageplane: the five O(N) passes one kernel each instead of the two fused ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-math-extra", action="store_true")
    ap.add_argument("--cpu-sample-rays", type=float, default=0)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))     # (nothing above or in there initialises the GPU)

    import torch
    from raytrace_cpu_amd import api, capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or leave RANK / WORLD_SIZE unset and bench.py starts them)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this node shows {torch.cuda.device_count()} device(s)")
    torch.cuda.set_device(local_rank)
    lib = api.lib()
    capi.check(lib, lib.kr_set_device(local_rank), "kr_set_device")
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("KR_BENCH_FORCE_DIST"):
        # under torch.distributed.run the collective path is used even with one rank, so that it is exercised on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a five-line version banner on STDOUT when its first communicator comes up; stdout carries rank 0's ONE JSON line and
        # nothing else, so the communicator is created (a first all-reduce) with file descriptor 1 pointing at stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    if args.workload == "return_radiation" and args.integrator == "rk4" and "--integrator" not in " ".join(sys.argv):
        args.integrator = "euler"         # the reference driver uses the Euler integrator (disc_source_photonfrac_r.cpp:92)
    wl = {"emissivity": EmissivityWorkload, "imageplane": ImagePlaneWorkload, "return_radiation": ReturnRadiationWorkload}[args.workload](args, lib, capi, api, rank, world)
    if args.fast_math:
        args.arithmetic = "fast"
    if args.arithmetic == "auto":
        args.arithmetic = "strict" if args.integrator == "rk45" else "hybrid"
    mode_flags = {"strict": 0, "hybrid": capi.FLAG_HYBRID, "fast": capi.FLAG_FAST_MATH}
    mode_mask = capi.FLAG_HYBRID | capi.FLAG_FAST_MATH
    wl.p.flags = (wl.p.flags & ~mode_mask) | mode_flags[args.arithmetic]
    n = wl.n
    rays = torch.empty(n * capi.RAY_F64.itemsize, dtype=torch.uint8, device="cuda")
    # the result (histogram / planes / fraction table) is double-buffered: the exchange of pass k runs on a stream of its own while
    # pass k+1 initialises and traces its rays into the other buffer
    overlap = dist is not None and not args.no_overlap_exchange
    results = [torch.zeros(wl.result_words, dtype=torch.float64, device="cuda") for _ in range(2 if overlap else 1)]
    comm_stream = torch.cuda.Stream() if overlap else None
    exchanged = [torch.cuda.Event() for _ in results]
    state = {"k": 0, "last": results[0]}
    stream = torch.cuda.current_stream().cuda_stream
    d_rays = rays.data_ptr()

    def exchange(res):
        if hasattr(wl, "exchange_planes"):
            wl.exchange_planes(res, dist)
        else:
            dist.all_reduce(res, op=dist.ReduceOp.SUM)      # RCCL over xGMI: the path's one exchange

    def one_step():
        cur = torch.cuda.current_stream()
        slot = state["k"] % len(results)
        res = results[slot]
        state["k"] += 1
        state["last"] = res
        if overlap:
            cur.wait_event(exchanged[slot])                 # the exchange that last read this buffer has finished
        res.zero_()
        st = wl.step(d_rays, res.data_ptr(), stream)
        if dist is not None:
            if overlap:
                comm_stream.wait_stream(cur)
                with torch.cuda.stream(comm_stream):
                    exchange(res)
                    exchanged[slot].record(comm_stream)
            else:
                exchange(res)
        return st

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, steps_total, traced, stats_last = [], 0, 0, {}
    for _ in range(args.steps):
        st = one_step()
        kernel_ms.append(st["kernel_ms"])
        if os.environ.get("KR_BENCH_VERBOSE"):
            print(f"step: kernel_ms {st['kernel_ms']:.1f} strict_side {st.get('strict_side_ms', 0):.1f} main {st.get('main_ms', 0):.1f}", file=sys.stderr, flush=True)
        steps_total, traced, stats_last = st["steps_total"], st["rays_traced"], st
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([float(traced), float(steps_total)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        traced_all, steps_all = int(tot[0].item()), int(tot[1].item())
    else:
        traced_all, steps_all = traced, steps_total

    h = state["last"].cpu().numpy()
    # the other two arithmetic modes on the same workload, reported beside the headline
    others = None
    if world == 1 and not args.no_fast_math_extra:
        others = {}
        for mode in ("strict", "hybrid", "fast"):
            if mode == args.arithmetic:
                continue
            keep = wl.p.flags
            wl.p.flags = (wl.p.flags & ~mode_mask) | mode_flags[mode]
            one_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fst = [one_step() for _ in range(2)]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 2
            wl.p.flags = keep
            others[mode] = {"value": fst[-1]["rays_traced"] / dt, "unit": "rays/s", "rk_steps_per_sec": fst[-1]["steps_total"] / dt,
                            "avg_kernel_ms": float(np.mean([x["kernel_ms"] for x in fst])), "ms_per_step": 1e3 * dt}
    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        avg_kernel_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        # RK45 work unit = one EVALUATED trial step (7 derivative evaluations): the steps of captured rays that are replayed as bare
        # t / phi additions (rk45_stationary_steps) or extrapolated in creep mode (k1 only; rk45_extrapolated_steps) are not priced
        units = (stats_last["rk45_attempts"] - stats_last["rk45_stationary_steps"] - stats_last["rk45_extrapolated_steps"]) if args.integrator == "rk45" else steps_total
        flop = FLOP_PER_STEP[args.integrator] * units                                            # this rank's launch
        achieved_tflops = flop / (avg_kernel_ms * 1e-3) / 1e12
        # HBM bytes of the trace kernels per pass: PMC counters cannot be read from inside this process (rocprofv3 wraps it), so `traffic` is null
        # on this line and the figure of the round's committed counter run rides along under its own name, with its source
        traffic_from_profile = None
        tfile = os.path.join(ROOT, "profiles", "trace_kernel_hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if f"{args.workload}_{args.integrator}" in tj:
                    traffic_from_profile = {"bytes_per_pass": tj[f"{args.workload}_{args.integrator}"], "source": tj.get("_source", "profiles/trace_kernel_hbm_traffic.json")}
            except Exception:
                traffic_from_profile = None
        out = {
            "metric": "rays_per_sec", "value": traced_all * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist is not None else None), "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.describe, "rays_per_gpu": int(traced), "allocated_rays_per_gpu": int(n), "rays_total": int(traced_all),
                       "integrator": args.integrator, "arithmetic": ARITHMETIC_NOTE[args.arithmetic], "rays_on_strict_side_launch": int(stats_last.get("rays_strict_side", 0)), "pipeline": wl.pipeline + (("+rccl_" + (getattr(wl, "exchange", "allreduce")) + ("(beside the next pass)" if overlap else "")) if dist is not None else ""),
                       "sharding": wl.sharding},
            "rk_steps_per_sec": steps_all * args.steps / elapsed,
            "rk_steps_per_launch": int(steps_total), "mean_steps_per_ray": steps_total / max(traced, 1),
            # the strict side launch's share of the work (split traces): with per-kernel profiler counters, instructions per step of EACH launch
            "launch_split": {"rays_strict_side": int(stats_last.get("rays_strict_side", 0)), "steps_strict_side": int(stats_last.get("steps_strict_side", 0)),
                             "work_units_strict_side": int(stats_last.get("rk45_evaluated_strict_side", 0) if args.integrator == "rk45" else stats_last.get("steps_strict_side", 0))},
            "roofline": {"bound": "valu_fp64", "kernel": "kr::trace_kernel<double>", "achieved": achieved_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / FP64_VECTOR_PEAK_TFLOPS,
                         "flop_per_step": FLOP_PER_STEP[args.integrator], "work_units_per_launch": int(units), "avg_kernel_ms": avg_kernel_ms,
                         "kernel_steps_per_sec": steps_total / (avg_kernel_ms * 1e-3),
                         "split_launch_ms": {"strict_side": stats_last.get("strict_side_ms", 0.0), "main": stats_last.get("main_ms", 0.0)},
                         "longest_ray": critical_ray(stats_last, units, FLOP_PER_STEP[args.integrator]),
                         "hbm": {"algorithmic_bytes": 288 * int(traced), "achieved_gbs": 288 * traced / (avg_kernel_ms * 1e-3) / 1e9,
                                 "peak_gbs": HBM_PEAK_GBS, "frac": 288 * traced / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "traffic": None, "traffic_from_profile": traffic_from_profile,
                         "note": "latency-bound scalar fp64 ODE: neither HBM nor MFMA bounds it (SURVEY.md 8d); priced against vector fp64 peak"},
        }
        out.update(wl.summary(h))
        if others:
            out["other_arithmetic_modes"] = others
        if args.integrator == "rk45":
            out["rk45"] = {k: int(stats_last[k]) for k in ("rk45_attempts", "rk45_rejects", "rk45_stationary_steps", "rk45_extrapolated_steps")}
            out["rk45"]["evaluated_trial_steps"] = int(units)
        if world > 1:
            pass                                    # cpu_baseline is an N = 1 leg only
        elif not args.no_cpu_baseline and args.workload == "emissivity":
            out["cpu_baseline"] = cpu_baseline(args, capi, api, wl.method, wl.d, mode_flags[args.arithmetic], timed={"hist": h, "steps": steps_total})
        elif not args.no_cpu_baseline and args.workload == "imageplane":
            out["cpu_baseline"] = cpu_baseline_imageplane(args, capi, api, wl)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def critical_ray(st, units, flop_per_unit):
    """What the launch's longest ray says about the roofline fraction (kr_stats.longest_ray_steps*): a ray is ONE sequential chain of steps
    on a wave, so the launch that carries it lasts at least (its steps) x (that wave's time per step), however many CUs idle beside it.
    For a split trace the strict side launch is that chain: `strict_side_us_per_step` is its duration over its longest ray's steps, and
    `frac_cap` the roofline fraction the pass would have if it lasted exactly as long as that launch."""
    out = {"steps": int(st.get("longest_ray_steps", 0))}
    side_steps, side_ms = int(st.get("longest_ray_steps_strict_side", 0)), st.get("strict_side_ms", 0.0)
    if side_steps > 0 and side_ms > 0:
        out.update({"strict_side_steps": side_steps, "strict_side_ms": side_ms, "strict_side_us_per_step": 1e3 * side_ms / side_steps})
        if side_ms >= 0.5 * st.get("kernel_ms", 0.0):           # (a cap only where that launch is a large part of the pass)
            out["frac_cap"] = flop_per_unit * units / (side_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS
    return out


def cpu_baseline_imageplane(args, capi, api, wl):
    """The reference's ImagePlane<double> + run_raytrace on a coarser grid of the same plane, on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    N = int(math.sqrt(args.cpu_sample_rays or 20000 * cores)) | 1        # odd: no pixel at exactly (0,0)
    spec = ol.imageplane_spec(10000.0, 80.0, -30.0, 30.0, 60.0 / N, -30.0, 30.0, 60.0 / N, SPIN)
    p = capi.copy_params(wl.p)
    kind = "reference" if ol.ref() is not None else "port"
    if kind == "reference":
        src = ol.RefSource(spec)
        src.lib.ref_redshift_start(src.h, 0.0, 1, 0)
        init = src.snapshot()
        t0 = time.perf_counter()
        src.run(p)
        wall = time.perf_counter() - t0
        out = src.snapshot()
        src.close()
    else:
        init = ol.oracle_imageplane(spec)
        ol.oracle().kro_redshift_start_f64(-SPIN, 0.0, 1, 0, ol.ptr(init), len(init))
        t0 = time.perf_counter()
        out, _ = ol.oracle_trace(p, init, nthreads=cores)
        wall = time.perf_counter() - t0
    live = out["steps"] != -1
    steps = int(np.abs(out["steps"][live].astype(np.int64)).sum())
    # parity of the image planes on this sample: GPU (same init rays, headline arithmetic, through the C ABI) vs CPU
    o = ol.oracle()
    o.kro_redshift_f64(-SPIN, -1.0, 1, 0, 0, ol.ptr(out), len(out))
    o.kro_range_phi_f64(-math.pi, math.pi, ol.ptr(out), len(out))
    img = 256
    b = capi.ImageBins()
    b.x0, b.y0, b.img_dx, b.img_dy = -30.0, -30.0, 60.0 / img, 60.0 / img
    b.r_isco, b.r_disc = wl.bins.r_isco, wl.bins.r_disc
    b.q1, b.rb1, b.q2, b.rb2, b.q3 = wl.bins.q1, wl.bins.rb1, wl.bins.q2, wl.bins.rb2, wl.bins.q3
    b.img_nx, b.img_ny, b.flip_image, b.pad = img, img, 1, 0
    npix = img * img
    w_n = np.zeros(npix, dtype=np.int32)
    keys = ("flux", "r", "phi", "enshift", "time", "emis")
    w_pl = {k: np.zeros(npix) for k in keys}
    dc = C.c_int64()
    o.kro_reduce_image_f64(C.byref(b), ol.ptr(out), len(out), ol.ptr(w_n), *[ol.ptr(w_pl[k]) for k in keys], C.byref(dc))
    gpu, _ = api.trace(p, init)
    api.redshift(-SPIN, -1.0, 1, 0, gpu)
    api.range_phi(gpu)
    got = api.reduce_image(b, gpu)
    same = got["nrays"] == w_n
    worst = {}
    for k in ("r", "enshift", "flux", "time"):
        m = same & (w_n > 0)
        with np.errstate(invalid="ignore", divide="ignore"):
            worst[k] = float(np.max(np.abs(got[k][m] - w_pl[k][m]) / np.abs(w_pl[k][m]))) if m.any() else 0.0
    ints_differ = 0
    for k in ("status", "rdot_flips", "equatorial_crossings", "steps"):
        ints_differ += int((gpu[k] != out[k]).sum())
    # the strict arithmetic on the same rays: how many carry the CPU's bits in every output of trace + redshift + range_phi
    strict, _ = api.trace(capi.copy_params(p, flags=0), init)
    api.redshift(-SPIN, -1.0, 1, 0, strict)
    api.range_phi(strict)
    bits = live.copy()
    for k in ("t", "r", "theta", "phi", "redshift"):
        bits &= (strict[k].view(np.int64) == out[k].view(np.int64)) | (np.isnan(strict[k]) & np.isnan(out[k]))
    for k in ("status", "rdot_flips", "equatorial_crossings", "steps"):
        bits &= strict[k] == out[k]
    strict_bits = float(bits[live].mean())
    # ... and the pipeline bench.py TIMES, on this coarser grid: device-built rays (kr_imageplane_init_emit_runs_dev_f64), fused passes, planes left in HBM.
    # The device constructor's acos / asin / atan2 / tan are not glibc's on every ray, so a ray next to a pixel or disc edge may land
    # on the other side: reported, not hidden.
    import torch
    lib, vp = api.lib(), C.c_void_p
    n_dev = api.imageplane_count(spec)[0]
    d_rays = torch.empty(n_dev * capi.RAY_F64.itemsize, dtype=torch.uint8, device="cuda")
    d_res = torch.zeros(7 * npix + 1, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(lib, lib.kr_imageplane_init_emit_runs_dev_f64(C.byref(spec), 0, 1, 1, 0.0, 1, 0, vp(d_rays.data_ptr()), n_dev, vp(stream)), "init_emit")
    pst = api.trace_dev(p, d_rays.data_ptr(), n_dev, stream=stream, want_stats=True)
    capi.check(lib, lib.kr_post_image_dev_f64(-SPIN, -1.0, 1, 0, 0, -math.pi, math.pi, C.byref(b), vp(d_rays.data_ptr()), n_dev, vp(d_res.data_ptr()), vp(stream)), "post")
    torch.cuda.synchronize()
    hp = d_res.cpu().numpy()
    dev_planes = api.image_planes_from_words(hp, img, img)
    psame = dev_planes["nrays"] == w_n
    pworst = {}
    for k in ("r", "enshift", "flux", "time"):
        m = psame & (w_n > 0)
        with np.errstate(invalid="ignore", divide="ignore"):
            pworst[k] = float(np.max(np.abs(dev_planes[k][m] - w_pl[k][m]) / np.abs(w_pl[k][m]))) if m.any() else 0.0
    pipeline_planes_check = {
        "what": "one run of the timed pipeline (device ImagePlane + redshift_start -> trace -> redshift + range_phi + planes) on the CPU leg's grid",
        "pixels_count_mismatch": int((~psame).sum()), "pixels_excluded_from_the_sum_check": int((~psame & (w_n > 0)).sum()),
        "max_count_diff": int(np.abs(dev_planes["nrays"].astype(np.int64) - w_n).max()),
        "disc_rays_device": int(round(float(hp[-1]))), "disc_rays_cpu": int(dc.value), "max_rel_diff_of_pixel_sums(RADIUS, ENSHIFT, FLUX, TIME)": pworst,
        "rk_steps_device": int(pst["steps_total"]), "rk_steps_cpu": steps, "tolerance": 1e-6}
    planes_check = {"image": f"{img}x{img}", "lit_pixels": int((w_n > 0).sum()), "pixels_count_mismatch": int((~same).sum()), "disc_rays_cpu": int(dc.value),
                    "disc_rays_gpu": int(got["disc_count"]), "max_rel_diff_of_pixel_sums(RADIUS, ENSHIFT, FLUX, TIME)": worst, "tolerance": 1e-6,
                    "ray_integer_fields_differ": ints_differ,
                    "strict_arithmetic_bit_identical_frac(t, r, theta, phi, redshift, integer fields)": strict_bits}
    return {"value": int(live.sum()) / wall, "unit": "rays/s", "cores": cores, "kind": kind, "steps_per_sec": steps / wall, "wall_s": wall,
            "sample": f"same image plane on a {N + 1}x{N + 1} ray grid: {int(live.sum())} rays, {steps} steps, run_raytrace only",
            "pipeline_planes_check": pipeline_planes_check, "planes_check": planes_check}


if __name__ == "__main__":
    main()
