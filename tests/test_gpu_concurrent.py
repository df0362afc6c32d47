"""GPU: several traces in flight on one device (kr_trace_async_* / kr_trace_wait; per-call workspaces, include/kr_trace.h).

The multi-launch drivers of the reference -- one run_raytrace per source radius (src/return_radiation/
disc_source_photonfrac_r.cpp:74-92), one per tolerance (src/tests/emissivity_rk45_tol_sweep.py:38) -- are the callers that
want their launches to overlap.  Overlapping traces on different streams must give, ray for ray and bit for bit, what the
same traces give one after the other, and each ticket must report its own counters."""
import ctypes as C

import numpy as np
import pytest

import bench
import parity
from raytrace_cpu_amd import api, capi

pytestmark = pytest.mark.gpu


class DeviceRays:
    """A device buffer holding a PointSource grid, initialised on the device."""

    def __init__(self, lib, spec, V=0.0):
        self.lib, self.spec = lib, spec
        self.n = api.pointsource_count(spec)[0]
        self.d = C.c_void_p()
        capi.check(lib, lib.kr_malloc(C.byref(self.d), self.n * capi.RAY_F64.itemsize), "kr_malloc")
        self.V = V

    def init(self, stream=None):
        capi.check(self.lib, self.lib.kr_pointsource_init_emit_dev_f64(C.byref(self.spec), 0, 1, self.V, 0, 0, self.d, self.n, C.c_void_p(stream or 0)), "init")

    def fetch(self):
        out = np.zeros(self.n, dtype=capi.RAY_F64)
        capi.check(self.lib, self.lib.kr_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.d, out.nbytes), "d2h")
        return out

    def free(self):
        self.lib.kr_free(self.d)


same_bits = parity.same_records      # (bit for bit; a NaN equals a NaN)


@pytest.mark.parametrize("flags", [pytest.param(capi.FLAG_HYBRID, id="hybrid"), pytest.param(0, id="strict-split")])
def test_overlapping_traces_equal_sequential_ones(krlib, flags):
    lib = krlib
    # two different lamp-post grids, both large enough for the split launch (>= 2^18 rays)
    specs = [bench.make_spec(capi, bench.grid_spacing_for(4.0e5)), bench.make_spec(capi, bench.grid_spacing_for(6.0e5))]
    specs[1].pos[1] = 5.0
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, flags
    bufs = [DeviceRays(lib, s) for s in specs]
    try:
        # sequential reference: default stream, one after the other, each waited for
        want, want_st = [], []
        for b in bufs:
            b.init()
            want_st.append(api.trace_dev(p, b.d.value, b.n))
            want.append(b.fetch())
        assert all(st["rays_strict_side"] > 0 for st in want_st)            # the split path, with its side launch
        # overlapped: two streams, both traces enqueued before either is waited for; three rounds so that workspaces get reused
        streams = []
        for _ in bufs:
            s = C.c_void_p()
            capi.check(lib, lib.kr_stream_create(C.byref(s)), "stream")
            streams.append(s)
        for rnd in range(3):
            tickets = []
            for b, s in zip(bufs, streams):
                b.init(s.value)
                tickets.append(api.trace_async(p, b.d.value, b.n, stream=s.value))
            stats = [api.trace_wait(t) for t in reversed(tickets)][::-1]      # waited for in the opposite order
            for b, w, st, wst in zip(bufs, want, stats, want_st):
                got = b.fetch()
                assert same_bits(got, w), f"round {rnd}: overlapped trace differs from the sequential one"
                for k in ("rays_total", "rays_traced", "steps_total", "rays_strict_side"):
                    assert st[k] == wst[k], (k, st[k], wst[k])
                assert st["kernel_ms"] > 0 and st["strict_side_ms"] > 0 and st["main_ms"] > 0
        for s in streams:
            lib.kr_stream_destroy(s)
    finally:
        for b in bufs:
            b.free()


def test_fire_and_forget_calls_on_many_streams(krlib):
    """kr_trace_dev_f64(stats = NULL) from a loop over four streams, more calls than streams: every buffer ends up traced."""
    lib = krlib
    spec = bench.make_spec(capi, bench.grid_spacing_for(3.0e5))
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, capi.FLAG_HYBRID
    bufs = [DeviceRays(lib, spec) for _ in range(8)]
    streams = []
    try:
        for _ in range(4):
            s = C.c_void_p()
            capi.check(lib, lib.kr_stream_create(C.byref(s)), "stream")
            streams.append(s)
        for i, b in enumerate(bufs):
            s = streams[i % 4]
            b.init(s.value)
            assert api.trace_dev(p, b.d.value, b.n, stream=s.value, want_stats=False) is None
        for s in streams:
            capi.check(lib, lib.kr_synchronize(s), "sync")
        first = bufs[0].fetch()
        assert (first["steps"][first["steps"] != -1] != 0).all()
        for b in bufs[1:]:
            assert same_bits(b.fetch(), first)
    finally:
        for s in streams:
            lib.kr_stream_destroy(s)
        for b in bufs:
            b.free()


def test_list_overflow_goes_to_the_ordinary_strict_launch(krlib):
    """A source made only of ill-conditioned rays (every ray in the beta = -pi meridional plane) flags more rays than the side
    launch's list holds; the overflow launch must trace the rest, with the bits of a plain strict launch."""
    lib = krlib
    import math
    spec = bench.make_spec(capi, 1.99 / 299999.0)                # 3e5 values of cos(alpha) ...
    spec.beta0, spec.betamax, spec.dbeta = -math.pi, -math.pi + 1e-9, 1.0       # ... in one column
    n = api.pointsource_count(spec)[0]
    assert n >= (1 << 18)
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.steplim = capi.RK4, bench.R_MAX, 400            # (bounded: these are the polar-axis rays)
    b = DeviceRays(lib, spec)
    try:
        import os
        b.init()
        st = api.trace_dev(p, b.d.value, b.n)
        got = b.fetch()
        assert st["rays_strict_side"] == st["rays_traced"] > 32768
        os.environ["KR_NO_ISOLATE"] = "1"
        try:
            b.init()
            st1 = api.trace_dev(p, b.d.value, b.n)
            want = b.fetch()
        finally:
            del os.environ["KR_NO_ISOLATE"]
        assert st1["rays_strict_side"] == 0 and st1["steps_total"] == st["steps_total"]
        assert same_bits(got, want)
    finally:
        b.free()


def test_batch_of_traces_equals_the_single_calls(krlib):
    """kr_trace_batch_async_f64 (front halves of all traces first): RK45 points of different tolerance on a small grid -- strict
    traces below the split threshold are split in a batch -- give exactly what one kr_trace_dev_f64 call each gives."""
    lib = krlib
    spec = bench.make_spec(capi, 0.02)
    tols = [1e-6, 1e-8, 1e-10]
    bufs = [DeviceRays(lib, spec) for _ in tols]
    params = []
    for tol in tols:
        p = capi.default_params(bench.SPIN)
        p.integrator, p.r_max, p.rk45_tol = capi.RK45, bench.R_MAX, tol
        params.append(p)
    streams = []
    try:
        want, want_st = [], []
        for b, p in zip(bufs, params):
            b.init()
            want_st.append(api.trace_dev(p, b.d.value, b.n))
            want.append(b.fetch())
        assert all(st["rays_strict_side"] == 0 for st in want_st) and bufs[0].n < (1 << 18)      # single launches when called one by one
        for _ in bufs:
            s = C.c_void_p()
            capi.check(lib, lib.kr_stream_create(C.byref(s)), "stream")
            streams.append(s)
        for b, s in zip(bufs, streams):
            b.init(s.value)
        tickets = api.trace_batch_async(params, [b.d.value for b in bufs], [b.n for b in bufs], [s.value for s in streams])
        stats = [api.trace_wait(t) for t in tickets]
        for b, w, st, wst in zip(bufs, want, stats, want_st):
            assert st["rays_strict_side"] > 0                     # split in the batch ...
            assert same_bits(b.fetch(), w)                        # ... with the bits of the single launch
            for k in ("rays_traced", "steps_total", "rk45_attempts", "rk45_rejects"):
                assert st[k] == wst[k], k
    finally:
        for s in streams:
            lib.kr_stream_destroy(s)
        for b in bufs:
            b.free()


@pytest.mark.parametrize("merged", [True, False])
@pytest.mark.parametrize("flags,method", [(capi.FLAG_HYBRID, capi.RK4), (0, capi.RK45), (capi.FLAG_HYBRID, capi.EULER)])
def test_merged_batch_is_bitwise_the_single_traces(krlib, flags, method, merged, monkeypatch):
    """A batch whose traces share their kernel instances runs as ONE side launch + ONE main launch over all of them
    (trace_multi_kernel; KR_NO_MERGED_BATCH=1: one pair of launches per trace, front halves first).  Either way every trace gets the
    bits and the counters of a call of its own -- different sources, sizes and tolerances in one batch, all on one stream."""
    lib = krlib
    if not merged:
        monkeypatch.setenv("KR_NO_MERGED_BATCH", "1")
    specs = [bench.make_spec(capi, 0.02), bench.make_spec(capi, 0.013), bench.make_spec(capi, 0.03), bench.make_spec(capi, 0.017)]
    specs[1].pos[1], specs[2].pos[1] = 5.0, 20.0
    bufs = [DeviceRays(lib, s) for s in specs]
    params = []
    for i in range(len(specs)):
        p = capi.default_params(bench.SPIN)
        p.integrator, p.r_max, p.flags, p.rk45_tol = method, bench.R_MAX, flags, [1e-6, 1e-8, 1e-7, 1e-9][i]
        params.append(p)
    try:
        want, want_st = [], []
        for b, p in zip(bufs, params):
            b.init()
            want_st.append(api.trace_dev(p, b.d.value, b.n))
            want.append(b.fetch())
        for b in bufs:
            b.init()
        tickets = api.trace_batch_async(params, [b.d.value for b in bufs], [b.n for b in bufs], None)
        stats = [api.trace_wait(t) for t in tickets]
        for b, w, st, wst in zip(bufs, want, stats, want_st):
            assert same_bits(b.fetch(), w)
            for k in ("rays_total", "rays_traced", "steps_total", "rk45_attempts", "rk45_rejects", "rk45_stationary_steps", "rk45_extrapolated_steps"):
                assert st[k] == wst[k], k
            assert st["rays_strict_side"] > 0
    finally:
        for b in bufs:
            b.free()


def test_two_host_threads_trace_at_the_same_time(krlib):
    """Two host threads, each with a host ray array of its own, inside kr_trace_f64 at the same time (ctypes drops the GIL for the
    call): every call draws its own workspace, so each thread gets, bit for bit and counter for counter, what it gets alone.
    (Round 1 kept one process-wide set of queue counters: ADVICE r01, 'shared DeviceScratch race'.)"""
    import threading
    specs = [bench.make_spec(capi, bench.grid_spacing_for(4.0e5)), bench.make_spec(capi, bench.grid_spacing_for(3.0e5))]
    specs[1].pos[1] = 6.0
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, capi.FLAG_HYBRID
    inits = [api.pointsource_init(s) for s in specs]
    alone = [api.trace(p, r) for r in inits]
    assert all(st["rays_strict_side"] > 0 for _, st in alone)
    errors = []

    def worker(i):
        try:
            for rnd in range(3):
                out, st = api.trace(p, inits[i])
                if not same_bits(out, alone[i][0]):
                    errors.append(f"thread {i} round {rnd}: rays differ")
                for k in ("rays_traced", "steps_total", "rays_strict_side"):
                    if st[k] != alone[i][1][k]:
                        errors.append(f"thread {i} round {rnd}: {k} {st[k]} != {alone[i][1][k]}")
        except Exception as e:                         # noqa: BLE001 -- reported through the list
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []


def test_wait_many_sums_the_tickets_of_a_batch(krlib):
    """kr_trace_wait_many: one call for all tickets of a batch; the summed counters equal the sums over single kr_trace_dev_f64 calls."""
    lib = krlib
    specs = [bench.make_spec(capi, 0.02), bench.make_spec(capi, 0.013), bench.make_spec(capi, 0.03)]
    specs[1].pos[1] = 5.0
    bufs = [DeviceRays(lib, s) for s in specs]
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, capi.FLAG_HYBRID
    try:
        want = {"rays_total": 0, "rays_traced": 0, "steps_total": 0, "rays_strict_side": 0}
        finals = []
        for b in bufs:
            b.init()
            st = api.trace_dev(p, b.d.value, b.n)
            finals.append(b.fetch())
        for b in bufs:
            b.init()
        tickets = api.trace_batch_async([p] * 3, [b.d.value for b in bufs], [b.n for b in bufs], None)
        single = [api.trace_dev(p, b.d.value, b.n) for b in []]          # (nothing: the batch owns the buffers now)
        tot = api.trace_wait_many(tickets)
        for b, f in zip(bufs, finals):
            assert same_bits(b.fetch(), f)
        # the batch splits every trace (strict side launch), a lone call below 2^18 rays does not: compare what does not depend on that
        assert tot["rays_total"] == sum(b.n for b in bufs)
        assert tot["rays_traced"] == sum(int((f["steps"] != -1).sum()) for f in finals)
        assert tot["steps_total"] == sum(int(np.abs(f["steps"][f["steps"] != -1].astype(np.int64)).sum()) for f in finals)
        assert tot["rays_strict_side"] > 0 and tot["kernel_ms"] > 0
    finally:
        for b in bufs:
            b.free()


def test_many_streams_share_side_streams_and_shutdown_releases_them(krlib):
    """A process that makes one stream per job: beyond the per-device table of side streams (16) callers share this device's
    side streams, kr_stream_destroy takes a stream's entry (and its side stream) with it, growing workspaces never free in the launch
    path, and kr_shutdown gives everything back -- after which the library works as before.  Results: bit for bit the first trace's."""
    lib = krlib
    spec = bench.make_spec(capi, bench.grid_spacing_for(3.0e4))
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, capi.FLAG_HYBRID
    buf = DeviceRays(lib, spec)
    big = DeviceRays(lib, bench.make_spec(capi, bench.grid_spacing_for(9.0e4)))      # a larger n on the same pooled workspaces: the mask grows
    try:
        buf.init()
        want_st = api.trace_dev(p, buf.d.value, buf.n)
        want = buf.fetch()
        assert want_st["rays_strict_side"] > 0
        streams = []
        for k in range(40):
            s = C.c_void_p()
            capi.check(lib, lib.kr_stream_create(C.byref(s)), "stream")
            streams.append(s)
            buf.init(s.value)
            st = api.trace_dev(p, buf.d.value, buf.n, stream=s.value)
            assert st["steps_total"] == want_st["steps_total"]
            assert same_bits(buf.fetch(), want), f"stream {k}"
            if k % 3 == 2:                                   # some are destroyed on the way (their table entries go), most stay
                capi.check(lib, lib.kr_stream_destroy(streams.pop(-2)), "destroy")
        big.init(streams[0].value)
        st_big = api.trace_dev(p, big.d.value, big.n, stream=streams[0].value)
        assert st_big["rays_traced"] == int((big.fetch()["steps"] != -1).sum())
        for s in streams:
            capi.check(lib, lib.kr_stream_destroy(s), "destroy")
        capi.check(lib, lib.kr_shutdown(), "kr_shutdown")
        buf.init()
        st = api.trace_dev(p, buf.d.value, buf.n)            # pools and side streams are rebuilt on demand
        assert st["steps_total"] == want_st["steps_total"] and same_bits(buf.fetch(), want)
    finally:
        buf.free()
        big.free()


def test_shared_side_streams_outlive_one_of_their_users_and_shutdown_waits_for_tickets(krlib):
    """Host-state rules of the split trace (kr_trace.hip): (i) beyond 16 caller streams per device the side streams are shared, every sharer is
    counted, and destroying one caller stream must leave the side stream alive for the others -- 20 streams trace, the first 10 are destroyed,
    the other 10 trace again, same bits; (ii) kr_shutdown is refused while a ticket is outstanding and works afterwards, after which the
    library starts from scratch (new workspaces, new side streams, new PointSource tables)."""
    lib = krlib
    spec = bench.make_spec(capi, bench.grid_spacing_for(3.0e5))
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, capi.FLAG_HYBRID
    b = DeviceRays(lib, spec)
    streams = []
    try:
        b.init()
        api.trace_dev(p, b.d.value, b.n)
        want = b.fetch()
        for _ in range(20):
            s = C.c_void_p()
            capi.check(lib, lib.kr_stream_create(C.byref(s)), "stream")
            streams.append(s)
        for s in streams:                      # every stream runs a split trace: 16 side streams, 4 sharers
            b.init(s.value)
            api.trace_dev(p, b.d.value, b.n, stream=s.value)
        for s in streams[:10]:
            capi.check(lib, lib.kr_stream_destroy(s), "destroy")
        streams = streams[10:]
        for s in streams:
            b.init(s.value)
            api.trace_dev(p, b.d.value, b.n, stream=s.value)
            assert same_bits(b.fetch(), want)
        # (ii)
        b.init()
        t = api.trace_async(p, b.d.value, b.n)
        assert lib.kr_shutdown() == capi.KR_EINVAL and b"outstanding" in lib.kr_last_error()
        api.trace_wait(t)
        assert same_bits(b.fetch(), want)
        for s in streams:
            capi.check(lib, lib.kr_stream_destroy(s), "destroy")
        streams = []
        capi.check(lib, lib.kr_shutdown(), "shutdown")
        b.init()
        api.trace_dev(p, b.d.value, b.n)
        assert same_bits(b.fetch(), want)
    finally:
        for s in streams:
            lib.kr_stream_destroy(s)
        b.free()


def test_progress_of_a_trace_in_flight(krlib):
    """run_raytrace's show_progress (raytracer.cpp:84-85, :107-115) on the HIP path: kr_trace_poll reads the work queue's head while the persistent
    kernels run -- monotone, between 0 and n, n once the trace has finished, and it must actually MOVE during a 0.15-s launch (a read that only
    completes when the kernels have drained would be no progress report at all); kr_trace_progress_f64 reports multiples of `every`."""
    import time
    lib = krlib
    spec = bench.make_spec(capi, bench.grid_spacing_for(1.0e7))
    p = capi.default_params(bench.SPIN)
    p.integrator, p.r_max, p.flags = capi.RK4, bench.R_MAX, 0            # all-strict: ~0.16 s
    b = DeviceRays(lib, spec)
    try:
        b.init()
        capi.check(lib, lib.kr_synchronize(None), "sync")
        t = api.trace_async(p, b.d.value, b.n)
        seen = []
        started, fin = C.c_int64(), C.c_int32()
        while True:
            capi.check(lib, lib.kr_trace_poll(t, C.byref(started), C.byref(fin)), "poll")
            seen.append(started.value)
            if fin.value:
                break
            time.sleep(0.005)
        st = api.trace_wait(t)
        assert seen[-1] == b.n == st["rays_total"]
        assert all(x <= y for x, y in zip(seen, seen[1:])) and all(0 <= x <= b.n for x in seen)
        assert any(0 < x < b.n for x in seen), seen[:5] + seen[-5:]
    finally:
        b.free()
    # the host-pointer form with a callback, as the class mirror uses it
    spec = bench.make_spec(capi, bench.grid_spacing_for(2.0e6))
    rays = api.pointsource_init(spec)
    marks = []
    cb = capi.PROGRESS_FN(lambda at, total, user: marks.append((at, total)))
    st = capi.Stats()
    capi.check(lib, lib.kr_trace_progress_f64(C.byref(p), rays.ctypes.data_as(C.c_void_p), len(rays), C.byref(st), 250000, cb, None), "trace with progress")
    assert st.rays_traced > 1.9e6 and marks, marks
    assert all(at % 250000 == 0 and 0 < at <= len(rays) and total == len(rays) for at, total in marks)
    assert all(x[0] < y[0] for x, y in zip(marks, marks[1:]))
    assert (rays["steps"][rays["steps"] != -1] != 0).all()
