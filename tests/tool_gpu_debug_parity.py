"""Debug helper (GPU box): per-ray disagreement report of the HIP path vs golden for every case/run."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_cases as gc, parity
from raytrace_cpu_amd import api, capi
from test_gpu_parity import hip_pipeline

FLAGS = int(os.environ.get("KR_FLAGS", "0"))
only = sys.argv[1:]
for cname, case in gc.cases().items():
    g = np.load(gc.golden_path(cname))
    for run, params in case["runs"].items():
        if only and f"{cname}/{run}" not in only: continue
        out, st = hip_pipeline(case, params, g["init"], FLAGS)
        want = g[f"final__{run}"]
        res = parity.compare_rays(out, want, rtol=parity.rtol_for(params), check_redshift=True)
        live = want["steps"] != -1
        exact = sum(1 for i in np.flatnonzero(live) if out[i].tobytes() == want[i].tobytes())
        print(f"{cname}/{run}: traced {res['n_traced']} bad {res['n_bad']} worst_ok {res['worst_ok']:.2e} bitwise-identical rays {exact} kernel_ms {st['kernel_ms']:.2f} steps {st['steps_total']}")
        for i in res["bad_index"][:12]:
            a, b = out[i], want[i]
            print(f"   ray {i}: steps {a['steps']}/{b['steps']} status {a['status']}/{b['status']} flips {a['rdot_flips']}/{b['rdot_flips']} eq {a['equatorial_crossings']}/{b['equatorial_crossings']} "
                  f"r {a['r']:.15g}/{b['r']:.15g} th {a['theta']:.12g}/{b['theta']:.12g} t {a['t']:.12g}/{b['t']:.12g} phi {a['phi']:.10g}/{b['phi']:.10g} g {a['redshift']:.8g}/{b['redshift']:.8g}")
