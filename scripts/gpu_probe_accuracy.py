#!/usr/bin/env python3
"""GPU box: accuracy (ulp) of the reciprocal / square-root building blocks against numpy, via kr_debug_arith_f64:
raw v_rcp_f64 / v_rsq_f64, one Newton step, and the routines the fast path uses.  Measured 2026-10: rcp raw 4e8 ulp, +1 step
18 ulp max, fast_rcp (2 steps) 1 ulp; rsq raw 4.5e8, +1 coupled step 35 ulp max, fast_sqrt 0 ulp -- which is why the fast path keeps
both refinement steps (dropping them would buy ~6 % at 20-35 ulp per operation)."""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_gpu_primitives import probe, ulps, operands
rng=np.random.default_rng(7)
a,b=operands(rng, 2_000_000)
for op,name in ((11,'rcp raw'),(12,'rcp+1'),(6,'fast_rcp (2 steps)')):
    u=ulps(probe(op,a,b), a/b); print(name, 'max ulp %.3g  mean %.3g'%(u.max(), u.mean()))
x=np.abs(a)
for op,name in ((13,'rsq raw'),(14,'rsq+1 coupled'),(7,'fast_sqrt')):
    u=ulps(probe(op,x), np.sqrt(x)); print(name, 'max ulp %.3g  mean %.3g'%(u.max(), u.mean()))
