F32_STEPLIM = 20000   # keep in sync with tests/golden/make_golden_f32.py
