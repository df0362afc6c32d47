"""raytrace_cpu_amd -- MI355X-native Kerr null-geodesic hot path behind the reference's class API.

Importing the package sets the process default GPU_MAX_HW_QUEUES=16 (unless the user chose a value): overlapping traces need hardware queues of
their own and the HIP runtime reads the variable ONCE, when it starts -- which `import torch` or the first HIP call of any library may do.  Import
this package before torch (bench.py, tests/conftest.py and the apps do); include/kr_trace.h::kr_configure_process has the measurements."""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
