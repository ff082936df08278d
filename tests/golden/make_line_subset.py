"""Builds tc-viml_amd/data/lines3d_v101_subset.txt: a 256-row sample of the reference's V1_01
prior 3D line map (benchmark_publisher/config/V1_01_easy/line_3d.txt, 891 rows x 6 doubles:
start xyz, end xyz in the map frame).  Data fixture only; run in the authoring container."""
import numpy as np, os
src = "/root/reference/benchmark_publisher/config/V1_01_easy/line_3d.txt"
a = np.loadtxt(src)
idx = np.linspace(0, a.shape[0] - 1, 256).astype(int)
out = os.path.join(os.path.dirname(__file__), "..", "..", "tc-viml_amd", "data", "lines3d_v101_subset.txt")
np.savetxt(out, a[idx], fmt="%.6f")
print("wrote", out, a[idx].shape)
