"""One configuration of profiles/tools/random_sweep.py in detail: the features with the largest |depth - oracle|.
usage: dbg_seed.py seed [route]     route: default | fused | wave-only | dense"""
import os
import sys
from pathlib import Path
route = sys.argv[2] if len(sys.argv) > 2 else "fused"
if route != "default":
    os.environ["MLD_FORCE_WAVE_PATH" if route == "wave-only" else "MLD_FORCE_THREAD_PATH"] = "1"
    if route == "dense":
        os.environ["MLD_K1MAX"] = "48"
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import numpy as np
from mono_lidar_depth_amd import GroundPlane, synth, capi
if route != "default":
    capi._lib = capi.load_ab()
from helpers import make_estimator, run_oracle
from test_randomized_gpu import _random_setup
seed=int(sys.argv[1]) if len(sys.argv)>1 else 1990
P, cam, T, scanner, kw = _random_setup(seed)
print(kw); print(cam.width, cam.height, cam.focal_length, scanner)
cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 5)
uv = synth.make_features(900, seed=300 + seed, width=cam.width, height=cam.height)
plane = synth.make_ground_plane(cloud)
est = make_estimator(P, camera=cam, T=T)
d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
ref, (d0, t0) = run_oracle(P, cloud, uv, plane, camera=cam, T=T)
diff=np.abs(np.nan_to_num(d)-np.nan_to_num(d0))
idx=np.argsort(-diff)[:8]
for i in idx: print(i, uv[i], t[i], t0[i], d[i], d0[i], diff[i])
print("types equal", np.array_equal(t,t0))
