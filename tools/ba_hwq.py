"""LocalBA throughput vs number of host threads; run with GPU_MAX_HW_QUEUES=<n> in the environment."""
import numpy as np, sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=2)
local = np.ones(100, np.uint8)
args = (g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
optimizer.local_bundle_adjustment(*args)
for T in [int(a) for a in sys.argv[1:]] or [12]:
    n_each = 6
    bar = threading.Barrier(T + 1)
    def work():
        for _ in range(2): optimizer.local_bundle_adjustment(*args)
        bar.wait()
        for _ in range(n_each): optimizer.local_bundle_adjustment(*args)
    ths = [threading.Thread(target=work) for _ in range(T)]
    for t in ths: t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print('HWQ', os.environ.get("GPU_MAX_HW_QUEUES"), 'threads', T, 'solves/s %.1f' % (T * n_each / dt), flush=True)
