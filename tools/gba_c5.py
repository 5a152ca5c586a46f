import numpy as np, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
ncam, npts, nobs, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
t0 = time.perf_counter()
ST = os.environ.get("ORBHIP_BENCH_STRUCTURE", "band")      # band | covis | dense | loop (synth.make_ba_graph_covis)
g = synth.make_ba_graph(1, ncam=ncam, npts=npts, nobs=nobs, n_fixed=1) if ST == "band" else synth.make_ba_graph_covis(3000, ncam=ncam, npts=npts, nobs=nobs, structure=ST)
print('gen', time.perf_counter() - t0)
for rep in range(2):
    t0 = time.perf_counter()
    poses, pts, s = optimizer.global_bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"],
                                                       g["obs_uv"], g["obs_inv_sigma2"], n_iterations=iters)
    dt = time.perf_counter() - t0
    print('GBA ms', dt * 1e3, s)
n = 6 * (ncam - 1)
print('cholesky flops/iter %.2f GFLOP' % (n ** 3 / 3 / 1e9), 'plan', optimizer.get_last_plan())
