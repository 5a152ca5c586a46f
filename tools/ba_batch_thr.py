"""LocalBA (C4 size) throughput: batch size B per call x T host threads.  usage: [ORBHIP_BENCH_STRUCTURE=band|covis|dense] ba_batch_thr.py B:T[:N] [B:T[:N] ...]"""
import numpy as np, sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
ST = os.environ.get("ORBHIP_BENCH_STRUCTURE", "band")      # band (SURVEY 8(d)'s odometry band, as bench.py's c4_batched) | covis | dense (synth.make_ba_graph_covis)
gs = [synth.make_ba_graph(s, ncam=100, npts=10000, nobs=50000, n_fixed=1) if ST == "band" else
      synth.make_ba_graph_covis(100 + s, ncam=100, npts=10000, nobs=50000, structure=ST) for s in range(16)]      # as bench.py: gauge keyframe only, 16 distinct local maps
local = np.ones(100, np.uint8)
def prob(g): return (g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
for spec in sys.argv[1:]:
    f = [int(x) for x in spec.split(":")]
    B, T = f[0], f[1]
    probs = [prob(gs[i % 16]) for i in range(B)]
    n_each = f[2] if len(f) > 2 else max(2, 48 // (B * T))      # B:T:N = N batches per thread in the timed part
    bar = threading.Barrier(T + 1)
    def work():
        try:
            for _ in range(2): optimizer.local_bundle_adjustment_batch(probs)
        except BaseException:
            bar.abort()                      # (a worker that dies must not leave the others waiting at the barrier for ever)
            raise
        bar.wait()
        for _ in range(n_each):
            t1 = time.perf_counter(); optimizer.local_bundle_adjustment_batch(probs)
            if os.environ.get('ORBHIP_BA_TIMING'): print('[python] call %.2f ms' % ((time.perf_counter() - t1) * 1e3), file=sys.stderr, flush=True)
    ths = [threading.Thread(target=work) for _ in range(T)]
    for t in ths: t.start()
    try:
        bar.wait(timeout=600)
    except threading.BrokenBarrierError:
        sys.exit("a worker thread failed")
    t0 = time.perf_counter()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print('batch', B, 'threads', T, 'solves/s %.1f' % (B * T * n_each / dt), 'ms/batch %.1f' % (dt / n_each * 1e3), flush=True)
